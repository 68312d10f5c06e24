#!/bin/bash
# Runs on the GPU box: one kernel-trace pass and three separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, instruction counts) over the default bench command.  Results land under
# gpurun_out/<tag>_{trace,fetch,write}; tools/summarize_profile.py turns them into
# the files kept under profiles/.
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- \
  python3 bench.py --steps 20 --warmup 3 --only-hot-path > gpurun_out/${tag}_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- \
  python3 bench.py --steps 3 --warmup 1 --only-hot-path > gpurun_out/${tag}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- \
  python3 bench.py --steps 3 --warmup 1 --only-hot-path > gpurun_out/${tag}_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${tag}_sq -- \
  python3 bench.py --steps 3 --warmup 1 --only-hot-path > gpurun_out/${tag}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_grbm -- \
  python3 bench.py --steps 3 --warmup 1 --only-hot-path > gpurun_out/${tag}_grbm.log 2>&1
grep '^{' gpurun_out/${tag}_trace.log > gpurun_out/${tag}_bench_under_trace.json
echo profiled
