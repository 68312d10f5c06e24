#!/bin/bash
# Runs on the GPU box: one kernel-trace pass and separate PMC passes over the
# bench command of one workload (the hot path only, so every launch belongs to
# the timed pass).  Results land under gpurun_out/<tag>_{trace,fetch,write,sq,sq2};
# tools/summarize_profile.py turns them into the files kept under profiles/.
#   tools/profile_bench.sh <tag> [workload] [steps] [warmup]
tag=${1:-prof}
wl=${2:-alarm37_1e6_bdeu_mp3}
steps=${3:-20}
warm=${4:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- \
  python3 bench.py --workload $wl --steps $steps --warmup $warm --only-hot-path > gpurun_out/${tag}_trace.log 2>&1 || exit 1
grep '^{' gpurun_out/${tag}_trace.log > gpurun_out/${tag}_bench_under_trace.json
psteps=$(( steps < 3 ? steps : 3 ))
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- \
  python3 bench.py --workload $wl --steps $psteps --warmup 1 --only-hot-path > gpurun_out/${tag}_fetch.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- \
  python3 bench.py --workload $wl --steps $psteps --warmup 1 --only-hot-path > gpurun_out/${tag}_write.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --output-format csv -d gpurun_out/${tag}_sq -- \
  python3 bench.py --workload $wl --steps $psteps --warmup 1 --only-hot-path > gpurun_out/${tag}_sq.log 2>&1 || exit 4
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU \
  --output-format csv -d gpurun_out/${tag}_sq2 -- \
  python3 bench.py --workload $wl --steps $psteps --warmup 1 --only-hot-path > gpurun_out/${tag}_sq2.log 2>&1 || exit 5
echo profiled
