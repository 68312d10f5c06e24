#!/bin/bash
# Runs on the GPU box: HBM traffic counters (FETCH_SIZE, then WRITE_SIZE, separate
# passes) for one pass over the 128-variable / 10^7-row workload.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --workload uniform128x4_1e7_bdeu_mp3 --steps 1 --warmup 1 --only-hot-path > gpurun_out/cfg5_bench.json 2> gpurun_out/cfg5_bench.err
echo "plain run done"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/cfg5_fetch -- \
  python3 bench.py --workload uniform128x4_1e7_bdeu_mp3 --steps 1 --warmup 0 --only-hot-path > gpurun_out/cfg5_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/cfg5_write -- \
  python3 bench.py --workload uniform128x4_1e7_bdeu_mp3 --steps 1 --warmup 0 --only-hot-path > gpurun_out/cfg5_write.log 2>&1
echo "write pass done"
