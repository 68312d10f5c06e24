"""Development helper: time one rank's share of a pass for every rank of a
hypothetical world size on ONE GPU (no collective), to see how the fixed part of
a pass bounds strong scaling."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openbnsl_amd import engine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="alarm37_1e6_bdeu_mp3")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--shard-lower", action="store_true",
                help="count the second-largest size in slices too (the sum over ranks is left out: "
                     "same counting work, no collective)")
args = ap.parse_args()
spec = bench.WORKLOADS[args.workload]
codes, card, names = bench.make_data(spec)
mp = spec["max_parents"]
kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
d = engine.Data(codes, card)
d.set_stream(torch.cuda.current_stream().cuda_stream)
tabs = [torch.empty(d.num_subsets(k) * k, dtype=torch.float64, device="cuda") for k in range(1, mp + 2)]
ptrs = [t.data_ptr() for t in tabs]
best = torch.empty((d.n, 3), dtype=torch.float64, device="cuda")
for world in [int(x) for x in args.worlds.split(",")]:
    worst = 0.0
    per = []
    gem = []
    for rank in range(world):
        for t in tabs:
            t.fill_(float("-inf"))

        cbuf = torch.empty(1 << 24, dtype=torch.int32, device="cuda")

        def step():
            d.pack()
            if args.shard_lower and world > 1:
                size, cells = d.precount_sharded(mp + 1, rank, world)
                if size:
                    d.counts_pointer(size)      # where the in-place all-reduce would run
                    d.counts_mark_summed(size)
            else:
                d.precount(mp + 1, rank, world)
            for k in range(1, mp + 2):
                d.score_all_device(k, ptrs[k - 1], kind=kind, ess=1.0, shard=rank, num_shards=world)
            d.best_parents_device(ptrs, mp, best.data_ptr())

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        for k in range(1, mp + 2):
            d.kernel_time(k, reset=True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        per.append(ms)
        gem.append(" ".join(f"{d.kernel_time(k)[0] / max(1, d.kernel_time(k)[1]):.3f}" for k in range(1, mp + 2)))
    for r, g in enumerate(gem):
        print(f"  rank {r}: count-kernel ms by size: {g}")
    print(f"world {world}: per-rank ms/step {' '.join(f'{x:.3f}' for x in per)}  max {max(per):.3f}", flush=True)
