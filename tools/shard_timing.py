"""Development helper: time one rank's share of a pass for every rank of a
hypothetical world size on ONE GPU (no collective), to see how the fixed part of
a pass bounds strong scaling.  --sharding subsets: the rank counts its shard of the
largest subset size, its slice of the second largest and the smaller sizes whole;
--sharding rows: the rank holds a slice of the rows and counts every size over it
(the all-reduce of the count tables that follows on real ranks is left out in
both modes, so the scores of a 'rows' run are those of the slice, not the data set:
this tool measures time only)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openbnsl_amd import engine, pipeline
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="alarm37_1e6_bdeu_mp3")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=60)
ap.add_argument("--sharding", default="auto", choices=["auto", "subsets", "rows"])
args = ap.parse_args()
spec = bench.WORKLOADS[args.workload]
codes, card, names = bench.make_data(spec)
N = codes.shape[1]
mp = spec["max_parents"]
kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
stream = torch.cuda.current_stream().cuda_stream
whole = engine.Data(codes, card)
whole.set_stream(stream)
tabs = [torch.empty(whole.num_subsets(k) * k, dtype=torch.float64, device="cuda") for k in range(1, mp + 2)]
ptrs = [t.data_ptr() for t in tabs]
best = torch.empty((whole.n, 3), dtype=torch.float64, device="cuda")
for world in [int(x) for x in args.worlds.split(",")]:
    mode = args.sharding
    if mode == "auto":
        mode = pipeline.choose_sharding(card, N, mp, world)
    if world == 1:
        mode = "subsets"
    per, gem = [], []
    for rank in range(world):
        for t in tabs:
            t.fill_(float("-inf"))
        if mode == "rows":
            lo, hi = pipeline.row_slice(N, rank, world)
            d = engine.Data(codes[:, lo:hi], card, total_rows=N)
            d.set_stream(stream)
        else:
            d = whole

        def step():
            d.pack()
            if mode == "rows":
                d.precount_rows(mp + 1)         # (the all-reduce of the range it names would run here)
                d.counts_mark_summed(0)
            elif world > 1:
                size, cells = d.precount_sharded(mp + 1, rank, world)
                if size:
                    d.counts_pointer(size)      # where the in-place all-reduce would run
                    d.counts_mark_summed(size)
            else:
                d.precount(mp + 1, rank, world)
            d.score_pass(ptrs, mp, best.data_ptr(), kind=kind, ess=1.0, shard=rank, num_shards=world)

        d.kernel_timing(0)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        per.append(ms)
        # the count launches' own times from a few more passes with the timing events on
        d.kernel_timing(1)
        for k in range(1, mp + 2):
            d.kernel_time(k, reset=True)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        gem.append(" ".join(f"{d.kernel_time(k)[0] / max(1, d.kernel_time(k)[1]):.3f}" for k in range(1, mp + 2)))
        if d is not whole:
            d.close()
    for r, g in enumerate(gem):
        print(f"  rank {r}: count-kernel ms by size: {g}")
    print(f"world {world} ({mode}): per-rank ms/step {' '.join(f'{x:.3f}' for x in per)}  max {max(per):.3f}",
          flush=True)
