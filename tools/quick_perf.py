"""Development helper: device time of the counting kernel per subset size on a
benchmark workload (HIP-event timing inside the engine)."""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openbnsl_amd import engine, synth
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="alarm37_1e6_bdeu_mp3")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ks", default="")
args = ap.parse_args()
spec = bench.WORKLOADS[args.workload]
codes, card, names = bench.make_data(spec)
n, N = codes.shape
d = engine.Data(codes, card)
ks = [int(x) for x in args.ks.split(",")] if args.ks else list(range(1, spec["max_parents"] + 2))
for k in ks:
    T = d.num_subsets(k)
    buf = torch.empty(T * k, dtype=torch.float64, device="cuda")
    d.score_all_device(k, buf.data_ptr())
    torch.cuda.synchronize()
    d.kernel_timing(1)
    d.kernel_time(k, reset=True)
    for j in range(1, k):
        d.kernel_time(j, reset=True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.reps):
        d.pack()   # new data epoch: cached count tables are recomputed
        d.score_all_device(k, buf.data_ptr())
    t1.record(); torch.cuda.synchronize()
    ms, launches = d.kernel_time(k)
    lower = sum(d.kernel_time(j)[0] for j in range(1, k))
    print(f"k={k} tables={T} count-kernel {ms/args.reps:.3f} ms/call ({launches//args.reps} launches) "
          f"(+{lower/args.reps:.3f} ms lower sizes) whole call {t0.elapsed_time(t1)/args.reps:.3f} ms  alg {T*k*N/(ms/args.reps*1e-3)/1e12:.2f} TB/s", flush=True)
