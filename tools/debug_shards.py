"""Development check: the sharded top-size score tables of a benchmark workload, merged
over the shards, against the unsharded table (one process, one GPU).
    python tools/debug_shards.py <workload> <world>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from openbnsl_amd import engine

wl, world = sys.argv[1], int(sys.argv[2])
spec = bench.WORKLOADS[wl]
codes, card, _ = bench.make_data(spec)
kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
ktop = spec["max_parents"] + 1
one = engine.Data(codes, card)
for k in range(1, ktop + 1):
    T = one.num_subsets(k)
    ref = torch.empty(T * k, dtype=torch.float64, device="cuda")
    one.score_all_device(k, ref.data_ptr(), kind=kind)
    torch.cuda.synchronize()
    merged = torch.full((T * k,), float("-inf"), dtype=torch.float64, device="cuda")
    for r in range(world):
        d = engine.Data(codes, card)
        for rep in range(2):
            buf = torch.full((T * k,), float("-inf"), dtype=torch.float64, device="cuda")
            d.pack()
            d.precount(ktop, r, world)
            for j in range(1, k):   # the lower sizes first, as a pass does
                tmp = torch.empty(d.num_subsets(j) * j, dtype=torch.float64, device="cuda")
                d.score_all_device(j, tmp.data_ptr(), kind=kind, shard=r, num_shards=world)
            d.score_all_device(k, buf.data_ptr(), kind=kind, shard=r, num_shards=world)
            torch.cuda.synchronize()
        fin = torch.isfinite(buf)
        both = fin & torch.isfinite(merged)
        print("k", k, "shard", r, "scores", int(fin.sum()), "overlap", int(both.sum()),
              "wrong", int((buf[fin] != ref[fin]).sum()), flush=True)
        merged = torch.where(fin, buf, merged)
        d.close()
    print("k", k, "covered", int(torch.isfinite(merged).sum()), "of", T * k, "equal", bool((merged == ref).all()))
