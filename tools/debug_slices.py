"""Development check: the slices of the second-largest size's count table
(obnsl_precount_sharded), summed over the ranks, against the whole table.
    python tools/debug_slices.py <workload> <world>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openbnsl_amd import engine

wl, world = sys.argv[1], int(sys.argv[2])
spec = bench.WORKLOADS[wl]
codes, card, _ = bench.make_data(spec)
k = spec["max_parents"] + 1
whole = engine.Data(codes, card)
whole.precount(k - 1)
parts = []
for r in range(world):
    d = engine.Data(codes, card)
    for rep in range(2):
        d.pack()
        size, cells = d.precount_sharded(k, r, world)
        buf = torch.empty(cells, dtype=torch.int32, device="cuda")
        d.counts_export(size, buf.data_ptr())
        torch.cuda.synchronize()
    print("rank", r, "size", size, "cells", cells, "nonzero", int((buf != 0).sum()), flush=True)
    parts.append(buf)
    d.close()
stack = torch.stack(parts)
print("overlapping cells", int(((stack != 0).sum(dim=0) > 1).sum()))
total = stack.sum(dim=0, dtype=torch.int32)
ref = torch.empty_like(total)
whole.counts_export(k - 1, ref.data_ptr())
torch.cuda.synchronize()
bad = (total != ref).nonzero().flatten()
print("cells", total.numel(), "differ", bad.numel(), bad[:10].tolist(), total[bad[:10]].tolist(), ref[bad[:10]].tolist())
