"""Run under torch.distributed.run with OBNSL_BENCH_BACKEND=gloo on a one-GPU box:
every rank runs the sharded pass of a benchmark workload (its shard of the
largest size, its slice of the second largest, the two exchanges); rank 0 then
runs the same workload unsharded and compares the per-node best parent sets."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from openbnsl_amd import engine
from openbnsl_amd.pipeline import HotPath

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="alarm37_1e6_bdeu_mp3")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(os.environ.get("OBNSL_BENCH_BACKEND", "gloo"))
spec = bench.WORKLOADS[args.workload]
codes, card, _ = bench.make_data(spec)
kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
hot = HotPath(codes, card, spec["max_parents"], kind=kind, rank=rank, world=world, device_index=0)
for _ in range(2):
    best = hot.step()
torch.cuda.synchronize()
got = HotPath.to_host(best)
covered = sum(hot.shard_scores(k) for k in range(1, spec["max_parents"] + 2))
t = torch.tensor([covered], dtype=torch.int64)
dist.all_reduce(t)
hot.close()
if rank == 0:
    one = HotPath(codes, card, spec["max_parents"], kind=kind, rank=0, world=1, device_index=0)
    want = HotPath.to_host(one.step())
    torch.cuda.synchronize()
    same = all((a == b).all() for a, b in zip(got, want))
    print(f"{args.workload} on {world} ranks: scores covered {int(t.item())} of {one.scores_per_pass}; "
          f"best parent sets identical to the unsharded pass: {same}", flush=True)
    one.close()
    assert same and int(t.item()) == one.scores_per_pass
dist.barrier()
dist.destroy_process_group()
