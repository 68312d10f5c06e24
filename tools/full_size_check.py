"""Parity spot check at a benchmark workload's full size: sampled tables
bit-exact against the oracle, their scores within 1e-9, sum(counts) == N, and
the per-node best parent sets of the sampled scores consistent with the full
score tables.  Takes minutes (the oracle walks 10^7 rows per table)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import bench
from openbnsl_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="hepar70_1e7_bdeu_mp3")
ap.add_argument("--samples", type=int, default=12)
args = ap.parse_args()
spec = bench.WORKLOADS[args.workload]
t0 = time.time()
codes, card, _ = bench.make_data(spec)
n, N = codes.shape
print(f"{args.workload}: {n} x {N} generated in {time.time() - t0:.0f} s", flush=True)
kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
okind = oracle.BDEU if spec["score"] == "bdeu" else oracle.BIC
d = engine.Data(codes, card)
rng = np.random.default_rng(0)
worst = 0.0
for k in range(1, spec["max_parents"] + 2):
    t0 = time.time()
    scores = d.score_all(k, kind)
    print(f"size {k}: {scores.shape[0]} tables scored in {time.time() - t0:.2f} s; all finite: "
          f"{bool(np.all(np.isfinite(scores)))}", flush=True)
    for _ in range(args.samples if k > 1 else min(args.samples, n)):
        vs = sorted(rng.choice(n, size=k, replace=False).tolist())
        want = oracle.count(codes, card, vs)
        have = d.count(vs)
        assert np.array_equal(have, want), ("counts differ", vs)
        assert int(have.sum()) == N
        for ci in range(k):
            w = oracle.local_score(want, card[vs], ci, okind, 1.0)
            h = scores[engine.rank(vs), ci]
            worst = max(worst, abs(h - w) / abs(w))
    print(f"size {k}: sampled tables bit-exact, sum(counts) == N; worst relative score error so far {worst:.2e}",
          flush=True)
assert worst <= 1e-9
print("full-size check ok")
