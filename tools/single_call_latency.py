"""Development helper: latency of the one-table entry points (the reference's
own calling pattern: one ContingencyTable + one calculate_local_score at a time)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openbnsl_amd import engine, synth

for N in (100_000, 1_000_000, 10_000_000):
    codes, card, _ = synth.alarm_shaped(N, seed=0)
    d = engine.Data(codes, card)
    for vs in ([3], [3, 17], [3, 17, 30], [3, 17, 30, 36], [1, 3, 17, 30, 36]):
        d.count(vs)
        t0 = time.perf_counter()
        for _ in range(20):
            cnt = d.count(vs)
        t_count = (time.perf_counter() - t0) / 20
        cd = card[vs]
        engine.local_score(cnt, cd, 0)
        t0 = time.perf_counter()
        for _ in range(20):
            engine.local_score(cnt, cd, 0)
        t_score = (time.perf_counter() - t0) / 20
        sub = np.array([vs], dtype=np.uint32)
        d.score_subsets(sub)
        t0 = time.perf_counter()
        for _ in range(20):
            d.score_subsets(sub)
        t_sub = (time.perf_counter() - t0) / 20
        many = np.array([sorted(np.random.default_rng(i).choice(37, size=len(vs), replace=False)) for i in range(256)], dtype=np.uint32)
        d.score_subsets(many)
        t0 = time.perf_counter()
        d.score_subsets(many)
        t_many = time.perf_counter() - t0
        print(f"N={N:>8} k={len(vs)} cells={cnt.size:>5}: count {t_count*1e6:7.0f} us  local_score {t_score*1e6:6.0f} us  "
              f"score_subsets(1) {t_sub*1e6:7.0f} us  score_subsets(256) {t_many*1e3:7.2f} ms", flush=True)
    d.close()
