import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from openbnsl_amd import engine, synth
for n, N, mp in [(26, 10_000, 3), (28, 10_000, 3)]:
    cards = ([2, 3, 2, 4, 2, 3] * 5)[:n]
    codes, card, _, _ = synth.random_bn(cards, N, seed=n)
    d = engine.Data(codes, card)
    t0 = time.perf_counter()
    parents, total = d.exhaustive_search(mp)
    dt = time.perf_counter() - t0
    print(f"n={n} N={N} mp={mp}: exhaustive_search {dt*1e3:.1f} ms  total={total:.3f} edges={sum(len(p) for p in parents)}", flush=True)
    d.close()
