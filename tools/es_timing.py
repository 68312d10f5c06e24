"""Development helper: wall-clock of the whole exhaustive search on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openbnsl_amd import engine, synth
for n, N, mp in [(8, 100_000, 2), (16, 10_000, 3), (20, 1_000, 3), (20, 100_000, 3), (24, 100_000, 3)]:
    if n == 8:
        codes, card, _ = synth.asia(N, 0)
    else:
        cards = ([2, 3, 2, 4, 2, 3] * 5)[:n]
        codes, card, _, _ = synth.random_bn(cards, N, seed=n)
    d = engine.Data(codes, card)
    d.exhaustive_search(mp)
    t0 = time.perf_counter()
    parents, total = d.exhaustive_search(mp)
    dt = time.perf_counter() - t0
    print(f"n={n} N={N} mp={mp}: exhaustive_search {dt*1e3:.2f} ms  total={total:.3f} edges={sum(len(p) for p in parents)}", flush=True)
    d.close()
