"""Container-only check (needs oracle/_ref): annealing chains on oracle score tables against the
reference's simulated_annealing over random data sets, parameters, seeds and both container modes."""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from openbnsl_amd import synth, engine
ref = oracle.ref()
rng = np.random.default_rng(77)
same = 0; total = 0; t0 = time.time()
for case in range(40):
    n = int(rng.integers(3, 11))
    cards = rng.integers(2, 5, size=n).tolist()
    N = int(rng.integers(200, 4000))
    codes, card, _, _ = synth.random_bn(cards, N, seed=int(rng.integers(1e6)), n_arcs=min(n+2, n*(n-1)//2))
    names = [f"c{v:02d}" for v in range(n)]
    df = synth.to_dataframe(codes, names, labels=[[f"s{j}" for j in range(int(c))] for c in card])
    dfw = ref.DataframeWrapper(df)
    if list(dfw.num_of_values) != card.tolist():
        continue   # an unobserved state: the reference's observed cardinality differs from the nominal one
    mpmax = min(4, n - 1)
    tables = [oracle.score_all(codes, card, k, nthreads=4) for k in range(1, mpmax + 2)]
    for rep in range(5):
        mp = int(rng.integers(1, mpmax + 1))
        iters = int(rng.integers(100, 3000))
        T0 = float(rng.choice([0.5, 1.0, 5.0, 50.0])); cool = float(rng.choice([0.99, 0.999, 0.9995]))
        det = bool(rng.integers(0, 2)); seed = int(rng.integers(0, 2**31)); chains = int(rng.integers(1, 4))
        _, want = ref.simulated_annealing(dfw, 1.0, mp, iters, T0, cool, det, seed, chains)
        have, _ = engine.anneal_tables(n, mp, tables[:mp + 1], iters, T0, cool, det, seed, chains)
        total += 1; same += have == [list(map(int, p)) for p in want]
        if have != [list(map(int, p)) for p in want]:
            print("DIFF", case, rep, n, mp, iters, T0, cool, det, seed, chains)
print(f"{same} of {total} graphs identical in {time.time()-t0:.0f} s")
