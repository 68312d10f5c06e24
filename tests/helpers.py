"""Test-only helpers (may use the CPU checker under oracle/)."""
import numpy as np

from openbnsl_amd import engine


def best_parents_reference(score_tables, n):
    """Brute-force per-node best over host score tables [C(n,k), k] (tests)."""
    import oracle
    best = {}
    for k, tab in enumerate(score_tables, start=1):
        subs = oracle.enumerate_subsets(n, k)
        for t, s in enumerate(subs):
            for i, v in enumerate(s):
                ps = [int(x) for x in s if x != v]
                key = (tab[t, i], -(k - 1), -engine.rank(ps))
                if int(v) not in best or key > best[int(v)]:
                    best[int(v)] = key
    sc = np.array([best[v][0] for v in range(n)])
    sz = np.array([-best[v][1] for v in range(n)], dtype=np.uint32)
    rk = np.array([-best[v][2] for v in range(n)], dtype=np.uint64)
    return sc, sz, rk
