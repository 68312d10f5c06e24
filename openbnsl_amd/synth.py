"""Synthetic discrete data sets shaped like the benchmark networks.

No data set ships with the reference and its tests draw theirs from pgmpy's
example models (reference: pytest/score/test_local_score.py:22-24), which are
not available offline.  These generators produce data of the same shape:
a random DAG with bounded in-degree, Dirichlet CPTs and forward sampling.

All functions return ``(codes, card, names)`` with ``codes`` a C-contiguous
``uint8[n_vars, N]`` array (column-major data matrix: one row of the array per
variable), ``card`` the per-variable number of observed states and ``names``
the column names in index order.
"""
import numpy as np

# State-count multiset of the ALARM network (37 nodes, 46 arcs).
ALARM_CARDS = [2] * 13 + [3] * 17 + [4] * 7


def asia(N=100_000, seed=0):
    """Lauritzen-Spiegelhalter ASIA network, columns sorted by name.

    Codes follow the reference's lexicographic value coding
    (reference: src/base/dataframe_wrapper.cpp:38-51): "no" -> 0, "yes" -> 1.
    """
    rng = np.random.default_rng(seed)
    u = {k: rng.random(N) for k in ("asia", "smoke", "tub", "lung", "bronc", "xray", "dysp")}
    asia_ = u["asia"] < 0.01
    smoke = u["smoke"] < 0.5
    tub = u["tub"] < np.where(asia_, 0.05, 0.01)
    lung = u["lung"] < np.where(smoke, 0.1, 0.01)
    bronc = u["bronc"] < np.where(smoke, 0.6, 0.3)
    either = tub | lung
    xray = u["xray"] < np.where(either, 0.98, 0.05)
    dysp = u["dysp"] < np.where(bronc, np.where(either, 0.9, 0.8), np.where(either, 0.7, 0.1))
    cols = {"asia": asia_, "bronc": bronc, "dysp": dysp, "either": either,
            "lung": lung, "smoke": smoke, "tub": tub, "xray": xray}
    names = sorted(cols)
    codes = np.stack([cols[k].astype(np.uint8) for k in names])
    card = np.array([2] * 8, dtype=np.uint32)
    return np.ascontiguousarray(codes), card, names


ASIA_EDGES = [("asia", "tub"), ("smoke", "lung"), ("smoke", "bronc"), ("tub", "either"),
              ("lung", "either"), ("either", "xray"), ("either", "dysp"), ("bronc", "dysp")]


def cancer(N=100_000, seed=0):
    """The five-variable CANCER network (Korb & Nicholson): Pollution -> Cancer <-
    Smoker, Cancer -> Xray, Cancer -> Dyspnoea; columns sorted by name, values
    coded by lexicographic rank of their labels as the reference codes them.
    Returns (codes, card, names, parent lists)."""
    rng = np.random.default_rng(seed)
    u = {k: rng.random(N) for k in ("Pollution", "Smoker", "Cancer", "Xray", "Dyspnoea")}
    high = u["Pollution"] < 0.1                       # labels "high" < "low"
    smoker = u["Smoker"] < 0.3                        # "False" < "True"
    p_c = np.where(high, np.where(smoker, 0.05, 0.02), np.where(smoker, 0.03, 0.001))
    cancer_ = u["Cancer"] < p_c
    xray_pos = u["Xray"] < np.where(cancer_, 0.9, 0.2)   # "negative" < "positive"
    dysp = u["Dyspnoea"] < np.where(cancer_, 0.65, 0.3)
    cols = {"Cancer": cancer_.astype(np.uint8), "Dyspnoea": dysp.astype(np.uint8),
            "Pollution": (~high).astype(np.uint8), "Smoker": smoker.astype(np.uint8),
            "Xray": xray_pos.astype(np.uint8)}
    names = sorted(cols)
    codes = np.ascontiguousarray(np.stack([cols[k] for k in names]))
    idx = {k: i for i, k in enumerate(names)}
    parents = [[] for _ in names]
    for a, b in (("Pollution", "Cancer"), ("Smoker", "Cancer"), ("Cancer", "Xray"), ("Cancer", "Dyspnoea")):
        parents[idx[b]].append(idx[a])
    return codes, np.array([2] * 5, dtype=np.uint32), names, [sorted(p) for p in parents]


# State counts of the 20-node CHILD network (25 arcs, in-degree <= 2).
CHILD_CARDS = [2, 6, 3, 3, 4, 3, 3, 3, 2, 2, 3, 3, 2, 3, 2, 3, 3, 5, 2, 2]


def child_shaped(N=100_000, seed=0):
    """20 variables with CHILD's state counts, 25 arcs, at most two parents each.
    Returns (codes, card, names, parent lists of the generating DAG)."""
    return random_bn(CHILD_CARDS, N, seed=seed, n_arcs=25, max_indegree=2, min_prob=0.02)


def random_dag(n, n_arcs, max_indegree, rng):
    """Random DAG over 0..n-1 in index order: parents[v] subset of {0..v-1}."""
    parents = [[] for _ in range(n)]
    arcs = 0
    attempts = 0
    while arcs < n_arcs and attempts < 100 * n_arcs:
        attempts += 1
        v = int(rng.integers(1, n))
        if len(parents[v]) >= min(max_indegree, v):
            continue
        p = int(rng.integers(0, v))
        if p in parents[v]:
            continue
        parents[v].append(p)
        arcs += 1
    return [sorted(p) for p in parents]


def random_bn(cards, N, seed=0, n_arcs=None, max_indegree=3, min_prob=0.05):
    """Forward-sample N rows from a random Bayesian network with the given
    per-variable cardinalities.  Every state of every variable is observed
    (asserted), so observed cardinality == nominal."""
    rng = np.random.default_rng(seed)
    cards = [int(c) for c in cards]
    n = len(cards)
    if n_arcs is None:
        n_arcs = int(round(1.25 * n))
    parents = random_dag(n, n_arcs, max_indegree, rng)
    codes = np.zeros((n, N), dtype=np.uint8)
    for v in range(n):
        q = 1
        cfg = np.zeros(N, dtype=np.int64)
        for p in parents[v]:
            cfg = cfg * cards[p] + codes[p]
            q *= cards[p]
        cpt = rng.dirichlet(np.ones(cards[v]), size=q)
        cpt = (cpt + min_prob) / (1.0 + min_prob * cards[v])
        cum = np.cumsum(cpt, axis=1)
        u = rng.random(N)
        state = np.zeros(N, dtype=np.uint8)
        for s in range(cards[v] - 1):
            state += (u > cum[cfg, s]).astype(np.uint8)
        codes[v] = state
    card = np.array(cards, dtype=np.uint32)
    for v in range(n):
        assert len(np.unique(codes[v])) == cards[v], f"variable {v}: unobserved state"
    names = [f"v{v:03d}" for v in range(n)]
    return codes, card, names, parents


def alarm_shaped(N=1_000_000, seed=0):
    """37 variables with ALARM's state-count multiset (13x2, 17x3, 7x4), 46 arcs."""
    rng = np.random.default_rng(1000 + seed)
    cards = list(ALARM_CARDS)
    rng.shuffle(cards)
    codes, card, names, _ = random_bn(cards, N, seed=seed, n_arcs=46)
    return codes, card, names


def hepar2_shaped(N=10_000_000, seed=0):
    """70 variables, cardinalities in {2,3,4} (majority binary), 123 arcs."""
    rng = np.random.default_rng(2000 + seed)
    cards = [2] * 48 + [3] * 13 + [4] * 9
    rng.shuffle(cards)
    codes, card, names, _ = random_bn(cards, N, seed=seed, n_arcs=123)
    return codes, card, names


def uniform(n, r, N, seed=0):
    """n i.i.d. uniform r-state variables: the worst case for occupied cells."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, r, size=(n, N), dtype=np.uint8)
    card = np.full(n, r, dtype=np.uint32)
    return codes, card, [f"v{v:03d}" for v in range(n)]


def to_dataframe(codes, names, labels=None):
    """pandas DataFrame with string values, built column by column so that
    ``df.values`` is object-dtype and column-major in memory, which is what the
    reference's DataframeWrapper assumes
    (reference: src/base/dataframe_wrapper.cpp:30,42)."""
    import pandas as pd
    cols = {}
    for v, name in enumerate(names):
        if labels is not None:
            lab = np.array(labels[v], dtype=object)
            cols[name] = lab[codes[v]]
        else:
            cols[name] = codes[v].astype(str).astype(object)
    return pd.DataFrame(cols)
