"""CPU: the simulated-annealing chains of the engine on host score tables
(obnsl_anneal_tables; the tables come from the oracle here) against the graphs
the reference's own simulated_annealing returned
(src/structure_learning/simulated_annealing.cpp, outputs in tests/golden), and
against the compiled reference itself where it is present."""
import hashlib

import numpy as np
import pytest

import oracle
from openbnsl_amd import engine, synth

FRAMES = {
    "asia20k": lambda: synth.asia(20000, seed=0)[:2],
    "mixed12_3k": lambda: synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 2], 3000, seed=11,
                                          n_arcs=16)[:2],
}


@pytest.fixture(scope="module")
def frames(golden):
    meta, _ = golden
    out = {}
    for name, make in FRAMES.items():
        codes, card = make()
        want = meta["simulated_annealing"]["frames"][name]["sha256_codes"]
        assert hashlib.sha256(codes.tobytes()).hexdigest() == want, "generator drifted from the fixture"
        tables = [oracle.score_all(codes, card, k, nthreads=4) for k in range(1, 6)]
        out[name] = (codes, card, tables)
    return out


def test_chains_reproduce_the_reference_graphs(golden, frames):
    meta, _ = golden
    cases = meta["simulated_annealing"]["cases"]
    assert len(cases) >= 12
    for c in cases:
        codes, card, tables = frames[c["frame"]]
        mp = c["max_parents"]
        have, total = engine.anneal_tables(codes.shape[0], mp, tables[:mp + 1], c["max_iters"],
                                           c["init_temp"], c["cooling_rate"], c["is_deterministic"],
                                           c["seed"], c["num_chains"])
        assert have == c["parents"], c
        # the returned total is the sum of the local scores of the returned graph
        want_total = 0.0
        for v, ps in enumerate(have):
            vs = sorted(ps + [v])
            want_total += tables[len(vs) - 1][engine.rank(vs), vs.index(v)]
        assert abs(total - want_total) <= 1e-9 * abs(want_total)
        assert all(len(ps) <= mp for ps in have)


def test_against_compiled_reference_on_fresh_seeds(frames):
    ref = oracle.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    codes, card, tables = frames["mixed12_3k"]
    names = [f"c{v:02d}" for v in range(codes.shape[0])]
    df = synth.to_dataframe(codes, names, labels=[[f"s{j}" for j in range(int(c))] for c in card])
    dfw = ref.DataframeWrapper(df)
    for seed in (11, 12, 13):
        for det in (True, False):
            _, want = ref.simulated_annealing(dfw, 1.0, 3, 1500, 1.0, 0.999, det, seed, 2)
            have, _ = engine.anneal_tables(12, 3, tables[:4], 1500, 1.0, 0.999, det, seed, 2)
            assert have == [list(map(int, p)) for p in want]


def test_result_is_acyclic_and_bounded(frames):
    codes, card, tables = frames["mixed12_3k"]
    n = codes.shape[0]
    have, _ = engine.anneal_tables(n, 2, tables[:3], 5000, 5.0, 0.9999, False, 42, 3)
    assert all(len(p) <= 2 and p == sorted(p) for p in have)
    # Kahn: every node can be removed
    left = {v: set(p) for v, p in enumerate(have)}
    while left:
        free = [v for v, p in left.items() if not p]
        assert free, "cycle in the returned graph"
        for v in free:
            del left[v]
        for p in left.values():
            p.difference_update(free)


def test_argument_checks_and_edge_cases(golden, frames):
    meta, _ = golden
    codes, card, tables = frames["asia20k"]
    kind, msg = meta["errors"]["sa_max_parents_out_of_range"]
    assert kind == "ValueError"
    with pytest.raises(ValueError, match=msg):
        engine.anneal_tables(8, 8, tables, 10)
    # no iterations: the empty graph and the sum of the no-parent scores
    have, total = engine.anneal_tables(8, 2, tables[:3], 0, 1.0, 0.9995, True, 0, 1)
    assert have == [[]] * 8 and abs(total - float(tables[0].sum())) < 1e-6
    # max_parents = 0: the only graph within the bound
    have, total = engine.anneal_tables(8, 0, tables[:1], 100, 1.0, 0.9995, True, 0, 1)
    assert have == [[]] * 8 and abs(total - float(tables[0].sum())) < 1e-6
    # num_chains = 0 picks a default and still returns a valid graph
    have, _ = engine.anneal_tables(8, 2, tables[:3], 200, 1.0, 0.9995, True, 0, 0)
    assert len(have) == 8
