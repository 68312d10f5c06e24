"""GPU: the drop-in Python module, used the way the reference's own tests use
`openbnsllib` (reference: pytest/base/test_contingency_table.py,
pytest/score/test_local_score.py, pytest/structure_learning/test_exhaustive_search.py)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ob():
    from openbnsl_amd import openbnsllib
    assert openbnsllib.hip_enabled
    return openbnsllib


@pytest.mark.parametrize("frame", ["asia5k", "mixed3k", "lexi500"])
def test_contingency_table_and_local_score(ob, golden, frame):
    import pandas as pd
    meta, arrays = golden
    rec = meta[frame]
    codes = arrays[frame + "_codes"]
    df = pd.DataFrame({nm: np.array(rec["val_idx2str"][v], dtype=object)[codes[v]]
                       for v, nm in enumerate(rec["columns"])})
    w = ob.base.DataframeWrapper(df)
    for tb in rec["tables"]:
        vs = tb["var_ids"]
        ct = ob.base.ContingencyTable(vs, w)
        assert ct.var_ids == vs and ct.cardinalities == tb["cardinalities"]
        assert {str(k): v for k, v in ct.counts.items()} == tb["counts"]
        # reference: pytest/base/test_contingency_table.py:18-27
        assert sum(ct.counts.values()) == w.num_of_datapoints
        assert len(ct.counts) <= int(np.prod(tb["cardinalities"])) if vs else True
        for ess, hexes in tb["scores"].items():
            for ci, ch in enumerate(vs):
                want = float.fromhex(hexes[ci])
                have = ob.score.calculate_local_score(ch, [v for v in vs if v != ch], ct,
                                                      ob.score.BDeu(float(ess)))
                assert abs(have - want) <= 1e-9 * abs(want)
                assert abs(have - want) < 1e-6          # the reference's own bar, test_local_score.py:57
    ct = ob.base.ContingencyTable([0, 1], w)
    with pytest.raises(ValueError, match="child_var not present in contingency table"):
        ob.score.calculate_local_score(3, [0], ct, ob.score.BDeu(1.0))
    for st in (ob.score.K2(), ob.score.AIC()):
        with pytest.raises(ValueError, match="Unsupported score type"):
            ob.score.calculate_local_score(0, [1], ct, st)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_local_score_random_parent_sets(ob, seed):
    """Same protocol as pytest/score/test_local_score.py:20-57 (random child,
    1..5 random parents, N = 1e5) on ALARM-shaped synthetic data; expectation
    from the oracle instead of pgmpy."""
    import oracle
    from openbnsl_amd import synth
    random.seed(seed)
    codes, card, names = synth.alarm_shaped(100_000, seed=seed)
    w = ob.base.DataframeWrapper(synth.to_dataframe(codes, names))
    assert w.num_of_values == card.tolist()
    n = w.num_of_vars
    child = random.randint(0, n - 1)
    cand = [i for i in range(n) if i != child]
    parents = random.sample(cand, random.randint(1, 5))
    vs = sorted(parents + [child])
    ct = ob.base.ContingencyTable(vs, w)
    have = ob.score.calculate_local_score(child, parents, ct, ob.score.BDeu(1.0))
    want = oracle.local_score(oracle.count(codes, card, vs), card[vs], vs.index(child))
    assert abs(have - want) < 1e-6
    assert abs(have - want) <= 1e-9 * abs(want)


def test_exhaustive_search_asia_recovers_structure(ob, golden):
    """The ASIA configuration (8 vars, 1e5 rows, BDeu): the same edge list the
    reference returned on this data set, for max_parents 2 and 3."""
    from openbnsl_amd import synth
    meta, _ = golden
    want = sorted(tuple(e) for e in meta["asia100k_full"]["exhaustive_search_edges"])
    codes, card, names = synth.asia(100_000, seed=0)
    w = ob.base.DataframeWrapper(synth.to_dataframe(codes, names, labels=[["no", "yes"]] * 8))
    for mp in (2, 3):
        g = ob.structure_learning.exhaustive_search(w, ob.score.BDeu(1.0), max_parents=mp)
        have = sorted((names[j], names[v]) for v in range(8) for j in range(8) if g.has_edge(j, v))
        assert have == want


@pytest.mark.parametrize("cards,N,mp,kind", [
    ([2, 3, 4, 2, 3, 2, 4, 3, 2, 3, 2, 4], 20_000, 3, "bdeu"),
    ([2, 3, 2, 2, 3, 2, 2], 1_500, 2, "bic"),
    ([2, 3, 2, 3, 2], 2_000, 4, "bdeu"),
])
def test_exhaustive_search_matches_oracle(ob, cards, N, mp, kind):
    import oracle
    from openbnsl_amd import graphs, synth
    codes, card, names, _ = synth.random_bn(cards, N, seed=len(cards))
    w = ob.base.DataframeWrapper(synth.to_dataframe(codes, names))
    st = ob.score.BDeu(1.0) if kind == "bdeu" else ob.score.BIC()
    g = ob.structure_learning.exhaustive_search(w, st, mp)
    n = len(cards)
    okind = oracle.BDEU if kind == "bdeu" else oracle.BIC
    have = graphs.parent_lists(g)
    want, want_total = oracle.exhaustive_search(codes, card, mp, okind)
    # The emitted DAG is one member of the optimal Markov equivalence class; which
    # member depends on last-bit differences between score-equivalent sets (the
    # device lgamma is not glibc's), so compare what the reference's own test
    # compares -- the equivalence class (SHD 0 between CPDAGs) -- plus optimality:
    # the DAG's total score, evaluated by the oracle, equals the oracle's optimum.
    assert graphs.is_acyclic(have)
    assert all(len(ps) <= mp for ps in have)
    total = 0.0
    for v, ps in enumerate(have):
        vs = sorted(ps + [v])
        total += oracle.local_score(oracle.count(codes, card, vs), card[vs], vs.index(v), okind)
    assert abs(total - want_total) <= 1e-9 * abs(want_total)
    if kind == "bdeu":      # BDeu is score equivalent; BIC optima need not be unique classes
        assert graphs.markov_equivalent(have, want)


@pytest.mark.parametrize("case", ["mixed12", "ties7", "bic9", "mp0", "n16"])
def test_device_dp_equals_host_dp_on_the_same_scores(ob, case):
    """obnsl_exhaustive_search (scores + subset DP + back-track on the GPU) must
    return exactly the DAG and total that the host restatement of the reference's
    DP (search_from_tables) returns on the engine's own score tables: same
    arithmetic, same tie rules."""
    from openbnsl_amd import engine, graphs, synth
    kind, mp = engine.BDEU, 3
    if case == "mixed12":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2, 4, 3, 2, 3, 2, 4], 20_000, seed=12)
    elif case == "ties7":       # duplicated columns: exact score ties everywhere
        base, card0, _, _ = synth.random_bn([2, 3, 2], 2_000, seed=4)
        codes = np.ascontiguousarray(np.concatenate([base, base, base[:1]]))
        card = np.concatenate([card0, card0, card0[:1]])
    elif case == "bic9":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2, 4, 3, 2], 6_000, seed=5)
        kind, mp = engine.BIC, 2
    elif case == "mp0":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2], 1_000, seed=6)
        mp = 0
    else:
        codes, card, _, _ = synth.random_bn([2, 3, 2, 2, 3, 2, 4, 2, 3, 2, 2, 3, 2, 4, 2, 3], 5_000, seed=16)
    n = codes.shape[0]
    d = engine.Data(codes, card)
    parents, total = d.exhaustive_search(mp, kind)
    tables = [d.score_all(k, kind).ravel().tolist() for k in range(1, mp + 2)]
    g, host_total = ob.structure_learning.search_from_tables(n, mp, tables)
    assert parents == graphs.parent_lists(g)
    assert total == host_total
    assert graphs.is_acyclic(parents) and all(len(p) <= mp for p in parents)
    d.close()


def test_exhaustive_search_error_paths(ob):
    from openbnsl_amd import engine, synth
    codes, card, names = synth.asia(2_000, seed=0)
    d = engine.Data(codes, card)
    with pytest.raises(ValueError, match=r"max_parents must be in \[0, num_of_vars-1\]"):
        d.exhaustive_search(8)
    with pytest.raises(ValueError, match="Unsupported score type"):
        d.exhaustive_search(2, engine.K2)
    d.close()
    wide = np.zeros((40, 64), dtype=np.uint8)
    wide[:, ::2] = 1
    d = engine.Data(wide, [2] * 40)
    with pytest.raises(ValueError, match="dynamic program"):
        d.exhaustive_search(2)
    d.close()


def test_score_tables_saved_from_the_gpu_reload_exactly(ob, tmp_path):
    """save_score_tables writes what score_tables returns; a search on the
    reloaded tables gives the graph of the search on the device."""
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 3, 2, 4, 3, 2, 3, 2, 2, 3], 5000, seed=8, n_arcs=12)
    names = [f"c{v:02d}" for v in range(10)]
    w = ob.base.DataframeWrapper.from_codes(codes, names)
    path = str(tmp_path / "t.obnsl")
    ob.structure_learning.save_score_tables(path, w, ob.score.BDeu(1.0), 3)
    ld = ob.structure_learning.load_score_tables(path)
    direct = ob.structure_learning.score_tables(w, ob.score.BDeu(1.0), 3)
    assert ld.num_of_datapoints == 5000 and ld.num_of_values == card.tolist()
    for a, b in zip(direct, ld.tables):
        assert np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))
    g1, _ = ob.structure_learning.search_from_tables(10, 3, ld.tables)
    g2 = ob.structure_learning.exhaustive_search(w, ob.score.BDeu(1.0), 3)
    assert all(g1.has_edge(p, v) == g2.has_edge(p, v) for p in range(10) for v in range(10))
