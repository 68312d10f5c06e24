"""CPU: the score tables as a file (obnsl_scores_save / _info / _load) and the
host-side consumers that take reloaded tables.  Tables come from the oracle."""
import numpy as np
import pytest

import oracle
from openbnsl_amd import engine, synth

ob = pytest.importorskip("openbnsl_amd.openbnsllib")


@pytest.fixture(scope="module")
def tables():
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2, 4, 3, 2], 2000, seed=5, n_arcs=10)
    return codes, card, [oracle.score_all(codes, card, k, nthreads=2) for k in range(1, 5)]


def test_round_trip_is_exact(tables, tmp_path):
    codes, card, tabs = tables
    path = str(tmp_path / "scores.obnsl")
    engine.save_scores(path, codes.shape[1], card, engine.BDEU, 2.5, tabs)
    back = engine.load_scores(path)
    assert back["num_vars"] == 9 and back["num_rows"] == 2000 and back["max_parents"] == 3
    assert back["kind"] == engine.BDEU and back["ess"] == 2.5
    assert np.array_equal(back["card"], card)
    for a, b in zip(tabs, back["tables"]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
    # odd and even variable counts pad the header differently
    engine.save_scores(path, 7, card[:8], engine.BIC, 1.0, [oracle.score_all(codes[:8], card[:8], 1)])
    back = engine.load_scores(path)
    assert back["num_vars"] == 8 and back["kind"] == engine.BIC and back["max_parents"] == 0


def test_damaged_files_are_refused(tables, tmp_path):
    codes, card, tabs = tables
    path = tmp_path / "scores.obnsl"
    engine.save_scores(str(path), 2000, card, engine.BDEU, 1.0, tabs[:2])
    raw = path.read_bytes()
    (tmp_path / "short").write_bytes(raw[:len(raw) - 40])
    with pytest.raises(ValueError, match="truncated table"):
        engine.load_scores(str(tmp_path / "short"))
    (tmp_path / "magic").write_bytes(b"NOTSCORE" + raw[8:])
    with pytest.raises(ValueError, match="not a score-table file"):
        engine.load_scores(str(tmp_path / "magic"))
    with pytest.raises(RuntimeError, match="cannot read"):
        engine.load_scores(str(tmp_path / "absent"))
    with pytest.raises(IndexError):
        out = np.zeros(10)
        engine._check(engine.lib.obnsl_scores_load(str(path).encode(), 3,
                                                   out.ctypes.data_as(engine._f64p), 10))


def test_reloaded_tables_drive_search_and_annealing(tables, tmp_path):
    codes, card, tabs = tables
    path = str(tmp_path / "scores.obnsl")
    engine.save_scores(path, 2000, card, engine.BDEU, 1.0, tabs)
    ld = ob.structure_learning.load_score_tables(path)
    assert ld.num_of_vars == 9 and ld.max_parents == 3 and ld.num_of_values == card.tolist()
    assert repr(ld.score_type) == "BDeu(1.000000)"
    g, total = ob.structure_learning.search_from_tables(9, 3, ld.tables)
    want_parents, want_total = oracle.exhaustive_search(codes, card, 3)
    assert abs(total - want_total) <= 1e-9 * abs(want_total)
    have = [[p for p in range(9) if g.has_edge(p, v)] for v in range(9)]
    assert have == want_parents
    g = ob.structure_learning.anneal_from_tables(9, 3, ld.tables, max_iters=2000, seed=4,
                                                 is_deterministic=True, num_chains=2)
    want, _ = engine.anneal_tables(9, 3, tabs, 2000, 1.0, 0.9995, True, 4, 2)
    assert [[p for p in range(9) if g.has_edge(p, v)] for v in range(9)] == want
    with pytest.raises(ValueError, match="max_parents out of range"):
        ob.structure_learning.anneal_from_tables(9, 9, ld.tables)
