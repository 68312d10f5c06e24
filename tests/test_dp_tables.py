"""The subset dynamic program of exhaustive_search
(src/structure_learning/exhaustive_search.cpp:75-148) on GIVEN score tables:
the oracle's restatement (reference visiting order, Gosper) against the host
program of the drop-in module (ascending masks) and, on the GPU, against the
device kernels -- on tie-heavy tables (scores drawn from a handful of values, so
the `>=` of :82 and the strict `>` / "first child wins" of :90-94 decide most
entries) the three must agree edge for edge and in the total, bit for bit."""
import numpy as np
import pytest

from openbnsl_amd import engine


def tie_heavy_tables(n, mp, seed, levels):
    rng = np.random.default_rng(seed)
    vals = -np.arange(1, levels + 1, dtype=np.float64) * 0.5      # exact in binary: sums stay exact
    return [vals[rng.integers(0, levels, size=(engine.binomial(n, k), k))] for k in range(1, mp + 2)]


CASES = [(4, 3, 2), (5, 2, 2), (6, 5, 3), (7, 3, 2), (8, 2, 4), (9, 4, 3), (10, 3, 2), (11, 0, 2), (12, 3, 5)]


def test_oracle_tables_search_equals_its_search_on_data():
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 3, 2, 4, 3, 2, 3], 4_000, seed=3)
    for mp in (0, 1, 3, 6):
        tabs = [oracle.score_all(codes, card, k) for k in range(1, mp + 2)]
        assert oracle.exhaustive_search_tables(7, mp, tabs) == oracle.exhaustive_search(codes, card, mp)


@pytest.mark.parametrize("n,mp,levels", CASES)
def test_host_program_equals_the_oracle_on_tie_heavy_tables(n, mp, levels):
    import oracle
    import openbnsllib
    from openbnsl_amd import graphs
    for seed in range(3):
        tabs = tie_heavy_tables(n, mp, seed, levels)
        want, want_total = oracle.exhaustive_search_tables(n, mp, tabs)
        g, total = openbnsllib.structure_learning.search_from_tables(n, mp, [t.ravel().tolist() for t in tabs])
        assert graphs.parent_lists(g) == want and total == want_total, (n, mp, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("n,mp,levels", CASES + [(16, 3, 3), (18, 2, 2)])
def test_device_program_equals_the_oracle_on_tie_heavy_tables(n, mp, levels):
    import torch
    import oracle
    d = engine.Data(np.zeros((n, 64), dtype=np.uint8) + (np.arange(64) % 2).astype(np.uint8), [2] * n)
    for seed in range(3):
        tabs = tie_heavy_tables(n, mp, seed, levels)
        want, want_total = oracle.exhaustive_search_tables(n, mp, tabs)
        dev = [torch.from_numpy(t.reshape(-1).copy()).cuda() for t in tabs]
        parents, total = d.exhaustive_search_tables([t.data_ptr() for t in dev], mp)
        assert parents == want and total == want_total, (n, mp, seed)
    d.close()
