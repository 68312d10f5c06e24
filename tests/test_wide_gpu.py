"""GPU: data sets wider than 64 variables (two benchmark configurations are 70 and
128 variables wide; the reference's uint64 subset masks stop at 64,
src/structure_learning/exhaustive_search.cpp:12,22-23).  Reduced row counts so
that the oracle finishes in seconds; subset sizes as in those configs."""
import numpy as np
import pytest

from openbnsl_amd import engine, synth

pytestmark = pytest.mark.gpu
REL = 1e-9


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture(scope="module")
def hepar():
    codes, card, _ = synth.hepar2_shaped(20_000, seed=1)
    d = engine.Data(codes, card)
    yield codes, card, d
    d.close()


def test_70_variables_all_scores_up_to_3(hepar):
    import oracle
    codes, card, d = hepar
    assert codes.shape[0] == 70
    for k in (1, 2, 3):
        want = oracle.score_all(codes, card, k, nthreads=8)
        have = d.score_all(k)
        assert have.shape == want.shape and rel_err(have, want) <= REL, k


def test_70_variables_sampled_tables_of_4_and_best_parents(hepar):
    import torch
    import oracle
    from helpers import best_parents_reference
    codes, card, d = hepar
    rng = np.random.default_rng(4)
    scores4 = d.score_all(4)
    assert scores4.shape == (engine.binomial(70, 4), 4)
    for _ in range(25):
        vs = sorted(rng.choice(70, size=4, replace=False).tolist())
        cnt = oracle.count(codes, card, vs)
        assert np.array_equal(d.count(vs), cnt) and int(cnt.sum()) == 20_000
        want = [oracle.local_score(cnt, card[vs], ci) for ci in range(4)]
        assert rel_err(scores4[engine.rank(vs)], want) <= REL
    # per-node best parent sets over <= 2 parents against a brute-force scan
    tabs = [d.score_all(k) for k in (1, 2, 3)]
    dev = [torch.from_numpy(t.reshape(-1)).cuda() for t in tabs]
    sc, sz, rk = d.best_parents([t.data_ptr() for t in dev], 2)
    wsc, wsz, wrk = best_parents_reference(tabs, 70)
    assert np.array_equal(sc, wsc) and np.array_equal(sz, wsz) and np.array_equal(rk, wrk)


def test_128_four_state_variables():
    import oracle
    codes, card, _ = synth.uniform(128, 4, 6_000, seed=2)
    d = engine.Data(codes, card)
    for k in (1, 2):
        assert rel_err(d.score_all(k), oracle.score_all(codes, card, k, nthreads=8)) <= REL
    rng = np.random.default_rng(5)
    subs = np.array([sorted(rng.choice(128, size=4, replace=False).tolist()) for _ in range(40)],
                    dtype=np.uint32)
    have = d.score_subsets(subs)
    for t, vs in enumerate(subs):
        cnt = oracle.count(codes, card, vs)
        want = [oracle.local_score(cnt, card[vs], ci) for ci in range(4)]
        assert rel_err(have[t], want) <= REL
        assert np.array_equal(d.count(vs), cnt)
    # the search needs the reference's n <= 64 (its message), annealing does not
    with pytest.raises(ValueError, match=r"num_of_vars must be in \[0, 64\]"):
        d.exhaustive_search(2)
    parents, total = d.simulated_annealing(2, 3000, seed=1, num_chains=2, is_deterministic=True)
    assert len(parents) == 128 and all(len(p) <= 2 for p in parents) and total < 0
    want = 0.0
    for v, ps in enumerate(parents):
        vs = sorted(ps + [v])
        cnt = oracle.count(codes, card, vs)
        want += oracle.local_score(cnt, card[vs], vs.index(v))
    assert abs(total - want) <= 1e-9 * abs(want)
    d.close()
