"""GPU: the benchmark configurations at their own sizes.

    cfg2  37 variables (ALARM-shaped), 10^6 rows, BDeu, <= 3 parents
    cfg3  the same data, BIC, <= 4 parents (5-variable tables)
    cfg4  70 variables (HEPAR2-shaped), 10^7 rows, BDeu, <= 3 parents
    cfg5  128 four-state variables, 10^7 rows, BDeu, <= 3 parents

Every subset size of a configuration goes through obnsl_score_all (the batch
path the benchmark times); sampled tables are then checked three ways against
the oracle: the batch kernel's OWN count table (its slices of all sub-subsets,
expanded by inclusion-exclusion on the host), the one-table entry point, and
the scores.  Size-independent properties: sum(counts) == N for every sampled
table, every score finite and negative.  The checks follow the reference's
pytest/score/test_local_score.py:8-57 (random child + parent sets, score
against an independent implementation) and
pytest/base/test_contingency_table.py:18-27 (sum(counts) == N, no more cells
than prod(card))."""
import numpy as np
import pytest

from openbnsl_amd import engine, synth

pytestmark = pytest.mark.gpu
REL = 1e-9


def check_sampled(d, codes, card, k, scores, rng, samples, kind, okind, expect_path=None):
    """Sampled size-k subsets: batch-kernel table == one-table call == oracle, bit for
    bit; sum == N; scores within REL of the oracle's."""
    import oracle
    n, N = codes.shape
    if expect_path is not None:
        assert d.count_path(k) == expect_path
    worst = 0.0
    for _ in range(samples):
        vs = sorted(rng.choice(n, size=k, replace=False).tolist())
        want = oracle.count(codes, card, vs)
        assert int(want.sum()) == N and want.size == int(np.prod(card[vs]))
        if d.count_path(k) != "histogram":
            assert np.array_equal(d.full_table_from_reduced(vs), want), ("batch table differs", vs)
        assert np.array_equal(d.count(vs), want), ("single table differs", vs)
        row = scores[engine.rank(vs)]
        for ci in range(k):
            w = oracle.local_score(want, card[vs], ci, okind, 1.0)
            # BIC terms are N ln N sized and cancel: compare on the scale of the terms
            scale = max(abs(w), N * np.log(N)) if okind == oracle.BIC else abs(w)
            worst = max(worst, abs(row[ci] - w) / scale)
    assert worst <= REL, (k, worst)
    return worst


@pytest.fixture(scope="module")
def alarm():
    codes, card, _ = synth.alarm_shaped(1_000_000, seed=0)
    d = engine.Data(codes, card)
    yield codes, card, d
    d.close()


def test_cfg2_alarm_bdeu_every_size_from_the_batch_tables(alarm):
    import oracle
    codes, card, d = alarm
    rng = np.random.default_rng(11)
    for k in (1, 2, 3, 4):
        scores = d.score_all(k)
        assert scores.shape == (engine.binomial(37, k), k)
        assert np.all(np.isfinite(scores)) and np.all(scores < 0)
        check_sampled(d, codes, card, k, scores, rng, 12 if k > 1 else 8, engine.BDEU, oracle.BDEU,
                      expect_path="mfma")


@pytest.mark.timeout(600)
def test_cfg3_alarm_bic_up_to_five_variables(alarm):
    import oracle
    codes, card, d = alarm
    rng = np.random.default_rng(12)
    d.pack()                    # fresh row bitmaps: nothing of the BDeu pass is reused
    total = 0
    for k in (1, 2, 3, 4, 5):
        scores = d.score_all(k, engine.BIC)
        assert scores.shape == (engine.binomial(37, k), k)
        assert np.all(np.isfinite(scores)) and np.all(scores < 0)
        total += scores.size
        check_sampled(d, codes, card, k, scores, rng, 10 if k > 1 else 6, engine.BIC, oracle.BIC)
    assert total == 2_468_344   # (child, parent set) scores of the configuration
    # per-node best parent sets over <= 4 parents: the device reduction against numpy
    import torch
    tabs = [d.score_all(k, engine.BIC) for k in (1, 2, 3, 4, 5)]
    dev = [torch.from_numpy(t.reshape(-1)).cuda() for t in tabs]
    sc, sz, rk = d.best_parents([t.data_ptr() for t in dev], 4)
    best = np.full(37, -np.inf)
    for k, t in enumerate(tabs, start=1):
        subs = oracle.enumerate_subsets(37, k)       # row t of the table scores subs[t]'s members
        np.maximum.at(best, subs.reshape(-1), t.reshape(-1))
    assert np.array_equal(sc, best)


@pytest.mark.timeout(900)
def test_cfg4_hepar_shape_full_rows():
    import oracle
    codes, card, _ = synth.hepar2_shaped(10_000_000, seed=0)
    assert codes.shape == (70, 10_000_000)
    d = engine.Data(codes, card)
    rng = np.random.default_rng(13)
    total = 0
    for k in (1, 2, 3, 4):
        scores = d.score_all(k)
        assert scores.shape == (engine.binomial(70, k), k)
        assert np.all(np.isfinite(scores)) and np.all(scores < 0)
        total += scores.size
        check_sampled(d, codes, card, k, scores, rng, 8, engine.BDEU, oracle.BDEU)
    assert total == 3_836_700
    d.close()


@pytest.mark.timeout(600)
def test_cfg5_shape_batch_path_at_reduced_rows():
    """128 four-state variables: sizes 3 and 4 through the batch kernel (a different
    launch geometry from the narrow data sets: 73,152 rows and columns, no split of
    the contraction), 40,000 rows so that the oracle checks many tables."""
    import oracle
    codes, card, _ = synth.uniform(128, 4, 40_000, seed=5)
    d = engine.Data(codes, card)
    rng = np.random.default_rng(14)
    for k in (1, 2, 3, 4):
        scores = d.score_all(k)
        assert scores.shape == (engine.binomial(128, k), k)
        assert np.all(np.isfinite(scores)) and np.all(scores < 0)
        check_sampled(d, codes, card, k, scores, rng, 40, engine.BDEU, oracle.BDEU, expect_path="mfma")
    # first / last subsets of the colex order and the tile corners of the launch
    scores4 = d.score_all(4)
    for vs in ([0, 1, 2, 3], [124, 125, 126, 127], [0, 1, 2, 127], [0, 125, 126, 127], [31, 32, 63, 64]):
        cnt = oracle.count(codes, card, vs)
        assert np.array_equal(d.full_table_from_reduced(vs), cnt)
        want = [oracle.local_score(cnt, card[vs], ci) for ci in range(4)]
        assert np.max(np.abs(scores4[engine.rank(vs)] - want) / np.abs(want)) <= REL
    d.close()


@pytest.mark.timeout(900)
def test_cfg5_full_rows():
    """The configuration's own 10^7 rows: every size-3 table (341,376 of them) and,
    sampled, the size-4 launch (10.7 million tables, 3.4 GB of counters)."""
    import oracle
    codes, card, _ = synth.uniform(128, 4, 10_000_000, seed=0)
    d = engine.Data(codes, card)
    rng = np.random.default_rng(15)
    N = codes.shape[1]
    for k in (1, 2, 3):
        scores = d.score_all(k)
        assert np.all(np.isfinite(scores)) and np.all(scores < 0)
        check_sampled(d, codes, card, k, scores, rng, 6, engine.BDEU, oracle.BDEU, expect_path="mfma")
    # sum(counts) == N for EVERY size-3 table, from the batch kernel's own counters:
    # sum over the 27 reduced cells of R_S plus the inclusion-exclusion remainder is N by
    # construction, so the property with teeth is on the expanded tables of a sample
    for _ in range(20):
        vs = sorted(rng.choice(128, size=3, replace=False).tolist())
        full = d.full_table_from_reduced(vs)
        assert int(full.sum()) == N and np.array_equal(full, oracle.count(codes, card, vs))
    import torch
    buf = torch.empty(d.num_subsets(4) * 4, dtype=torch.float64, device="cuda")
    d.score_all_device(4, buf.data_ptr())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(buf).all()) and bool((buf < 0).all())
    scores4 = buf.view(-1, 4)
    for _ in range(6):
        vs = sorted(rng.choice(128, size=4, replace=False).tolist())
        cnt = oracle.count(codes, card, vs)
        assert np.array_equal(d.full_table_from_reduced(vs), cnt) and int(cnt.sum()) == N
        want = [oracle.local_score(cnt, card[vs], ci) for ci in range(4)]
        have = scores4[engine.rank(vs)].cpu().numpy()
        assert np.max(np.abs(have - want) / np.abs(want)) <= REL
    d.close()
