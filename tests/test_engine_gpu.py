"""GPU parity tests: the HIP engine, called through its C ABI (ctypes), against
the committed reference outputs (tests/golden) and the plain-C oracle.

Bar: counts bit-exact; scores within 1e-9 relative (the engine evaluates lgamma
with the device math library, the reference with glibc, and sums in a different
order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-9


def rel_err(have, want):
    have = np.asarray(have, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    # relative to the score, with a floor of 1 for scores that cancel to ~0 (a
    # constant child column scores exactly 0 up to rounding of the lgamma terms)
    return float(np.max(np.abs(have - want) / np.maximum(np.abs(want), 1.0)))


@pytest.fixture(scope="module")
def engine():
    from openbnsl_amd import engine as e
    assert e.device_count() >= 1, "GPU tests need a HIP device"
    return e


@pytest.mark.parametrize("frame", ["asia5k", "mixed3k", "lexi500"])
def test_golden_counts_and_scores(engine, golden, frame):
    meta, arrays = golden
    rec = meta[frame]
    codes = arrays[frame + "_codes"]
    card = np.array(rec["num_of_values"], dtype=np.uint32)
    d = engine.Data(codes, card)
    for tb in rec["tables"]:
        vs = tb["var_ids"]
        cnt = d.count(vs)
        got = {str(i): int(c) for i, c in enumerate(cnt) if c}
        assert got == tb["counts"], (frame, vs)
        if not vs:
            continue
        for ess, hexes in tb["scores"].items():
            want = [float.fromhex(h) for h in hexes]
            have = [engine.local_score(cnt, card[vs], ci, engine.BDEU, float(ess))
                    for ci in range(len(vs))]
            assert rel_err(have, want) <= REL, (frame, vs, ess, have, want)
            if len(vs) <= engine.MAX_SUBSET:
                have2 = d.score_subsets([vs], engine.BDEU, float(ess))[0]
                assert rel_err(have2, want) <= REL
    d.close()


def test_asia_full_score_table_matches_reference(engine, golden):
    from openbnsl_amd import synth
    meta, _ = golden
    rec = meta["asia100k_full"]
    codes, card, _ = synth.asia(100_000, seed=0)
    d = engine.Data(codes, card)
    for k in (1, 2, 3):
        want = np.array([[float.fromhex(x) for x in row] for row in rec["scores_by_size"][str(k)]])
        have = d.score_all(k)
        assert have.shape == want.shape
        assert rel_err(have, want) <= REL
    d.close()


@pytest.mark.parametrize("cards,N,seed", [
    ([2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 2], 20_000, 0),   # mixed, N % 32 == 0
    ([2, 2, 3, 3, 4, 4, 2, 3], 1_001, 1),                 # ragged tail
    ([4, 4, 4, 4, 4, 4], 33, 2),                          # almost empty tables
    ([3, 2, 5, 7, 2, 3, 6], 5_000, 3),                    # cardinalities above 4
])
def test_score_all_matches_oracle(engine, cards, N, seed):
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn(cards, N, seed=seed, min_prob=0.02)
    d = engine.Data(codes, card)
    n = len(cards)
    for k in range(1, min(n, 6) + 1):
        want = oracle.score_all(codes, card, k, oracle.BDEU, 1.0, nthreads=4)
        have = d.score_all(k)
        assert rel_err(have, want) <= REL, (cards, N, k)
        subs = oracle.enumerate_subsets(n, k)
        for t in range(0, len(subs), max(1, len(subs) // 7)):
            assert np.array_equal(d.count(subs[t]), oracle.count(codes, card, subs[t]))
    for k in (2, 3):
        want = oracle.score_all(codes, card, k, oracle.BIC, 1.0, nthreads=4)
        assert rel_err(d.score_all(k, engine.BIC), want) <= REL
        want = oracle.score_all(codes, card, k, oracle.BDEU, 10.0, nthreads=4)
        assert rel_err(d.score_all(k, engine.BDEU, 10.0), want) <= REL
    d.close()


def test_constant_column_and_single_row(engine):
    import oracle
    codes = np.array([[0, 0, 0, 0, 0], [0, 1, 0, 1, 1], [2, 1, 0, 0, 2]], dtype=np.uint8)
    card = np.array([1, 2, 3], dtype=np.uint32)
    d = engine.Data(codes, card)
    for k in (1, 2, 3):
        assert rel_err(d.score_all(k), oracle.score_all(codes, card, k)) <= REL
    assert d.count([]).tolist() == [5]
    d.close()
    one = np.array([[1], [0]], dtype=np.uint8)
    d = engine.Data(one, [2, 1])
    assert d.count([0, 1]).tolist() == [0, 1]
    d.close()


def test_score_subsets_matches_score_all(engine):
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2, 4], 4_000, seed=5)
    d = engine.Data(codes, card)
    for k in (1, 3, 4):
        subs = oracle.enumerate_subsets(7, k)
        full = d.score_all(k)
        pick = subs[::3]
        # the list form sums each score's two terms directly, the batch form takes
        # differences of per-table sums: same terms, different grouping
        assert rel_err(d.score_subsets(pick), full[::3]) <= 1e-12
    d.close()


def test_sharded_scores_cover_everything_once(engine):
    torch = pytest.importorskip("torch")
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3], 6_000, seed=6)
    d = engine.Data(codes, card)
    for k in (1, 2, 3, 4):
        want = oracle.score_all(codes, card, k, nthreads=4)
        T = d.num_subsets(k)
        filled = torch.zeros(T * k, dtype=torch.float64, device="cuda")
        for nsh in (2, 3, 8, 32):     # 32: more shards than row groups (empty shards)
            parts = []
            for sh in range(nsh):
                buf = torch.full((T * k,), float("-inf"), dtype=torch.float64, device="cuda")
                d.score_all_device(k, buf.data_ptr(), shard=sh, num_shards=nsh)
                torch.cuda.synchronize()
                parts.append(buf)
            stack = torch.stack(parts)
            assert int((stack > float("-inf")).sum(dim=0).min()) == 1   # each entry exactly once
            assert int((stack > float("-inf")).sum(dim=0).max()) == 1
            filled = stack.max(dim=0).values
        assert rel_err(filled.cpu().numpy().reshape(T, k), want) <= REL
    d.close()


def test_best_parents_matches_bruteforce(engine):
    torch = pytest.importorskip("torch")
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4], 8_000, seed=7)
    n, mp = 9, 3
    d = engine.Data(codes, card)
    bufs, want_best = [], {}
    for k in range(1, mp + 2):
        T = d.num_subsets(k)
        buf = torch.empty(T * k, dtype=torch.float64, device="cuda")
        d.score_all_device(k, buf.data_ptr())
        bufs.append(buf)
    torch.cuda.synchronize()
    sc, sz, rk = d.best_parents([b.data_ptr() for b in bufs], mp)
    for k in range(1, mp + 2):
        subs = oracle.enumerate_subsets(n, k)
        tab = bufs[k - 1].cpu().numpy().reshape(-1, k)
        for t, s in enumerate(subs):
            for i, v in enumerate(s):
                ps = [int(x) for x in s if x != v]
                key = (tab[t, i], -(k - 1), -engine.rank(ps))
                if int(v) not in want_best or key > want_best[int(v)][0]:
                    want_best[int(v)] = (key, ps)
    for v in range(n):
        key, ps = want_best[v]
        assert sc[v] == key[0]
        assert int(sz[v]) == len(ps)
        assert engine.unrank(int(rk[v]), int(sz[v])) == ps
    d.close()


def test_error_behaviour(engine, golden):
    meta, arrays = golden
    codes = arrays["asia5k_codes"]
    d = engine.Data(codes, [2] * 8)
    # reference messages: tests/golden/reference_outputs.json "errors"
    with pytest.raises(ValueError, match=meta["errors"]["unsorted_var_ids"][1]):
        d.count([3, 1])
    with pytest.raises(ValueError, match="Unsupported score type"):
        d.score_all(2, engine.K2)
    with pytest.raises(ValueError, match="Unsupported score type"):
        engine.local_score([1, 2, 3, 4], [2, 2], 0, engine.AIC)
    with pytest.raises(ValueError, match=meta["errors"]["child_not_in_table"][1]):
        engine.local_score([1, 2, 3, 4], [2, 2], 2)
    with pytest.raises(IndexError):
        d.count([1, 99])
    d.close()
    # a code that is not below the declared cardinality is refused at upload
    bad = codes.copy()
    bad[5, 4321] = 2
    with pytest.raises(ValueError, match="variable 5 has a code"):
        engine.Data(bad, [2] * 8)


def test_alarm_shape_full_size_properties(engine):
    """The benchmark configuration at full size (37 vars, 1e6 rows, subsets up to 4):
    sampled tables bit-exact vs the oracle, sampled scores within tolerance,
    and the size-independent invariant sum(counts) == N
    (reference: pytest/base/test_contingency_table.py:18-27)."""
    import oracle
    from openbnsl_amd import synth
    codes, card, _ = synth.alarm_shaped(1_000_000, seed=0)
    d = engine.Data(codes, card)
    rng = np.random.default_rng(0)
    scores4 = d.score_all(4)
    assert scores4.shape == (66045, 4)
    assert np.all(np.isfinite(scores4)) and np.all(scores4 < 0)
    subs = oracle.enumerate_subsets(37, 4)
    for t in rng.choice(len(subs), size=12, replace=False):
        vs = subs[t]
        cnt = d.count(vs)
        ocnt = oracle.count(codes, card, vs)
        assert np.array_equal(cnt, ocnt)
        assert int(cnt.sum()) == 1_000_000
        want = [oracle.local_score(ocnt, card[vs], ci) for ci in range(4)]
        assert rel_err(scores4[t], want) <= REL
    for k in (1, 2, 3):
        have = d.score_all(k)
        subs = oracle.enumerate_subsets(37, k)
        for t in rng.choice(len(subs), size=6, replace=False):
            ocnt = oracle.count(codes, card, subs[t])
            want = [oracle.local_score(ocnt, card[subs[t]], ci) for ci in range(k)]
            assert rel_err(have[t], want) <= REL
    d.close()


def test_high_cardinality_tables_use_the_histogram_path(engine):
    """Tables above 4096 cells are counted by per-row LDS histograms (and, above
    16384 cells, by global atomics): same bars as everywhere else."""
    torch = pytest.importorskip("torch")
    import oracle
    from openbnsl_amd import synth
    cards = [12, 9, 7, 11, 5, 8, 3]
    codes, card, _, _ = synth.random_bn(cards, 20_003, seed=12, min_prob=0.01)
    d = engine.Data(codes, card)
    for k in (3, 4, 5):          # 1188 (bit-plane path), 9504 (LDS histogram), 66528 cells (global)
        want = oracle.score_all(codes, card, k, nthreads=4)
        have = d.score_all(k)
        assert rel_err(have, want) <= REL, k
    k = 4
    T = d.num_subsets(k)
    parts = []
    for sh in range(3):
        buf = torch.full((T * k,), float("-inf"), dtype=torch.float64, device="cuda")
        d.score_all_device(k, buf.data_ptr(), shard=sh, num_shards=3)
        torch.cuda.synchronize()
        parts.append(buf)
    stack = torch.stack(parts)
    assert int((stack > float("-inf")).sum(dim=0).min()) == 1
    assert int((stack > float("-inf")).sum(dim=0).max()) == 1
    want = oracle.score_all(codes, card, k, nthreads=4)
    assert rel_err(stack.max(dim=0).values.cpu().numpy().reshape(T, k), want) <= REL
    vs = [0, 1, 3, 5]
    assert np.array_equal(d.count(vs), oracle.count(codes, card, vs))
    d.close()


def test_hot_path_device_result_matches_host_result(engine):
    """The asynchronous pipeline (device-resident per-node bests) against the
    synchronous obnsl_best_parents and a brute-force arg-max over oracle scores."""
    import oracle
    from openbnsl_amd import synth
    from openbnsl_amd.pipeline import HotPath
    from helpers import best_parents_reference
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3], 12_000, seed=31)
    hot = HotPath(codes, card, 3)
    assert hot.data.kernel_time(4) == (0.0, 0)       # count launches are not timed unless asked for
    hot.step()
    assert hot.data.kernel_time(4) == (0.0, 0)
    hot.data.kernel_timing(1)
    for _ in range(3):
        dev = hot.step()
    sc, sz, rk = hot.to_host(dev)
    sc2, sz2, rk2 = hot.data.best_parents(hot.ptrs, 3)
    assert np.array_equal(sc, sc2) and np.array_equal(sz, sz2) and np.array_equal(rk, rk2)
    tables = [oracle.score_all(codes, card, k, nthreads=4) for k in range(1, 5)]
    wsc, wsz, wrk = best_parents_reference(tables, 10)
    assert rel_err(sc, wsc) <= REL and np.array_equal(sz, wsz) and np.array_equal(rk, wrk)
    ms, launches = hot.data.kernel_time(4)
    assert launches == 3 and ms > 0
    hot.close()


def test_largest_subset_sizes_and_tiny_problems(engine):
    """Subset sizes 7 and 8 (4+3 and 4+4 variable splits of the bit-GEMM) and
    degenerate data sets (one and two variables)."""
    import oracle
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn([2, 2, 3, 2, 2, 3, 2, 2, 2], 3_000, seed=8, min_prob=0.05)
    d = engine.Data(codes, card)
    for k in (7, 8):
        want = oracle.score_all(codes, card, k, nthreads=4)
        assert rel_err(d.score_all(k), want) <= REL, k
    # nine variables: one table over all of them, through the per-row histogram kernel
    assert d.count_path(9) == "histogram"
    assert rel_err(d.score_all(9), oracle.score_all(codes, card, 9)) <= REL
    with pytest.raises(ValueError):
        d.score_all(10)             # more variables than the data set has
    d.close()
    for cards, N in (([3], 100), ([2, 4], 257)):
        codes, card, _, _ = synth.random_bn(cards, N, seed=1, n_arcs=len(cards) - 1)
        d = engine.Data(codes, card)
        for k in range(1, len(cards) + 1):
            assert rel_err(d.score_all(k), oracle.score_all(codes, card, k)) <= REL
        parents, total = d.exhaustive_search(len(cards) - 1)
        want, wtotal = oracle.exhaustive_search(codes, card, len(cards) - 1)
        assert abs(total - wtotal) <= 1e-9 * abs(wtotal)
        d.close()


def test_combine_kernel_applies_the_tie_rule(engine):
    """obnsl_combine_best_device against the numpy statement of the same rule
    (highest score, then fewer parents, then lower rank), with ties forced."""
    import torch
    from openbnsl_amd import distributed as obd
    rng = np.random.default_rng(2)
    n, world = 8, 5
    codes = rng.integers(0, 2, size=(n, 256)).astype(np.uint8)
    d = engine.Data(codes, [2] * n)
    parts = np.zeros((world, n, 3))
    parts[:, :, 0] = -rng.integers(1, 4, size=(world, n)).astype(np.float64)   # many equal scores
    parts[:, :, 1] = rng.integers(0, 3, size=(world, n))
    parts[:, :, 2] = rng.integers(0, 5, size=(world, n))
    parts[1, 3, 0] = -np.inf                                                     # a shard without this node
    dev = torch.from_numpy(parts).cuda()
    out = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    d.combine_best_device(dev.data_ptr(), world, out.data_ptr())
    torch.cuda.synchronize()
    sc, sz, rk = obd.combine_best(parts)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, 0], sc) and np.array_equal(got[:, 1], sz) and np.array_equal(got[:, 2], rk)
    d.close()


def test_randomised_sweep_against_the_oracle(engine):
    """Seeded random shapes: 1..6 states per variable (constant columns
    included), row counts around the 32-row word and 2048-row padding edges,
    BDeu with random ess and BIC, every subset size the data set allows up to 5;
    exhaustive search on the same data."""
    import oracle
    rng = np.random.default_rng(2024)
    edge_rows = [1, 2, 31, 32, 33, 63, 65, 2047, 2048, 2049, 4097]
    for case in range(14):
        n = int(rng.integers(2, 8))
        card = rng.integers(1, 7, size=n).astype(np.uint32)
        N = int(edge_rows[case]) if case < len(edge_rows) else int(rng.integers(100, 6000))
        codes = np.stack([rng.integers(0, c, size=N) for c in card]).astype(np.uint8)
        d = engine.Data(codes, card)
        ess = float(rng.choice([0.5, 1.0, 4.0]))
        for k in range(1, min(n, 5) + 1):
            for kind, okind, e in ((engine.BDEU, oracle.BDEU, ess), (engine.BIC, oracle.BIC, 1.0)):
                want = oracle.score_all(codes, card, k, okind, e, nthreads=2)
                have = d.score_all(k, kind, e)
                # BIC terms are N ln N sized and cancel: compare on the scale of the terms
                scale = np.maximum(np.abs(want), N * np.log(max(N, 2)))
                assert float(np.max(np.abs(have - want) / scale)) <= REL, (case, card, N, k, kind)
        mp = int(rng.integers(0, n))
        parents, total = d.exhaustive_search(min(mp, 4))
        want_parents, want_total = oracle.exhaustive_search(codes, card, min(mp, 4))
        assert abs(total - want_total) <= 1e-9 * max(abs(want_total), 1.0), (case, card, N, mp)
        d.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_second_size_sums_to_the_whole_table(engine, world):
    """obnsl_precount_sharded: each "rank" (its own data set object here) counts
    a slice of the second-largest size; the slices are disjoint, their sum is the
    whole table, scoring refuses an unsummed table, and after the import every
    rank's scores equal the unsharded ones."""
    torch = pytest.importorskip("torch")
    import oracle
    from openbnsl_amd import synth
    cards = [2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 3, 4, 2, 3, 3, 2, 4, 2, 3, 4, 2, 3, 2, 4, 3, 2, 2, 3, 4]
    codes, card, _, _ = synth.random_bn(cards, 4_000, seed=21)
    n, kmax = len(cards), 4
    ranks = [engine.Data(codes, card) for _ in range(world)]
    parts = []
    for r, d in enumerate(ranks):
        size, cells = d.precount_sharded(kmax, r, world)
        assert size == kmax - 1 and cells > 0
        buf = torch.empty(cells, dtype=torch.int32, device="cuda")
        d.counts_export(size, buf.data_ptr())
        torch.cuda.synchronize()
        parts.append(buf)
        out = torch.empty(d.num_subsets(2) * 2, dtype=torch.float64, device="cuda")
        d.score_all_device(1, torch.empty(n, dtype=torch.float64, device="cuda").data_ptr(),
                           shard=r, num_shards=world)          # sizes below the slice are whole
        with pytest.raises(ValueError, match="one rank's slice"):
            d.score_all_device(kmax, torch.empty(d.num_subsets(kmax) * kmax, dtype=torch.float64,
                                                 device="cuda").data_ptr(), shard=r, num_shards=world)
    stack = torch.stack(parts)
    assert int(((stack != 0).sum(dim=0) > 1).sum()) == 0        # disjoint slices
    total = stack.sum(dim=0, dtype=torch.int32)
    whole = engine.Data(codes, card)
    whole.precount(kmax - 1)
    ref = torch.empty_like(total)
    whole.counts_export(kmax - 1, ref.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(total, ref)
    want = oracle.score_all(codes, card, kmax, nthreads=4)
    T = ranks[0].num_subsets(kmax)
    outs = []
    for r, d in enumerate(ranks):
        d.counts_import(kmax - 1, total.data_ptr())
        buf = torch.full((T * kmax,), float("-inf"), dtype=torch.float64, device="cuda")
        for k in range(1, kmax):
            d.score_all_device(k, torch.empty(d.num_subsets(k) * k, dtype=torch.float64,
                                              device="cuda").data_ptr(), shard=r, num_shards=world)
        d.score_all_device(kmax, buf.data_ptr(), shard=r, num_shards=world)
        torch.cuda.synchronize()
        outs.append(buf)
    st = torch.stack(outs)
    assert int((st > float("-inf")).sum(dim=0).min()) == 1 and int((st > float("-inf")).sum(dim=0).max()) == 1
    assert rel_err(st.max(dim=0).values.cpu().numpy().reshape(T, kmax), want) <= REL
    for d in ranks + [whole]:
        d.close()


def test_unsummed_slice_is_forgotten_after_a_new_pack(engine):
    """A sharded precount whose table is never summed blocks scoring for those
    row bitmaps only: after the next pack the data set counts and scores as usual."""
    import oracle
    from openbnsl_amd import synth
    cards = [2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 3, 4, 2, 3, 3, 2, 4, 2, 3, 4, 2]
    codes, card, _, _ = synth.random_bn(cards, 3_000, seed=22)
    d = engine.Data(codes, card)
    size, cells = d.precount_sharded(4, 1, 2)
    assert size == 3 and cells == d.counts_cells(3)
    with pytest.raises(ValueError, match="one rank's slice"):
        d.score_all(4)
    d.pack()
    assert rel_err(d.score_all(4), oracle.score_all(codes, card, 4, nthreads=4)) <= REL
    d.pack()
    d.precount(4)
    assert rel_err(d.score_all(3), oracle.score_all(codes, card, 3, nthreads=4)) <= REL
    with pytest.raises(ValueError, match="no count table"):
        d.counts_cells(7)
    d.close()


def test_extreme_priors_and_a_very_wide_variable(engine):
    """BDeu with tiny and huge equivalent sample sizes (lgamma arguments from
    1e-6 up to 1e6), and a 200-state variable next to binary ones (tables of
    200..1600 cells: LDS expansion, histogram path, single-table calls)."""
    import oracle
    rng = np.random.default_rng(31)
    N = 50_000
    card = np.array([2, 200, 3, 2, 4], dtype=np.uint32)
    codes = np.stack([rng.integers(0, c, size=N) for c in card]).astype(np.uint8)
    codes[1, :4000] = 7                        # a heavy cell among many nearly empty ones
    d = engine.Data(codes, card)
    for ess in (1e-3, 1.0, 1e3):
        for k in (1, 2, 3):
            want = oracle.score_all(codes, card, k, oracle.BDEU, ess, nthreads=4)
            have = d.score_all(k, engine.BDEU, ess)
            assert rel_err(have, want) <= REL, (ess, k)
    for vs in ([1], [0, 1], [1, 2, 4], [0, 1, 2, 3]):
        assert np.array_equal(d.count(vs), oracle.count(codes, card, vs))
    sub = np.array([[0, 1, 4], [1, 2, 3]], dtype=np.uint32)
    have = d.score_subsets(sub, engine.BDEU, 1e-3)
    for t, vs in enumerate(sub):
        cnt = oracle.count(codes, card, vs)
        want = [oracle.local_score(cnt, card[vs], ci, oracle.BDEU, 1e-3) for ci in range(3)]
        assert rel_err(have[t], want) <= REL
    d.close()


def test_subsets_larger_than_eight_variables(engine):
    """The reference takes any sorted var_ids (include/base/contingency_table.h:34-56)
    and any max_parents <= n - 1 (src/structure_learning/exhaustive_search.cpp:18-20).
    Sizes above 8 leave the bit-plane contraction for the per-row histogram kernel:
    counts bit-exact, scores and the search's optimum as the oracle's."""
    import oracle
    from openbnsl_amd import synth
    cards = [2, 3, 2, 2, 4, 2, 3, 2, 2, 2, 3, 2]
    codes, card, _, _ = synth.random_bn(cards, 5_000, seed=31)
    d = engine.Data(codes, card)
    n = len(cards)
    rng = np.random.default_rng(3)
    for k in (9, 10, 11, 12):
        vs = sorted(rng.choice(n, size=k, replace=False).tolist())
        want = oracle.count(codes, card, vs)
        have = d.count(vs)
        assert np.array_equal(have, want) and int(have.sum()) == 5_000
        for ci in (0, k // 2, k - 1):
            w = oracle.local_score(want, card[vs], ci)
            assert abs(engine.local_score(have, card[vs], ci) - w) <= REL * abs(w)
        assert d.count_path(k) == "histogram"
        assert rel_err(d.score_all(k), oracle.score_all(codes, card, k, nthreads=4)) <= REL, k
        subs = np.array([sorted(rng.choice(n, size=k, replace=False).tolist()) for _ in range(3)], dtype=np.uint32)
        got = d.score_subsets(subs)
        for t, s in enumerate(subs):
            cnt = oracle.count(codes, card, s)
            assert rel_err(got[t], [oracle.local_score(cnt, card[s], ci) for ci in range(k)]) <= REL
    # the search with no bound on the number of parents (max_parents = n - 1)
    parents, total = d.exhaustive_search(n - 1)
    _, want_total = oracle.exhaustive_search(codes, card, n - 1)
    assert abs(total - want_total) <= REL * abs(want_total)
    # per-node best parent sets over every subset size
    import torch
    tabs = [d.score_all(k) for k in range(1, n + 1)]
    dev = [torch.from_numpy(t.reshape(-1)).cuda() for t in tabs]
    sc, sz, rk = d.best_parents([t.data_ptr() for t in dev], n - 1)
    best = np.full(n, -np.inf)
    for k, t in enumerate(tabs, start=1):
        np.maximum.at(best, oracle.enumerate_subsets(n, k).reshape(-1), t.reshape(-1))
    assert np.array_equal(sc, best)
    d.close()


@pytest.mark.timeout(600)
def test_counts_above_two_to_the_24_stay_exact(engine):
    """The matrix-core kernel accumulates in f32, exact below 2^24 per workgroup;
    the host cuts the contraction so that no workgroup sees more samples than that,
    and the partial counts meet in uint32 counters.  17.8 million rows with cells
    that hold more than 2^24 of them (one above 2^24 + 1, which f32 cannot hold)."""
    import oracle
    N = (1 << 24) + 1_000_003
    rng = np.random.default_rng(24)
    codes = np.empty((5, N), dtype=np.uint8)
    codes[0] = (rng.random(N) < 0.9995).astype(np.uint8)        # state 1 almost always
    codes[1] = (rng.random(N) < 0.9990).astype(np.uint8)
    codes[2] = rng.integers(0, 3, size=N, dtype=np.uint8)
    codes[3] = (rng.random(N) < 0.5).astype(np.uint8)
    codes[4] = np.minimum(rng.integers(0, 40, size=N), 2).astype(np.uint8)  # state 2 for 95 % of the rows
    card = np.array([2, 2, 3, 2, 3], dtype=np.uint32)
    d = engine.Data(codes, card)
    assert d.count_path(2) == "mfma"
    red = d.reduced_table([0, 1])                               # the batch kernel's own counter
    assert int(red[0, 0]) == int(np.count_nonzero((codes[0] == 1) & (codes[1] == 1))) > (1 << 24) + 1
    for vs in ([0], [0, 1], [0, 1, 4], [0, 1, 2, 4], [0, 1, 2, 3, 4]):
        want = oracle.count(codes, card, vs)
        assert int(want.sum()) == N
        assert np.array_equal(d.full_table_from_reduced(vs), want), vs
        assert np.array_equal(d.count(vs), want), vs
    for k in (1, 2, 3):
        assert rel_err(d.score_all(k), oracle.score_all(codes, card, k, nthreads=8)) <= REL
    d.close()
