"""CPU: the plain-C oracle against the reference's own outputs (tests/golden,
produced by the reference's C++ compiled from source) and against the
known-answer counts of the reference's gtest files."""
import hashlib

import numpy as np
import pytest

import oracle
from openbnsl_amd import synth


# reference: gtest/utils/next_combination.cpp:19-58
@pytest.mark.parametrize("n,m,expect", [(5, 3, 10), (10, 3, 120), (31, 3, 4495), (63, 3, 39711),
                                        (0, 0, 1), (31, 0, 1), (63, 0, 1), (31, 31, 1),
                                        (63, 63, 1)])
def test_next_combmask_counts(n, m, expect):
    assert oracle.lib.oracle_comb_count(n, m) == expect


# reference: gtest/utils/comb2vec.cpp:7-56 (set bits, ascending)
@pytest.mark.parametrize("mask,expect", [(0, []), (1, [0]), (0b1011, [0, 1, 3]),
                                         (1 << 63, [63]), ((1 << 40) | 2, [1, 40])])
def test_combmask2vec(mask, expect):
    import ctypes
    out = (ctypes.c_uint32 * 64)()
    k = oracle.lib.oracle_combmask2vec(mask, out)
    assert [out[i] for i in range(k)] == expect


def test_gosper_order_matches_index_enumeration():
    # exhaustive_search.cpp:51-52,129: masks ascending from the lowest `size` bits
    import ctypes
    for n, k in [(4, 2), (7, 3), (9, 4)]:
        subs = oracle.enumerate_subsets(n, k).tolist()
        comb = ctypes.c_uint64((1 << k) - 1)
        seen = []
        while True:
            seen.append([j for j in range(n) if (comb.value >> j) & 1])
            if not oracle.lib.oracle_next_combmask(ctypes.byref(comb), n):
                break
        assert seen == subs
    assert oracle.enumerate_subsets(4, 2).tolist() == [[0, 1], [0, 2], [1, 2], [0, 3], [1, 3], [2, 3]]


@pytest.mark.parametrize("frame", ["asia5k", "mixed3k", "lexi500"])
def test_counts_and_bdeu_match_reference(golden, frame):
    meta, arrays = golden
    rec = meta[frame]
    codes = arrays[frame + "_codes"]
    card = np.array(rec["num_of_values"], dtype=np.uint32)
    for tb in rec["tables"]:
        vs = tb["var_ids"]
        cnt = oracle.count(codes, card, vs)
        got = {str(i): int(c) for i, c in enumerate(cnt) if c}
        assert got == tb["counts"], (frame, vs)          # bit-exact
        assert int(cnt.sum()) == codes.shape[1]         # test_contingency_table.py:18-27
        for ess, hexes in tb["scores"].items():
            for ci, hx in enumerate(hexes):
                want = float.fromhex(hx)
                have = oracle.local_score(cnt, card[vs], ci, oracle.BDEU, float(ess))
                assert have == want, (frame, vs, ci, ess)  # same libm, same order: exact


def test_asia_full_table_matches_reference(golden):
    meta, _ = golden
    rec = meta["asia100k_full"]
    codes, card, names = synth.asia(100_000, seed=0)
    assert hashlib.sha256(codes.tobytes()).hexdigest() == rec["sha256_codes"]
    assert names == rec["columns"]
    total = 0
    for k in (1, 2, 3):
        want = np.array([[float.fromhex(x) for x in row] for row in rec["scores_by_size"][str(k)]])
        have = oracle.score_all(codes, card, k, oracle.BDEU, 1.0, nthreads=2)
        assert np.array_equal(have, want)
        total += want.size
    assert total == 232


def test_exhaustive_search_asia_edges(golden):
    meta, _ = golden
    rec = meta["asia100k_full"]
    codes, card, names = synth.asia(100_000, seed=0)
    want = sorted(tuple(e) for e in rec["exhaustive_search_edges"])
    for mp in (2, 3):
        parents, _ = oracle.exhaustive_search(codes, card, mp)
        have = sorted((names[p], names[v]) for v, ps in enumerate(parents) for p in ps)
        assert have == want


def test_exhaustive_search_argument_errors():
    codes, card, _ = synth.asia(1000, seed=1)
    with pytest.raises(ValueError, match="max_parents"):
        oracle.exhaustive_search(codes, card, 8)


def test_count_unsorted_raises():
    codes, card, _ = synth.asia(1000, seed=1)
    with pytest.raises(ValueError, match="sorted"):
        oracle.count(codes, card, [3, 1])


def test_bic_matches_numpy_restatement():
    # BIC has no reference implementation (parity unpinned upstream); pin the C
    # oracle against an independent numpy statement of the same formula.
    codes, card, _, _ = synth.random_bn([2, 3, 4, 3, 2], 4000, seed=2, n_arcs=6)
    for vs, ci in [([0, 1, 2], 1), ([1, 3], 0), ([2], 0), ([0, 2, 3, 4], 3)]:
        cnt = oracle.count(codes, card, vs).astype(np.float64)
        shape = [int(card[v]) for v in vs]
        t = cnt.reshape(shape)
        nij = t.sum(axis=ci, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            ll = np.nansum(np.where(t > 0, t * np.log(t / nij), 0.0))
        q = np.prod(shape) / shape[ci]
        want = ll - 0.5 * np.log(codes.shape[1]) * q * (shape[ci] - 1)
        have = oracle.local_score(cnt.astype(np.uint64), card[vs], ci, oracle.BIC)
        assert abs(have - want) <= 1e-9 * abs(want)


@pytest.mark.parametrize("cards,N,mp,seed", [([2, 3, 2, 2], 400, 3, 1), ([3, 2, 4, 2, 3], 600, 2, 2),
                                             ([2, 2, 2, 3, 2], 300, 4, 3)])
def test_exhaustive_search_optimum_against_all_dags(cards, N, mp, seed):
    """The search restated in oracle.c (src/structure_learning/exhaustive_search.cpp:47-148
    cannot be compiled here) must return a highest-scoring DAG: every DAG on the
    4 or 5 nodes with at most max_parents parents per node is enumerated (as
    parent-set choices that admit a topological order) and scored from the
    oracle's local scores, which are pinned to the reference bit for bit."""
    import itertools
    from openbnsl_amd import synth
    codes, card, _, _ = synth.random_bn(cards, N, seed=seed)
    n = len(cards)
    local = {}
    for v in range(n):
        others = [u for u in range(n) if u != v]
        for k in range(0, mp + 1):
            for ps in itertools.combinations(others, k):
                vs = sorted(ps + (v,))
                cnt = oracle.count(codes, card, vs)
                local[(v, ps)] = oracle.local_score(cnt, card[vs], vs.index(v))
    best = -np.inf
    choices = [[ps for (v, ps) in local if v == node] for node in range(n)]

    def acyclic(assign):
        left = {v: set(ps) for v, ps in enumerate(assign)}
        while left:
            free = [v for v, ps in left.items() if not ps]
            if not free:
                return False
            for v in free:
                del left[v]
            for ps in left.values():
                ps.difference_update(free)
        return True

    for assign in itertools.product(*choices):
        if acyclic(assign):
            best = max(best, sum(local[(v, ps)] for v, ps in enumerate(assign)))
    parents, total = oracle.exhaustive_search(codes, card, mp)
    assert abs(total - best) <= 1e-9 * abs(best)
    got = sum(local[(v, tuple(ps))] for v, ps in enumerate(parents))
    assert abs(got - best) <= 1e-9 * abs(best) and acyclic([tuple(p) for p in parents])
