"""Ranks that split the ROWS (obnsl_data_set_total_rows, obnsl_precount_rows,
obnsl_counts_mark_summed(d, 0), obnsl_score_pass in shards), rehearsed inside one process:
one data set per slice on the one GPU, the ranges that hold their count tables added up
with torch in place of the all-reduce.  The merged score tables and the combined per-node
records must equal the whole data set's bit for bit (the same kernels on the same counts),
and the count tables of the slices must add up to the whole data set's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Words:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _tables(torch, d, mp):
    return [torch.full((d.num_subsets(k) * k,), float("-inf"), dtype=torch.float64, device="cuda")
            for k in range(1, mp + 2)]


def _whole(engine, torch, codes, card, mp, kind):
    d = engine.Data(codes, card)
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    tabs = _tables(torch, d, mp)
    best = torch.zeros((d.n, 3), dtype=torch.float64, device="cuda")
    d.pack()
    d.precount(mp + 1)
    d.score_pass([t.data_ptr() for t in tabs], mp, best.data_ptr(), kind=kind)
    torch.cuda.synchronize()
    counts = {k: torch.as_tensor(_Words(*d.counts_pointer(k)), device="cuda").clone() for k in range(1, mp + 2)}
    out = [t.cpu().numpy() for t in tabs], best.cpu().numpy(), counts
    d.close()
    return out


@pytest.mark.parametrize("case", ["bdeu_mp3_w2", "bdeu_mp3_w5_uneven", "bic_mp4_w3", "bdeu_mp2_w8"])
def test_row_slices_add_up_to_the_whole_data_set(case):
    import torch
    from openbnsl_amd import engine, synth
    mp = int(case.split("mp")[1][0])
    world = int(case.split("_w")[1].split("_")[0])
    kind = engine.BIC if case.startswith("bic") else engine.BDEU
    cards = ([2, 3, 4, 2, 3, 2, 4, 3, 2, 3, 2, 2, 3] * 2)[:16 if mp < 4 else 11]
    N = 150_011
    codes, card, _, _ = synth.random_bn(cards, N, seed=len(case))
    t_want, b_want, c_want = _whole(engine, torch, codes, card, mp, kind)
    if "uneven" in case:
        edges = [0, 3, 40_000, 40_001, 100_000, N]          # slices of 3 rows and of 1 row among them
    else:
        edges = [N * r // world for r in range(world + 1)]
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(world):
        d = engine.Data(codes[:, edges[r]:edges[r + 1]], card, total_rows=N)
        d.set_stream(stream)
        ranks.append(d)
    n = ranks[0].n
    for rep in range(3):        # fresh plans, cached plans, cached plans after a pack
        views = []
        for d in ranks:
            d.pack()
            ptr, words = d.precount_rows(mp + 1)
            views.append(torch.as_tensor(_Words(ptr, words), device="cuda"))
        assert len({v.numel() for v in views}) == 1             # the same layout on every rank
        total = torch.stack(views).sum(dim=0, dtype=torch.int32)  # (the all-reduce)
        tabs = _tables(torch, ranks[0], mp)
        ptrs = [t.data_ptr() for t in tabs]
        parts = torch.zeros((world, n, 3), dtype=torch.float64, device="cuda")
        for r, d in enumerate(ranks):
            if rep == 0 and r == 0:
                with pytest.raises(ValueError, match="sum it over the ranks"):
                    d.score_pass(ptrs, mp, parts[r].data_ptr(), kind=kind, shard=r, num_shards=world)
            views[r].copy_(total)
            d.counts_mark_summed(0)
            if rep == 0:
                for k in range(1, mp + 2):
                    got = torch.as_tensor(_Words(*d.counts_pointer(k)), device="cuda")
                    assert torch.equal(got, c_want[k]), (case, r, k)
            d.score_pass(ptrs, mp, parts[r].data_ptr(), kind=kind, shard=r, num_shards=world)
        combined = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        ranks[0].combine_best_device(parts.data_ptr(), world, combined.data_ptr())
        torch.cuda.synchronize()
        for k, (have, want) in enumerate(zip(tabs, t_want), start=1):
            assert np.array_equal(have.cpu().numpy(), want), (case, rep, k)   # every score, exactly once
        assert np.array_equal(combined.cpu().numpy(), b_want), (case, rep)
    for d in ranks:
        d.close()


def test_row_slices_refuse_what_they_cannot_serve():
    import torch
    from openbnsl_amd import engine, synth
    codes, card, _ = synth.uniform(9, 8, 20_000, seed=5)     # size-4 tables of 4096 cells: counted per row
    d = engine.Data(codes[:, :10_000], card, total_rows=20_000)
    with pytest.raises(ValueError, match="counted per row"):
        d.precount_rows(4)
    with pytest.raises(ValueError, match="smaller than the rows"):
        d.set_total_rows(9_999)
    tabs = _tables(torch, d, 1)
    with pytest.raises(ValueError, match="slice of the rows"):       # nothing counted, nothing summed
        d.score_pass([t.data_ptr() for t in tabs], 1)
    with pytest.raises(ValueError, match="slice of the rows"):
        d.score_all(2)
    ptr, words = d.precount_rows(2)
    assert ptr and words >= 9 * 7 + 36 * 49
    d.counts_mark_summed(0)
    d.score_pass([t.data_ptr() for t in tabs], 1)                    # (scores of half the counts: not checked)
    torch.cuda.synchronize()
    d.close()


def test_drivers_fall_back_to_splitting_the_subsets():
    """Where a size is counted per row, neither driver splits the rows, whatever it was asked:
    the pipeline object reports 'subsets', the group's results equal the single device's."""
    import os
    import torch
    from openbnsl_amd import engine, synth
    from openbnsl_amd.pipeline import HotPath
    codes, card, _ = synth.uniform(9, 8, 40_000, seed=6)     # size-4 tables of 4096 cells: counted per row
    mp = 3
    hot = HotPath(codes, card, mp, rank=1, world=2, exchange=False, sharding="rows")
    assert hot.sharding == "subsets" and hot.data.N == 40_000
    hot.close()
    one = engine.Data(codes, card)
    tabs = [one.score_all(k) for k in range(1, mp + 2)]
    dev = [torch.from_numpy(t.reshape(-1)).cuda() for t in tabs]
    want = one.best_parents([t.data_ptr() for t in dev], mp)
    one.close()
    os.environ["OBNSL_GROUP_SHARDING"] = "rows"
    try:
        g = engine.Group(codes, card, [0, 0, 0])
    finally:
        del os.environ["OBNSL_GROUP_SHARDING"]
    got = g.best_parents(mp, repack=True)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    got2 = g.best_parents(1, repack=True)                      # sizes 1 and 2 alone can split the rows
    dev2 = [torch.from_numpy(t.reshape(-1)).cuda() for t in tabs[:2]]
    one = engine.Data(codes, card)
    assert all(np.array_equal(a, b) for a, b in zip(got2, one.best_parents([t.data_ptr() for t in dev2], 1)))
    one.close()
    g.close()
