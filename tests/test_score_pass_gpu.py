"""obnsl_score_pass (every subset size in one call, the sizes sharing their scorer launches)
against one obnsl_score_all per size plus obnsl_best_parents_device: the two must write the
same score tables and the same per-node records bit for bit -- whole, in shards, for BIC,
and where a size falls back to the per-row path -- and agree with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(torch, d, mp):
    return [torch.full((d.num_subsets(k) * k,), float("-inf"), dtype=torch.float64, device="cuda")
            for k in range(1, mp + 2)]


def _both_ways(engine, torch, codes, card, mp, kind, shard, nshards):
    d = engine.Data(codes, card)
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    n = d.n
    out = []
    for fused in (False, True):
        d.pack()                                   # fresh counts for each form
        tabs = _tables(torch, d, mp)
        ptrs = [t.data_ptr() for t in tabs]
        best = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        d.precount(mp + 1, shard, nshards)
        if fused:
            # three passes: the first works every table out from the descriptors (and leaves
            # the per-table records behind), the later ones read the records
            prev = None
            for _ in range(3):
                for t in tabs:
                    t.fill_(float("-inf"))
                d.score_pass(ptrs, mp, best.data_ptr(), kind=kind, ess=1.0, shard=shard, num_shards=nshards)
                torch.cuda.synchronize()
                now = [t.cpu().numpy() for t in tabs] + [best.cpu().numpy()]
                assert prev is None or all(np.array_equal(a, b) for a, b in zip(prev, now))
                prev = now
                d.pack()
                d.precount(mp + 1, shard, nshards)
            d.score_pass(ptrs, mp, best.data_ptr(), kind=kind, ess=1.0, shard=shard, num_shards=nshards)
        else:
            for k in range(1, mp + 2):
                d.score_all_device(k, ptrs[k - 1], kind=kind, ess=1.0, shard=shard, num_shards=nshards)
            d.best_parents_device(ptrs, mp, best.data_ptr())
        torch.cuda.synchronize()
        out.append(([t.cpu().numpy() for t in tabs], best.cpu().numpy()))
    d.close()
    return out


@pytest.mark.parametrize("case", ["bdeu_mp3", "bic_mp4", "bdeu_mp2_3shards", "bic_mp3_2shards", "per_row_fallback"])
def test_score_pass_equals_per_size_calls(case):
    import torch
    import oracle
    from openbnsl_amd import engine, synth
    if case == "per_row_fallback":
        codes, card, _ = synth.uniform(9, 8, 20_000, seed=5)     # size-4 tables of 4096 cells: counted per row
        mp, kind, shards = 3, engine.BDEU, 1
    else:
        cards = ([2, 3, 4, 2, 3, 2, 4, 3, 2, 3, 2, 2, 3] * 2)[:14 if "mp4" not in case else 11]
        codes, card, _, _ = synth.random_bn(cards, 30_011, seed=len(case))
        mp = int(case.split("mp")[1][0])
        kind = engine.BIC if case.startswith("bic") else engine.BDEU
        shards = int(case.split("_")[2][0]) if "shards" in case else 1
    covered = None
    for shard in range(shards):
        (t_ref, b_ref), (t_fused, b_fused) = _both_ways(engine, torch, codes, card, mp, kind, shard, shards)
        for k, (x, y) in enumerate(zip(t_ref, t_fused), start=1):
            assert np.array_equal(x, y), (case, shard, k)
        assert np.array_equal(b_ref, b_fused), (case, shard)
        fin = [np.isfinite(t) for t in t_fused]
        covered = fin if covered is None else [c.astype(int) + f for c, f in zip(covered, fin)]
        if shard == 0 and shards == 1:
            kname = oracle.BIC if kind == engine.BIC else oracle.BDEU
            for k in range(1, mp + 2):
                want = oracle.score_all(codes, card, k, kind=kname, nthreads=4)
                have = t_fused[k - 1].reshape(-1, k)
                assert float(np.max(np.abs(have - want) / np.maximum(np.abs(want), 1.0))) <= 1e-9, (case, k)
    if shards > 1:
        for k, c in enumerate(covered, start=1):
            assert np.all(np.asarray(c) == 1), (case, k)     # every score exactly once over the shards


def test_score_pass_without_best_and_argument_errors():
    import torch
    from openbnsl_amd import engine, synth
    codes, card, _, _ = synth.random_bn([2, 3, 2, 4, 3, 2], 5_000, seed=3)
    d = engine.Data(codes, card)
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    tabs = _tables(torch, d, 2)
    d.score_pass([t.data_ptr() for t in tabs], 2, None)
    torch.cuda.synchronize()
    for k, t in enumerate(tabs, start=1):
        assert np.array_equal(t.cpu().numpy().reshape(-1, k), d.score_all(k))
    with pytest.raises(ValueError, match="max_parents must be in"):
        d.score_pass([t.data_ptr() for t in tabs], 6, None)
    with pytest.raises(ValueError, match="shard must be in"):
        d.score_pass([t.data_ptr() for t in tabs], 2, None, shard=2, num_shards=2)
    d.close()
