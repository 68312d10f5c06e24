"""GPU: obnsl_best_parents_device_noalloc (library-held scratch, no device
allocation) against obnsl_best_parents_device, through the C ABI, on two calls in
a row, the second on a smaller data set (the scratch keeps the first call's
contents)."""
import numpy as np
import pytest

from openbnsl_amd import engine, synth

pytestmark = pytest.mark.gpu


def _records(cards, rows, seed, max_parents):
    import torch
    codes, card, _, _ = synth.random_bn(cards, rows, seed=seed)
    d = engine.Data(codes, card)
    n = len(cards)
    tabs = [torch.empty(d.num_subsets(k) * k, dtype=torch.float64, device="cuda")
            for k in range(1, max_parents + 2)]
    for k, t in enumerate(tabs, start=1):
        d.score_all_device(k, t.data_ptr())
    ptrs = [t.data_ptr() for t in tabs]
    want = torch.full((n, 3), -1.0, dtype=torch.float64, device="cuda")
    have = torch.full((n, 3), -2.0, dtype=torch.float64, device="cuda")
    d.best_parents_device(ptrs, max_parents, want.data_ptr())
    torch.cuda.synchronize()
    d.best_parents_device_noalloc(ptrs, max_parents, have.data_ptr())
    torch.cuda.synchronize()
    want, have = want.cpu().numpy(), have.cpu().numpy()
    d.close()
    return want, have


def test_noalloc_best_parents_matches_the_allocating_entry_point():
    big = [2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 3, 4, 2, 3, 3, 2, 4, 2, 3, 4, 2, 3, 2]
    want, have = _records(big, 5_000, seed=3, max_parents=3)
    assert np.all(np.isfinite(want[:, 0]))
    assert np.array_equal(want, have)           # bit-exact: same kernels, same inputs
    # second call, smaller input: the array still holds the first call's words
    want, have = _records(big[:7], 900, seed=4, max_parents=2)
    assert np.all(np.isfinite(want[:, 0]))
    assert np.array_equal(want, have)


def test_noalloc_refuses_more_variables_than_its_scratch_holds():
    n = 4097
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 2, size=(n, 64)).astype(np.uint8)
    codes[:, 0], codes[:, 1] = 0, 1
    d = engine.Data(codes, np.full(n, 2, dtype=np.uint32))
    import torch
    t = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError, match="at most 4096 variables"):
        d.best_parents_device_noalloc([t.data_ptr()], 0, out.data_ptr())
    d.close()
