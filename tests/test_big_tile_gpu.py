"""GPU: forms of the matrix-core contraction that the default launches of small data
sets do not take: 256 x 256 tiles, and split launches that add into the count table
itself instead of tile images.

Launches of fewer than 16384 tiles take 128 x 128 tiles, so only very wide data
sets reach the large form by themselves (the 128-variable configuration's
four-variable tables do, tests/test_configs_gpu.py).  Here a fresh process forces
it (OBNSL_MFMA_TILE=256, read once per process) for every subset size it serves
-- three, four and five variables: the (2,1), (2,2) and (3,2) row / column splits
-- on data with ragged tile edges, and its scores and count tables must equal
those of the default form bit for bit (the same integer counts, then the same
scorer), which the other tests pin to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from openbnsl_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
from openbnsl_amd import engine, synth
cards = [2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 3, 4, 2, 3, 3, 2, 4, 2, 3, 4, 2, 3, 2, 4, 3]
codes, card, _, _ = synth.random_bn(cards, 70_001, seed=77)
d = engine.Data(codes, card)
out = {}
for k in (1, 2, 3, 4, 5):
    out["s%d" % k] = d.score_all(k)
    assert d.count_path(k) == "mfma"
rng = np.random.default_rng(5)
for k in (3, 4, 5):
    for i in range(6):
        vs = sorted(rng.choice(len(cards), size=k, replace=False).tolist())
        out["t%d_%d" % (k, i)] = d.full_table_from_reduced(vs)
np.savez(sys.argv[1], **out)
"""


def run_child(path, tile, no_image=False):
    env = dict(os.environ)
    env.pop("OBNSL_MFMA_TILE", None)
    env.pop("OBNSL_NO_IMAGE", None)
    if tile:
        env["OBNSL_MFMA_TILE"] = str(tile)
    if no_image:
        env["OBNSL_NO_IMAGE"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD, path], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return np.load(path)


def test_large_tiles_equal_default_tiles(tmp_path):
    import oracle
    big = run_child(str(tmp_path / "big.npz"), 256)
    std = run_child(str(tmp_path / "std.npz"), 0)
    assert sorted(big.files) == sorted(std.files)
    for name in std.files:
        assert np.array_equal(big[name], std[name]), name
    # and the default form against the oracle on the same data (counts of the sampled tables)
    cards = [2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2, 3, 4, 2, 3, 3, 2, 4, 2, 3, 4, 2, 3, 2, 4, 3]
    codes, card, _, _ = synth.random_bn(cards, 70_001, seed=77)
    rng = np.random.default_rng(5)
    for k in (3, 4, 5):
        for i in range(6):
            vs = sorted(rng.choice(len(cards), size=k, replace=False).tolist())
            want = oracle.count(codes, card, vs)
            assert int(want.sum()) == 70_001
            assert np.array_equal(big["t%d_%d" % (k, i)], want), (k, vs)


def test_split_launches_without_tile_images_equal_default(tmp_path):
    """Launches whose images would exceed the budget (or that are forced so,
    OBNSL_NO_IMAGE) add their partial counts into the count table with scattered
    atomics: same tables, same scores."""
    direct = run_child(str(tmp_path / "direct.npz"), 0, no_image=True)
    std = run_child(str(tmp_path / "std.npz"), 0)
    assert sorted(direct.files) == sorted(std.files)
    for name in std.files:
        assert np.array_equal(direct[name], std[name]), name
