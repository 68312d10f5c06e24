"""CPU: the C-ABI library loads and exports every symbol include/obnsl_engine.h
declares; host-only helpers behave; compute calls fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "obnsl_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(obnsl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from openbnsl_amd import engine
    names = declared_symbols()
    assert len(names) >= 15
    raw = ctypes.CDLL(engine.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert sorted(engine.SIGNATURES) == names
    assert engine.lib.obnsl_abi_version() == 1


def test_rank_unrank_roundtrip():
    from openbnsl_amd import engine
    import oracle
    for n, k in [(8, 3), (37, 4), (70, 2)]:
        subs = oracle.enumerate_subsets(n, k)
        step = max(1, len(subs) // 200)
        for r in range(0, len(subs), step):
            assert engine.unrank(r, k) == subs[r].tolist()
            assert engine.rank(subs[r].tolist()) == r
    assert engine.binomial(37, 4) == 66045
    assert engine.binomial(128, 4) == 10668000
    assert engine.binomial(127, 3) == 333375     # gtest/utils/next_combination.cpp:39


def test_no_gpu_means_loud_failure():
    from openbnsl_amd import engine
    if engine.device_count() > 0:
        pytest.skip("a GPU is present")
    codes = np.zeros((3, 64), dtype=np.uint8)
    with pytest.raises(RuntimeError, match="no HIP device"):
        engine.Data(codes, [1, 1, 1])
    with pytest.raises(RuntimeError, match="no HIP device"):
        engine.local_score([3, 4], [2], 0)
