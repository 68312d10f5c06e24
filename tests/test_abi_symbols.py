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


def test_shipped_library_has_no_timing_probes_and_exports_only_the_abi():
    """The timing probes of the matrix-core kernel skip parts of it (wrong counts by
    construction): they exist only in tools/bin/libobnsl_engine_probes.so.  The shipped
    library contains no probe switch and exports nothing but obnsl_* symbols."""
    import subprocess
    from openbnsl_amd import engine
    data = open(engine.LIB_PATH, "rb").read()
    assert b"OBNSL_PROBE" not in data and b"obnsl_probe_" not in data
    out = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], capture_output=True, text=True, check=True)
    names = [line.split()[-1] for line in out.stdout.splitlines() if line.strip()]
    assert names and all(n.startswith("obnsl_") for n in names), [n for n in names if not n.startswith("obnsl_")][:5]
