"""CPU: host logic of the drop-in module (no compute calls): DataframeWrapper
ingest against pandas (reference: pytest/base/test_dataframe_wrapper.py:22-52)
and against the reference's own outputs, the subset dynamic program against the
oracle's restatement of exhaustive_search, PDAG, score types, error paths."""
import numpy as np
import pandas as pd
import pytest

import oracle
from openbnsl_amd import synth

ob = pytest.importorskip("openbnsl_amd.openbnsllib")


def test_module_surface():
    assert ob.cuda_enabled is False
    assert isinstance(ob.hip_enabled, bool)
    for name in ("DataframeWrapper", "ContingencyTable", "PDAG"):
        assert hasattr(ob.base, name)
    for name in ("BDeu", "K2", "BIC", "AIC", "calculate_local_score"):
        assert hasattr(ob.score, name)
    assert hasattr(ob.structure_learning, "exhaustive_search")
    assert repr(ob.score.BDeu()) == "BDeu(1.000000)"      # bindings/submodules/score.cpp:10-12
    assert repr(ob.score.BIC()) == "BIC()"


@pytest.mark.parametrize("frame", ["asia5k", "mixed3k", "lexi500"])
def test_dataframe_wrapper_matches_reference(golden, frame):
    meta, arrays = golden
    rec = meta[frame]
    codes = arrays[frame + "_codes"]
    if rec["labels"] is None:
        raw = arrays[frame + "_raw"]
        df = pd.DataFrame({nm: raw[v].astype(str).astype(object)
                           for v, nm in enumerate(rec["columns"])})
    else:
        # rebuild the labelled frame from the reference's own coding
        df = pd.DataFrame({nm: np.array(rec["val_idx2str"][v], dtype=object)[codes[v]]
                           for v, nm in enumerate(rec["columns"])})
    w = ob.base.DataframeWrapper(df)
    assert w.num_of_vars == len(rec["columns"]) and w.num_of_datapoints == codes.shape[1]
    assert w.col_idx2str == rec["columns"]
    assert w.num_of_values == rec["num_of_values"]
    assert w.val_idx2str == rec["val_idx2str"]
    assert np.array_equal(w.codes, codes)
    for v, nm in enumerate(rec["columns"]):
        assert w.col_str2idx[nm] == v
        assert w.val_idx2str[v] == sorted(df[nm].unique())     # test_dataframe_wrapper.py:44-52
        assert w.val_str2idx[v] == {s: i for i, s in enumerate(w.val_idx2str[v])}
    cm = np.array(w.data_column_major, dtype=np.uint8)
    assert np.array_equal(cm, codes)
    if codes.shape[1] <= 600:
        assert np.array_equal(np.array(w.data_row_major, dtype=np.uint8), codes.T)


def test_dataframe_wrapper_layouts_and_dtypes():
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 12, size=(200, 3))
    names = ["a", "b", "c"]
    want = [sorted({str(x) for x in raw[:, j]}) for j in range(3)]
    # C-ordered 2-D object array (mis-decoded upstream), Fortran-ordered, and int64
    for values in (raw.astype(str).astype(object), np.asfortranarray(raw.astype(str).astype(object)), raw):
        w = ob.base.DataframeWrapper(pd.DataFrame(values, columns=names))
        assert w.val_idx2str == want
        for j in range(3):
            lab = np.array(w.val_idx2str[j])[w.codes[j]]
            assert lab.tolist() == [str(x) for x in raw[:, j]]
    w = ob.base.DataframeWrapper(pd.DataFrame({"x": [True, False, True]}))
    assert w.val_idx2str == [["False", "True"]] and w.codes.tolist() == [[1, 0, 1]]


def test_integer_frames_decode_in_parallel_like_pandas():
    """Wide integer frame (more columns than decoding threads, signed and
    unsigned item sizes, negative values): codes and labels as pandas sees them
    (reference: pytest/base/test_dataframe_wrapper.py:22-52)."""
    rng = np.random.default_rng(3)
    for dtype in (np.int8, np.uint8, np.int16, np.uint32, np.int64):
        lo = -5 if np.issubdtype(dtype, np.signedinteger) else 0
        raw = rng.integers(lo, lo + 9, size=(5000, 40)).astype(dtype)
        names = [f"v{j:02d}" for j in range(40)]
        w = ob.base.DataframeWrapper(pd.DataFrame(raw, columns=names))
        assert w.num_of_vars == 40 and w.num_of_datapoints == 5000
        for j in range(40):
            labels = sorted({str(int(x)) for x in raw[:, j]})
            assert w.val_idx2str[j] == labels and w.num_of_values[j] == len(labels)
            assert np.array(labels)[w.codes[j]].tolist() == [str(int(x)) for x in raw[:, j]]


def test_from_codes():
    rng = np.random.default_rng(4)
    codes = rng.integers(0, 3, size=(4, 1000)).astype(np.uint8)
    codes[3] = rng.integers(0, 12, size=1000)
    w = ob.base.DataframeWrapper.from_codes(codes, ["a", "b", "c", "d"])
    assert w.num_of_values == [3, 3, 3, 12]
    assert w.val_idx2str[3] == [f"{s:03d}" for s in range(12)]      # order of labels = order of codes
    assert np.array_equal(w.codes, codes)
    assert w.col_str2idx == {"a": 0, "b": 1, "c": 2, "d": 3}
    w = ob.base.DataframeWrapper.from_codes(codes[:2], ["a", "b"], [["lo", "mid", "top"], ["x", "y", "z"]])
    assert w.val_str2idx[0] == {"lo": 0, "mid": 1, "top": 2}
    with pytest.raises(ValueError, match="code out of range in column b"):
        ob.base.DataframeWrapper.from_codes(codes[:2], ["a", "b"], [["lo", "mid", "top"], ["x", "y"]])
    with pytest.raises(ValueError, match="does not match"):
        ob.base.DataframeWrapper.from_codes(codes, ["a", "b"])


def test_dataframe_wrapper_errors():
    with pytest.raises(ValueError, match="Input must be a pandas dataframe"):
        ob.base.DataframeWrapper([1, 2, 3])
    with pytest.raises(ValueError, match="Failed to cast column name to string"):
        ob.base.DataframeWrapper(pd.DataFrame({0: ["a", "b"]}))
    with pytest.raises(ValueError, match="Failed to cast value to string"):
        ob.base.DataframeWrapper(pd.DataFrame({"a": [0.5, 1.5]}))
    with pytest.raises(ValueError, match="Failed to cast value to string"):
        ob.base.DataframeWrapper(pd.DataFrame({"a": np.array(["x", 3], dtype=object)}))
    with pytest.raises(ValueError, match="more than 256 distinct"):
        ob.base.DataframeWrapper(pd.DataFrame({"a": np.arange(300)}))


def test_pdag():
    g = ob.base.PDAG(4)
    assert g.num_vars == 4 and not g.has_edge(0, 1)
    g.add_edge(0, 1)
    g.add_edge(2, 1)
    h = ob.base.PDAG(g)
    g.remove_edge(0, 1)
    assert not g.has_edge(0, 1) and g.has_edge(2, 1)
    assert h.has_edge(0, 1)
    assert g.has_edge(**{"from": 2, "to": 1})


def _edges(g, n):
    return [[j for j in range(n) if g.has_edge(j, v)] for v in range(n)]


@pytest.mark.parametrize("case", ["asia", "mixed", "ties", "bic", "mp0", "full"])
def test_search_from_tables_matches_oracle_search(case):
    kind, mp = oracle.BDEU, 2
    if case == "asia":
        codes, card, _ = synth.asia(20_000, seed=1)
        mp = 3
    elif case == "mixed":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2, 4, 3, 2, 3], 5_000, seed=3)
    elif case == "ties":      # duplicated columns: exact score ties everywhere
        base, card0, _, _ = synth.random_bn([2, 3, 2], 2_000, seed=4)
        codes = np.ascontiguousarray(np.concatenate([base, base, base[:1]]))
        card = np.concatenate([card0, card0, card0[:1]])
        mp = 3
    elif case == "bic":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 2], 3_000, seed=5)
        kind = oracle.BIC
    elif case == "mp0":
        codes, card, _, _ = synth.random_bn([2, 3, 4, 2], 1_000, seed=6)
        mp = 0
    else:                       # max_parents = n - 1: every subset is scored
        codes, card, _, _ = synth.random_bn([2, 3, 2, 3, 2], 2_000, seed=7)
        mp = 4
    n = codes.shape[0]
    tables = [oracle.score_all(codes, card, k, kind, 1.0, nthreads=4).ravel().tolist()
              for k in range(1, mp + 2)]
    g, total = ob.structure_learning.search_from_tables(n, mp, tables)
    want, want_total = oracle.exhaustive_search(codes, card, mp, kind)
    assert _edges(g, n) == want
    assert total == want_total


def test_search_argument_errors():
    codes, card, names = synth.asia(500, seed=0)
    w = ob.base.DataframeWrapper(synth.to_dataframe(codes, names))
    with pytest.raises(ValueError, match=r"max_parents must be in \[0, num_of_vars-1\]"):
        ob.structure_learning.exhaustive_search(w, ob.score.BDeu(1.0), 8)
    for st in (ob.score.K2(), ob.score.AIC()):
        with pytest.raises(ValueError, match="Unsupported score type"):
            ob.structure_learning.exhaustive_search(w, st, 2)
    big = pd.DataFrame({f"v{i:02d}": ["a", "b"] for i in range(65)})
    with pytest.raises(ValueError, match=r"num_of_vars must be in \[0, 64\]"):
        ob.structure_learning.exhaustive_search(ob.base.DataframeWrapper(big), ob.score.BDeu(), 2)
    with pytest.raises(ValueError, match="var_ids must be sorted"):
        ob.base.ContingencyTable([2, 1], w)
    with pytest.raises(IndexError):
        ob.base.ContingencyTable([1, 12], w)


def test_compute_fails_loudly_without_gpu():
    if ob.hip_enabled:
        pytest.skip("a GPU is present")
    codes, card, names = synth.asia(500, seed=0)
    w = ob.base.DataframeWrapper(synth.to_dataframe(codes, names))
    with pytest.raises(RuntimeError, match="no HIP device"):
        ob.base.ContingencyTable([0, 1], w)
    with pytest.raises(RuntimeError, match="no HIP device"):
        ob.structure_learning.exhaustive_search(w, ob.score.BDeu(1.0), 2)


def test_device_lists_are_validated(monkeypatch):
    """OPENBNSL_DEVICES / OPENBNSL_DEVICE with anything but device indices raise instead of
    silently meaning device 0 (the check runs before any GPU call)."""
    import pandas as pd
    import openbnsllib
    df = pd.DataFrame({"a": ["x", "y", "x", "y"], "b": ["u", "u", "v", "v"]})
    dfw = openbnsllib.base.DataframeWrapper(df)
    for bad in ("a,b", "1x", "0,,1", "-1"):
        monkeypatch.setenv("OPENBNSL_DEVICES", bad)
        with pytest.raises(ValueError, match="OPENBNSL_DEVICES"):
            openbnsllib.structure_learning.exhaustive_search(dfw, openbnsllib.score.BDeu(1.0), 1)
    monkeypatch.delenv("OPENBNSL_DEVICES")
    monkeypatch.setenv("OPENBNSL_DEVICE", "zero")
    with pytest.raises(ValueError, match="OPENBNSL_DEVICE"):
        openbnsllib.structure_learning.exhaustive_search(dfw, openbnsllib.score.BDeu(1.0), 1)
