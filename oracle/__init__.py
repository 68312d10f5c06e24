"""Test-only checkers for the local-score hot path.

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py may import this package.  The shipped package (openbnsl_amd)
never does.

* ``oracle.lib``  -- ctypes view of oracle/_build/liboracle.so (oracle.c, the
  plain-C restatement of the reference algorithm).
* ``oracle.ref()`` -- the reference's own C++ (oracle/_ref/openbnsl_ref*.so,
  compiled from /root/reference sources by oracle/Makefile), or None when that
  file has not been built.
"""
import ctypes
import importlib.util
import glob
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

BDEU, K2, BIC, AIC = 0, 1, 2, 3

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


def _load():
    path = os.path.join(_HERE, "_build", "liboracle.so")
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    lib.oracle_next_combmask.argtypes = [_u64p, ctypes.c_int]
    lib.oracle_next_combmask.restype = ctypes.c_int
    lib.oracle_combmask2vec.argtypes = [ctypes.c_uint64, _u32p]
    lib.oracle_combmask2vec.restype = ctypes.c_int
    lib.oracle_comb_count.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.oracle_comb_count.restype = ctypes.c_uint64
    lib.oracle_binom.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.oracle_binom.restype = ctypes.c_uint64
    lib.oracle_count.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_uint64, _u32p, _u32p,
                                 ctypes.c_int, _u64p, ctypes.c_uint64]
    lib.oracle_count.restype = ctypes.c_int
    lib.oracle_local_score.argtypes = [_u64p, _u32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double, _u64p]
    lib.oracle_local_score.restype = ctypes.c_double
    lib.oracle_enumerate.argtypes = [ctypes.c_uint32, ctypes.c_int, _u32p]
    lib.oracle_enumerate.restype = None
    lib.oracle_score_all.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_uint32, _u32p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, _f64p, ctypes.c_int]
    lib.oracle_score_all.restype = ctypes.c_int
    lib.oracle_exhaustive_search.argtypes = [_u8p, ctypes.c_uint64, ctypes.c_uint64,
                                             ctypes.c_uint32, _u32p, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_uint32, _u64p, _f64p]
    lib.oracle_exhaustive_search.restype = ctypes.c_int
    lib.oracle_exhaustive_search_tables.argtypes = [ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.POINTER(ctypes.c_void_p), _u64p, _f64p]
    lib.oracle_exhaustive_search_tables.restype = ctypes.c_int
    lib.oracle_max_threads.argtypes = []
    lib.oracle_max_threads.restype = ctypes.c_int
    return lib


lib = _load()


def _colmajor(codes):
    """codes: uint8 [n_vars, N], C-contiguous (column-major data matrix)."""
    a = np.ascontiguousarray(codes, dtype=np.uint8)
    assert a.ndim == 2
    return a


def count(codes, card, var_ids):
    """Dense joint counts of `var_ids` (ascending); uint64 array of prod(card)."""
    a = _colmajor(codes)
    card = np.ascontiguousarray(card, dtype=np.uint32)
    v = np.ascontiguousarray(var_ids, dtype=np.uint32)
    cells = int(np.prod([int(card[i]) for i in v], dtype=object)) if len(v) else 1
    out = np.zeros(cells, dtype=np.uint64)
    rc = lib.oracle_count(a.ctypes.data_as(_u8p), a.shape[1], a.shape[1],
                          card.ctypes.data_as(_u32p), v.ctypes.data_as(_u32p), len(v),
                          out.ctypes.data_as(_u64p), cells)
    if rc == -1:
        raise ValueError("var_ids must be sorted")
    if rc:
        raise RuntimeError(f"oracle_count failed: {rc}")
    return out


def local_score(counts, cards, child_idx, kind=BDEU, ess=1.0):
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    cards = np.ascontiguousarray(cards, dtype=np.uint32)
    marg = np.zeros(len(c), dtype=np.uint64)
    return float(lib.oracle_local_score(c.ctypes.data_as(_u64p), cards.ctypes.data_as(_u32p),
                                        len(cards), child_idx, kind, ess,
                                        marg.ctypes.data_as(_u64p)))


def enumerate_subsets(n, k):
    T = int(lib.oracle_binom(n, k))
    out = np.zeros((T, k), dtype=np.uint32)
    if T and k:
        lib.oracle_enumerate(n, k, out.ctypes.data_as(_u32p))
    return out


def score_all(codes, card, k, kind=BDEU, ess=1.0, nthreads=1):
    """Scores of every size-k subset in Gosper order: float64 [C(n,k), k]."""
    a = _colmajor(codes)
    card = np.ascontiguousarray(card, dtype=np.uint32)
    n = a.shape[0]
    T = int(lib.oracle_binom(n, k))
    out = np.zeros((T, k), dtype=np.float64)
    rc = lib.oracle_score_all(a.ctypes.data_as(_u8p), a.shape[1], a.shape[1], n,
                              card.ctypes.data_as(_u32p), k, kind, ess,
                              out.ctypes.data_as(_f64p), nthreads)
    if rc:
        raise RuntimeError(f"oracle_score_all failed: {rc}")
    return out


def exhaustive_search(codes, card, max_parents, kind=BDEU, ess=1.0):
    """Returns (list of parent lists per node, total score)."""
    a = _colmajor(codes)
    card = np.ascontiguousarray(card, dtype=np.uint32)
    n = a.shape[0]
    masks = np.zeros(n, dtype=np.uint64)
    total = ctypes.c_double(0.0)
    rc = lib.oracle_exhaustive_search(a.ctypes.data_as(_u8p), a.shape[1], a.shape[1], n,
                                      card.ctypes.data_as(_u32p), kind, ess, max_parents,
                                      masks.ctypes.data_as(_u64p), ctypes.byref(total))
    if rc == -1:
        raise ValueError("max_parents must be in [0, num_of_vars-1]")
    if rc == -2:
        raise ValueError("num_of_vars must be in [0, 64]")
    if rc:
        raise RuntimeError(f"oracle_exhaustive_search failed: {rc}")
    parents = [[j for j in range(n) if (int(m) >> j) & 1] for m in masks]
    return parents, total.value


def exhaustive_search_tables(n, max_parents, tables):
    """The reference's dynamic program on given score tables: tables[k-1] is float64
    [C(n,k), k] in visiting (colex) order.  Returns (parent lists, total score)."""
    keep = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
    arr = (ctypes.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
    masks = np.zeros(n, dtype=np.uint64)
    total = ctypes.c_double(0.0)
    rc = lib.oracle_exhaustive_search_tables(n, max_parents, arr, masks.ctypes.data_as(_u64p),
                                             ctypes.byref(total))
    if rc:
        raise RuntimeError(f"oracle_exhaustive_search_tables failed: {rc}")
    return [[j for j in range(n) if (int(m) >> j) & 1] for m in masks], total.value


_ref_mod = None


def ref():
    """The compiled reference module, or None if oracle/_ref was not built."""
    global _ref_mod
    if _ref_mod is not None:
        return _ref_mod
    hits = glob.glob(os.path.join(_HERE, "_ref", "openbnsl_ref*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("openbnsl_ref", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _ref_mod = mod
    return mod
