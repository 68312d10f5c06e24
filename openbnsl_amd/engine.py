"""ctypes view of the C ABI in include/obnsl_engine.h (libobnsl_engine.so).

This is plumbing for the benchmark and the parity tests, which call the engine
exactly as a foreign-language host would: plain pointers and sizes.  There is
no CPU fallback here: if the library is missing, import fails; if no HIP device
is present, every compute call raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OBNSL_ENGINE_LIB: another build of the same library (A/B timing of kernel versions)
LIB_PATH = os.environ.get("OBNSL_ENGINE_LIB") or os.path.join(_HERE, "lib", "libobnsl_engine.so")

BDEU, K2, BIC, AIC = 0, 1, 2, 3
MAX_SUBSET = 32

_STATUS_EXC = {1: ValueError, 2: IndexError, 3: RuntimeError, 4: RuntimeError, 5: ValueError}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/obnsl_engine.h declares
SIGNATURES = {
    "obnsl_last_error": (ctypes.c_char_p, []),
    "obnsl_abi_version": (ctypes.c_int, []),
    "obnsl_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "obnsl_data_create": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                         _u32p, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(_vp)]),
    "obnsl_data_destroy": (ctypes.c_int, [_vp]),
    "obnsl_data_shape": (ctypes.c_int, [_vp, _u64p, _u32p]),
    "obnsl_data_set_stream": (ctypes.c_int, [_vp, _vp]),
    "obnsl_data_pack": (ctypes.c_int, [_vp]),
    "obnsl_count": (ctypes.c_int, [_vp, _u32p, ctypes.c_uint32, _u64p, ctypes.c_uint64]),
    "obnsl_local_score": (ctypes.c_int, [ctypes.c_int, _u64p, _u32p, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_int, ctypes.c_double, _f64p]),
    "obnsl_score_all": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, _vp, ctypes.c_int]),
    "obnsl_precount": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "obnsl_precount_sharded": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                              _u32p, _u64p]),
    "obnsl_counts_cells": (ctypes.c_int, [_vp, ctypes.c_uint32, _u64p]),
    "obnsl_counts_pointer": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.POINTER(_vp), _u64p]),
    "obnsl_counts_mark_summed": (ctypes.c_int, [_vp, ctypes.c_uint32]),
    "obnsl_data_set_total_rows": (ctypes.c_int, [_vp, ctypes.c_uint64]),
    "obnsl_precount_rows": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64)]),
    "obnsl_counts_export": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp]),
    "obnsl_counts_import": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp]),
    "obnsl_reduced_table": (ctypes.c_int, [_vp, _u32p, ctypes.c_uint32, _u32p, ctypes.c_uint64]),
    "obnsl_num_subsets": (ctypes.c_int, [_vp, ctypes.c_uint32, _u64p]),
    "obnsl_score_subsets": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, _u32p,
                                           ctypes.c_uint32, ctypes.c_uint64, _f64p]),
    "obnsl_best_parents": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _f64p,
                                          _u32p, _u64p]),
    "obnsl_best_parents_device": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _vp]),
    "obnsl_best_parents_device_noalloc": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _vp]),
    "obnsl_count_launch_info": (ctypes.c_int, [_vp, ctypes.c_uint32, _u64p]),
    "obnsl_score_pass": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.POINTER(_vp), _vp]),
    "obnsl_combine_best_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp]),
    "obnsl_exhaustive_search": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                               _u64p, _f64p]),
    "obnsl_exhaustive_search_tables": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _u64p, _f64p]),
    "obnsl_group_host_time": (ctypes.c_int, [_vp, _f64p]),
    "obnsl_group_create": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, _u32p,
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "obnsl_group_destroy": (ctypes.c_int, [_vp]),
    "obnsl_group_info": (ctypes.c_int, [_vp, _u32p, ctypes.POINTER(ctypes.c_int)]),
    "obnsl_group_best_parents": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                                ctypes.c_int, _f64p, _u32p, _u64p]),
    "obnsl_group_score_tables": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                                ctypes.c_int, ctypes.POINTER(_vp)]),
    "obnsl_group_exhaustive_search": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                                     _u64p, _f64p]),
    "obnsl_simulated_annealing": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                                 ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                                                 ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32,
                                                 _u32p, _u32p, _f64p]),
    "obnsl_anneal_tables": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_vp),
                                           ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32,
                                           _u32p, _u32p, _f64p]),
    "obnsl_scores_save": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64, _u32p,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                         ctypes.POINTER(_vp)]),
    "obnsl_scores_info": (ctypes.c_int, [ctypes.c_char_p, _u32p, _u64p, _u32p, ctypes.c_uint32,
                                         ctypes.POINTER(ctypes.c_int), _f64p, _u32p]),
    "obnsl_scores_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, _f64p, ctypes.c_uint64]),
    "obnsl_binomial": (ctypes.c_uint64, [ctypes.c_uint32, ctypes.c_uint32]),
    "obnsl_unrank": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint32, _u32p]),
    "obnsl_rank": (ctypes.c_uint64, [_u32p, ctypes.c_uint32]),
    "obnsl_count_path": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]),
    "obnsl_stream_read": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, _vp]),
    "obnsl_kernel_time": (ctypes.c_int, [_vp, ctypes.c_uint32, _f64p, _u32p, ctypes.c_int]),
    "obnsl_kernel_timing": (ctypes.c_int, [_vp, ctypes.c_uint32]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _check(status):
    if status != 0:
        msg = lib.obnsl_last_error().decode()
        raise _STATUS_EXC.get(status, RuntimeError)(msg)


def device_count():
    c = ctypes.c_int(0)
    _check(lib.obnsl_device_count(ctypes.byref(c)))
    return c.value


def stream_read(src_ptr, nbytes, stream_handle, sink_ptr):
    """Queues one plain streaming read of device memory (calibration; the caller times it)."""
    _check(lib.obnsl_stream_read(src_ptr, nbytes, stream_handle, sink_ptr))


def binomial(n, k):
    return int(lib.obnsl_binomial(n, k))


def unrank(rank, k):
    out = (ctypes.c_uint32 * max(k, 1))()
    _check(lib.obnsl_unrank(rank, k, out))
    return [int(out[i]) for i in range(k)]


def rank(var_ids):
    a = (ctypes.c_uint32 * max(len(var_ids), 1))(*var_ids)
    return int(lib.obnsl_rank(a, len(var_ids)))


def local_score(counts, cards, child_idx, kind=BDEU, ess=1.0, device=0):
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    cd = np.ascontiguousarray(cards, dtype=np.uint32)
    out = ctypes.c_double(0.0)
    _check(lib.obnsl_local_score(device, c.ctypes.data_as(_u64p), cd.ctypes.data_as(_u32p),
                                 len(cd), child_idx, kind, ess, ctypes.byref(out)))
    return out.value


def anneal_tables(n, max_parents, tables, max_iters=10000, init_temp=1.0, cooling_rate=0.9995,
                  is_deterministic=False, seed=0, num_chains=0):
    """Simulated annealing on host score tables (no GPU): tables[k-1] is the
    float64 [C(n,k), k] table of subset size k.  Returns (parent lists, score)."""
    keep = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
    arr = (_vp * len(keep))(*[t.ctypes.data for t in keep])
    width = max(max_parents, 1)
    par = np.zeros((n, width), dtype=np.uint32)
    cnt = np.zeros(n, dtype=np.uint32)
    total = ctypes.c_double(0.0)
    _check(lib.obnsl_anneal_tables(n, max_parents, arr, max_iters, init_temp, cooling_rate,
                                   int(is_deterministic), seed, num_chains,
                                   par.ctypes.data_as(_u32p), cnt.ctypes.data_as(_u32p),
                                   ctypes.byref(total)))
    return [par[v, :cnt[v]].tolist() for v in range(n)], total.value


def save_scores(path, num_rows, card, kind, ess, tables):
    """Writes host score tables (tables[k-1]: float64 [C(n,k), k]) to `path`."""
    card = np.ascontiguousarray(card, dtype=np.uint32)
    keep = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
    arr = (_vp * len(keep))(*[t.ctypes.data for t in keep])
    _check(lib.obnsl_scores_save(os.fsencode(path), len(card), num_rows, card.ctypes.data_as(_u32p),
                                 kind, ess, len(keep) - 1, arr))


def load_scores(path):
    """dict(num_vars, num_rows, card, kind, ess, max_parents, tables) from a score-table file."""
    n, mp, kind = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0)
    rows, ess = ctypes.c_uint64(0), ctypes.c_double(0.0)
    p = os.fsencode(path)
    _check(lib.obnsl_scores_info(p, ctypes.byref(n), ctypes.byref(rows), None, 0, ctypes.byref(kind),
                                 ctypes.byref(ess), ctypes.byref(mp)))
    card = np.zeros(n.value, dtype=np.uint32)
    _check(lib.obnsl_scores_info(p, None, None, card.ctypes.data_as(_u32p), n.value, None, None, None))
    tables = []
    for k in range(1, mp.value + 2):
        t = np.zeros((binomial(n.value, k), k), dtype=np.float64)
        _check(lib.obnsl_scores_load(p, k, t.ctypes.data_as(_f64p), t.size))
        tables.append(t)
    return dict(num_vars=n.value, num_rows=rows.value, card=card, kind=kind.value, ess=ess.value,
                max_parents=mp.value, tables=tables)


class Data:
    """Device-resident data set (obnsl_data).  `codes`: uint8[n_vars, N] numpy
    array (host) or a (device_ptr, n_vars, N, col_stride) tuple.  total_rows: the rows
    of the whole data set when `codes` is one rank's slice of them (ranks that split
    the rows: precount_rows, all-reduce, counts_mark_summed(0), score_pass)."""

    def __init__(self, codes, card, device=0, total_rows=None):
        card = np.ascontiguousarray(card, dtype=np.uint32)
        self._h = _vp()
        if isinstance(codes, tuple):
            ptr, n, N, stride = codes
            on_dev = 1
        else:
            codes = np.ascontiguousarray(codes, dtype=np.uint8)
            n, N = codes.shape
            stride = N
            ptr = codes.ctypes.data
            on_dev = 0
        self.n, self.N, self.card, self.device = int(n), int(N), card, device
        _check(lib.obnsl_data_create(ptr, N, stride, n, card.ctypes.data_as(_u32p), device,
                                     on_dev, ctypes.byref(self._h)))
        if total_rows is not None:
            self.set_total_rows(total_rows)

    def set_total_rows(self, total_rows):
        _check(lib.obnsl_data_set_total_rows(self._h, int(total_rows)))

    def precount_rows(self, kmax):
        """Counts every subset size 1..kmax over this handle's rows; returns (device
        address, uint32 words) of the one range that holds all the count tables."""
        ptr = _vp()
        words = ctypes.c_uint64(0)
        _check(lib.obnsl_precount_rows(self._h, kmax, ctypes.byref(ptr), ctypes.byref(words)))
        return ptr.value, words.value

    def close(self):
        if self._h:
            lib.obnsl_data_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        _check(lib.obnsl_data_set_stream(self._h, stream_handle))

    def pack(self):
        _check(lib.obnsl_data_pack(self._h))

    def precount(self, kmax, shard=0, num_shards=1):
        _check(lib.obnsl_precount(self._h, kmax, shard, num_shards))

    def precount_sharded(self, kmax, shard, num_shards):
        """(subset size whose count table must be summed over the ranks, its number of
        uint32 counters); (0, 0) when there is nothing to sum."""
        size = ctypes.c_uint32(0)
        cells = ctypes.c_uint64(0)
        _check(lib.obnsl_precount_sharded(self._h, kmax, shard, num_shards, ctypes.byref(size),
                                          ctypes.byref(cells)))
        return size.value, cells.value

    def counts_cells(self, k):
        cells = ctypes.c_uint64(0)
        _check(lib.obnsl_counts_cells(self._h, k, ctypes.byref(cells)))
        return cells.value

    def counts_pointer(self, k):
        """(device address, number of uint32 counters) of the whole size-k count table."""
        ptr = _vp()
        cells = ctypes.c_uint64(0)
        _check(lib.obnsl_counts_pointer(self._h, k, ctypes.byref(ptr), ctypes.byref(cells)))
        return ptr.value, cells.value

    def counts_mark_summed(self, k):
        _check(lib.obnsl_counts_mark_summed(self._h, k))

    def counts_export(self, k, dst_ptr):
        _check(lib.obnsl_counts_export(self._h, k, dst_ptr))

    def counts_import(self, k, src_ptr):
        _check(lib.obnsl_counts_import(self._h, k, src_ptr))

    def reduced_table(self, var_ids):
        """The batch kernel's own counts of the subset: uint32 array shaped (r_v - 1 ...),
        entry [s_1 - 1, ..., s_k - 1] = rows with every variable v_i in state s_i >= 1."""
        v = np.ascontiguousarray(var_ids, dtype=np.uint32)
        shape = tuple(int(self.card[int(i)]) - 1 for i in v)
        out = np.zeros(max(int(np.prod(shape, dtype=np.int64)), 1), dtype=np.uint32)
        _check(lib.obnsl_reduced_table(self._h, v.ctypes.data_as(_u32p), len(v),
                                       out.ctypes.data_as(_u32p), len(out)))
        return out[:int(np.prod(shape, dtype=np.int64))].reshape(shape)

    def full_table_from_reduced(self, var_ids):
        """The dense table of the subset (keyed as count()) rebuilt on the host from the
        batch kernel's reduced tables of all its sub-subsets, by inclusion-exclusion."""
        import itertools
        v = [int(i) for i in var_ids]
        k = len(v)
        cards = [int(self.card[i]) for i in v]
        # E[e_1..e_k]: digit 0 = "any state", digit s >= 1 = state s
        E = np.zeros(cards, dtype=np.int64)
        E[(0,) * k] = self.N
        for size in range(1, k + 1):
            for pos in itertools.combinations(range(k), size):
                red = self.reduced_table([v[i] for i in pos]).astype(np.int64)
                idx = [0] * k
                for i in pos:
                    idx[i] = slice(1, None)
                E[tuple(idx)] = red
        for axis in range(k):        # "any" -> state 0, one variable at a time
            idx0 = [slice(None)] * k
            idx0[axis] = 0
            rest = [slice(None)] * k
            rest[axis] = slice(1, None)
            E[tuple(idx0)] -= E[tuple(rest)].sum(axis=axis)
        return E.reshape(-1).astype(np.uint64)

    def num_subsets(self, k):
        return binomial(self.n, k)

    def count(self, var_ids):
        v = np.ascontiguousarray(var_ids, dtype=np.uint32)
        cells = 1
        for i in v:
            if int(i) < self.n:
                cells *= int(self.card[int(i)])
        out = np.zeros(max(cells, 1), dtype=np.uint64)
        _check(lib.obnsl_count(self._h, v.ctypes.data_as(_u32p), len(v),
                               out.ctypes.data_as(_u64p), len(out)))
        return out

    def score_all(self, k, kind=BDEU, ess=1.0):
        """float64 [C(n,k), k] on the host."""
        T = self.num_subsets(k)
        out = np.zeros((T, k), dtype=np.float64)
        _check(lib.obnsl_score_all(self._h, kind, ess, k, 0, 1, out.ctypes.data, 0))
        return out

    def score_all_device(self, k, out_ptr, kind=BDEU, ess=1.0, shard=0, num_shards=1):
        """Asynchronous; out_ptr is a device pointer to C(n,k)*k doubles."""
        _check(lib.obnsl_score_all(self._h, kind, ess, k, shard, num_shards, out_ptr, 1))

    def score_subsets(self, subsets, kind=BDEU, ess=1.0):
        s = np.ascontiguousarray(subsets, dtype=np.uint32)
        m, k = s.shape
        out = np.zeros((m, k), dtype=np.float64)
        _check(lib.obnsl_score_subsets(self._h, kind, ess, s.ctypes.data_as(_u32p), k, m,
                                       out.ctypes.data_as(_f64p)))
        return out

    def best_parents(self, score_ptrs, max_parents):
        """score_ptrs[k-1]: device pointer of the size-k score table."""
        arr = (_vp * len(score_ptrs))(*score_ptrs)
        sc = np.zeros(self.n, dtype=np.float64)
        sz = np.zeros(self.n, dtype=np.uint32)
        rk = np.zeros(self.n, dtype=np.uint64)
        _check(lib.obnsl_best_parents(self._h, arr, max_parents, sc.ctypes.data_as(_f64p),
                                      sz.ctypes.data_as(_u32p), rk.ctypes.data_as(_u64p)))
        return sc, sz, rk

    def best_parents_device(self, score_ptrs, max_parents, out_ptr):
        """Asynchronous; out_ptr: device pointer to 3 * n doubles (score, size, rank per node)."""
        arr = (_vp * len(score_ptrs))(*score_ptrs)
        _check(lib.obnsl_best_parents_device(self._h, arr, max_parents, out_ptr))

    def best_parents_device_noalloc(self, score_ptrs, max_parents, out_ptr):
        """best_parents_device without any device allocation (library-held scratch)."""
        arr = (_vp * len(score_ptrs))(*score_ptrs)
        _check(lib.obnsl_best_parents_device_noalloc(self._h, arr, max_parents, out_ptr))

    def count_launch_info(self, k):
        """dict: tiles, units, workgroups, cells_computed, cells_asked, sizes_counted of the last
        matrix-core count launch timed under subset size k."""
        out = (ctypes.c_uint64 * 6)()
        _check(lib.obnsl_count_launch_info(self._h, k, out))
        return dict(zip(("tiles", "units", "workgroups", "cells_computed", "cells_asked", "sizes_counted"),
                        [int(x) for x in out]))

    def score_pass(self, score_ptrs, max_parents, best_ptr=None, kind=BDEU, ess=1.0, shard=0, num_shards=1):
        """Asynchronous: every score table of subset sizes 1..max_parents+1 (score_ptrs: device
        pointers, one per size) and, when best_ptr is given, the per-node record of
        best_parents_device (3 * n doubles on the device)."""
        arr = (_vp * len(score_ptrs))(*score_ptrs)
        _check(lib.obnsl_score_pass(self._h, kind, ess, max_parents, shard, num_shards, arr, best_ptr))

    def combine_best_device(self, parts_ptr, world, out_ptr):
        """Asynchronous; parts_ptr: device pointer to world * 3 * n doubles (the all-gathered
        records of best_parents_device); out_ptr: device pointer to 3 * n doubles."""
        _check(lib.obnsl_combine_best_device(self._h, parts_ptr, world, out_ptr))

    def exhaustive_search(self, max_parents, kind=BDEU, ess=1.0):
        """(list of parent lists per node, total score) of the optimal DAG."""
        masks = np.zeros(self.n, dtype=np.uint64)
        total = ctypes.c_double(0.0)
        _check(lib.obnsl_exhaustive_search(self._h, kind, ess, max_parents,
                                           masks.ctypes.data_as(_u64p), ctypes.byref(total)))
        parents = [[j for j in range(self.n) if (int(m) >> j) & 1] for m in masks]
        return parents, total.value

    def exhaustive_search_tables(self, score_ptrs, max_parents):
        """The subset dynamic program + back-track on device score tables (score_ptrs[k-1]:
        device pointer of the size-k table).  Returns (parent lists, total score)."""
        arr = (_vp * len(score_ptrs))(*score_ptrs)
        masks = np.zeros(self.n, dtype=np.uint64)
        total = ctypes.c_double(0.0)
        _check(lib.obnsl_exhaustive_search_tables(self._h, arr, max_parents,
                                                  masks.ctypes.data_as(_u64p), ctypes.byref(total)))
        return [[j for j in range(self.n) if (int(m) >> j) & 1] for m in masks], total.value

    def simulated_annealing(self, max_parents, max_iters=10000, init_temp=1.0, cooling_rate=0.9995,
                            is_deterministic=False, seed=0, num_chains=0, kind=BDEU, ess=1.0):
        """(list of parent lists per node, total score) of the best graph found."""
        width = max(max_parents, 1)
        par = np.zeros((self.n, width), dtype=np.uint32)
        cnt = np.zeros(self.n, dtype=np.uint32)
        total = ctypes.c_double(0.0)
        _check(lib.obnsl_simulated_annealing(self._h, kind, ess, max_parents, max_iters, init_temp,
                                             cooling_rate, int(is_deterministic), seed, num_chains,
                                             par.ctypes.data_as(_u32p), cnt.ctypes.data_as(_u32p),
                                             ctypes.byref(total)))
        return [par[v, :cnt[v]].tolist() for v in range(self.n)], total.value

    def count_path(self, k):
        """"mfma", "valu" or "histogram": the kernel that counts the size-k tables."""
        path = ctypes.c_int(-1)
        _check(lib.obnsl_count_path(self._h, k, ctypes.byref(path)))
        return {0: "mfma", 1: "valu", 2: "histogram"}[path.value]

    def kernel_timing(self, every):
        """Time the count launches of every `every`-th set of row bitmaps (0: none, the default)."""
        _check(lib.obnsl_kernel_timing(self._h, every))

    def kernel_time(self, k, reset=False):
        """(total ms, timed launches) of the size-k counting kernel since the last reset."""
        ms = ctypes.c_double(0.0)
        n = ctypes.c_uint32(0)
        _check(lib.obnsl_kernel_time(self._h, k, ctypes.byref(ms), ctypes.byref(n), int(reset)))
        return ms.value, n.value


class Group:
    """Several GPUs behind one call (obnsl_group): `devices` lists the HIP device of
    every rank; a device listed twice makes ranks share it (copy exchange)."""

    def __init__(self, codes, card, devices):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        card = np.ascontiguousarray(card, dtype=np.uint32)
        self.n, self.N, self.card = int(codes.shape[0]), int(codes.shape[1]), card
        self.devices = [int(x) for x in devices]
        dev = (ctypes.c_int * len(self.devices))(*self.devices)
        self._h = _vp()
        _check(lib.obnsl_group_create(codes.ctypes.data, self.N, self.N, self.n, card.ctypes.data_as(_u32p),
                                      dev, len(self.devices), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.obnsl_group_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def host_time_ms(self):
        """Host time the last call spent queueing the ranks' launches (all ranks at once)."""
        t = ctypes.c_double(0.0)
        _check(lib.obnsl_group_host_time(self._h, ctypes.byref(t)))
        return t.value

    def info(self):
        """(number of ranks, True when the exchanges go through RCCL)."""
        w, r = ctypes.c_uint32(0), ctypes.c_int(0)
        _check(lib.obnsl_group_info(self._h, ctypes.byref(w), ctypes.byref(r)))
        return w.value, bool(r.value)

    def best_parents(self, max_parents, kind=BDEU, ess=1.0, repack=False):
        sc = np.zeros(self.n, dtype=np.float64)
        sz = np.zeros(self.n, dtype=np.uint32)
        rk = np.zeros(self.n, dtype=np.uint64)
        _check(lib.obnsl_group_best_parents(self._h, kind, ess, max_parents, int(repack),
                                            sc.ctypes.data_as(_f64p), sz.ctypes.data_as(_u32p),
                                            rk.ctypes.data_as(_u64p)))
        return sc, sz, rk

    def score_tables(self, max_parents, kind=BDEU, ess=1.0, repack=False):
        tabs = [np.zeros((binomial(self.n, k), k), dtype=np.float64) for k in range(1, max_parents + 2)]
        arr = (_vp * len(tabs))(*[t.ctypes.data for t in tabs])
        _check(lib.obnsl_group_score_tables(self._h, kind, ess, max_parents, int(repack), arr))
        return tabs

    def exhaustive_search(self, max_parents, kind=BDEU, ess=1.0):
        masks = np.zeros(self.n, dtype=np.uint64)
        total = ctypes.c_double(0.0)
        _check(lib.obnsl_group_exhaustive_search(self._h, kind, ess, max_parents,
                                                 masks.ctypes.data_as(_u64p), ctypes.byref(total)))
        return [[j for j in range(self.n) if (int(m) >> j) & 1] for m in masks], total.value
