"""Per-workgroup timeline of the matrix-core launch of one subset size (probe build only).

    OBNSL_ENGINE_LIB=tools/bin/libobnsl_engine_probes.so python tools/unit_trace.py [workload] [k]

Prints, for one pass: the launch's span, the distribution of workgroup durations, how busy
the workgroup slots were (sum of workgroup time / (span x concurrent workgroups)) and when
the last workgroup of each tenth of the launch started and ended."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from openbnsl_amd import engine  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "alarm37_1e6_bdeu_mp3"
    spec = bench.WORKLOADS[wl]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else spec["max_parents"] + 1
    import torch
    from openbnsl_amd.pipeline import HotPath
    codes, card, names = bench.make_data(spec)
    kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
    hot = HotPath(codes, card, spec["max_parents"], kind=kind, ess=1.0)
    lib = engine.lib
    lib.obnsl_probe_trace_begin.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.obnsl_probe_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                           ctypes.POINTER(ctypes.c_uint32)]
    for _ in range(3):
        hot.step()
    torch.cuda.synchronize()
    hot.data.kernel_timing(1)
    hot.data.kernel_time(k, reset=True)
    assert lib.obnsl_probe_trace_begin(hot.data._h, k) == 0
    for _ in range(5):
        hot.step()
    torch.cuda.synchronize()
    ms, launches = hot.data.kernel_time(k)
    buf = np.zeros(4 * (1 << 16), dtype=np.uint64)
    nwg = ctypes.c_uint32(0)
    assert lib.obnsl_probe_trace_read(hot.data._h, buf.ctypes.data, buf.size, ctypes.byref(nwg)) == 0
    n = nwg.value
    t = buf[:4 * n].reshape(n, 4)
    begin = t[:, 0].astype(np.int64)
    end = t[:, 1].astype(np.int64)
    t0 = begin.min()
    b_us = (begin - t0) / 100.0
    e_us = (end - t0) / 100.0
    dur = e_us - b_us
    span = e_us.max()
    hw = t[:, 2]
    xcc = t[:, 3] & 0xf
    runs_done = (t[:, 3] >> 8).astype(np.int64)
    tag = os.environ.get("OBNSL_TRACE_TAG")
    if tag:
        np.save(os.path.join("gpurun_out", f"trace_{tag}.npy"), t)
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    slot = xcc.astype(np.int64) * 64 + se.astype(np.int64) * 16 + cu.astype(np.int64)
    # concurrency over time
    ev = sorted([(x, 1) for x in b_us] + [(x, -1) for x in e_us])
    cur = peak = 0
    for _, dlt in ev:
        cur += dlt
        peak = max(peak, cur)
    out = {
        "workload": wl, "k": k, "workgroups": int(n), "launch_ms_events": ms / max(launches, 1),
        "span_us": float(span), "peak_concurrency": int(peak),
        "slot_busy_frac": float(dur.sum() / (span * peak)),
        "dur_us": {q: float(np.percentile(dur, q)) for q in (0, 10, 50, 90, 100)},
        "first_end_us": float(e_us.min()), "last_begin_us": float(b_us.max()),
        "distinct_cus": int(len(np.unique(slot))),
        "runs_done": {"min": int(runs_done.min()), "median": float(np.median(runs_done)), "max": int(runs_done.max())},
        "dur_us_mean_by_xcc": [float(dur[xcc == x].mean()) if (xcc == x).any() else None for x in range(8)],
        "wgs_per_xcc": np.bincount(xcc.astype(np.int64), minlength=8).tolist(),
        "end_us_by_decile_of_start_order": [float(e_us[np.argsort(b_us)][int(n * q / 10):max(int(n * (q + 1) / 10), int(n * q / 10) + 1)].max()) for q in range(10)],
        "idle_tail_us_mean": float(np.mean(span - np.array([e_us[slot == s].max() for s in np.unique(slot)]))),
        "env": {k2: v for k2, v in os.environ.items() if k2.startswith("OBNSL_")},
    }
    print(json.dumps(out))
    hot.close()


if __name__ == "__main__":
    main()
