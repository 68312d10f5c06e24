"""CPU: the committed profile summary that bench.py quotes as `roofline.traffic` belongs
to the default benchmark workload and to the matrix-core counting path, and its own
figures are consistent (matrix-pipe time below the launch time, per-round summaries
present for every benchmark configuration)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def test_default_summary_matches_the_default_workload():
    s = load("pmc_summary.json")
    default = "alarm37_1e6_bdeu_mp3"      # bench.py's --workload default
    assert s["workload"] == default and default in bench.WORKLOADS
    assert s["count_path"] == "mfma"
    assert s["traffic_bytes_per_launch"] > 0
    assert s["traffic_bytes_per_launch"] == s["fetch_bytes_per_launch_corrected"] + s["write_bytes_per_launch"]


def test_matrix_pipe_time_is_below_launch_time_in_every_summary():
    for cfg, workload in ((2, "alarm37_1e6_bdeu_mp3"), (3, "alarm37_1e6_bic_mp4"),
                          (4, "hepar70_1e7_bdeu_mp3"), (5, "uniform128x4_1e7_bdeu_mp3")):
        for rnd in (2, 3):
            s = load("r%d_cfg%d_pmc_summary.json" % (rnd, cfg))
            assert s["workload"] == workload
            assert s["mfma"]["busy_cycles_per_instruction"] == 32.0
            clock_hz = s["clock"]["effective_clock_ghz"] * 1e9
            pipe_s = s["mfma"]["instructions_per_launch"] * 32 / 1024 / clock_hz   # per SIMD, 1024 SIMDs
            launch_s = s["clock"]["launch_ns_under_counters"] * 1e-9
            assert 0.2 < pipe_s / launch_s < 1.0, (cfg, pipe_s, launch_s)
            assert s["kernel_trace"]["calls"] >= (4 if rnd == 2 else 1)


def test_histogram_path_summary():
    """The per-row counting path has its own profile: one launch over many variables."""
    s = load("r3_hist_pmc_summary.json")
    assert s["count_path"] != "mfma"
    assert s["kernel_trace"]["calls"] >= 1
