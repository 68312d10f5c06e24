#!/usr/bin/env python3
"""Benchmark of the local-score hot path: (child, parent-set) scores per second.

One step = one pass of the whole hot path over the workload: rebuild the row
bitmaps from the resident uint8 matrix, count and score every subset of size
1..max_parents+1 (this rank's shard of them), reduce to per-node best parent
sets, and, with more than one GPU, all-gather those over RCCL.  The data matrix
is resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
N > 1: one rank per GPU.  Started under torch.distributed.run the process is one
rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment); started plainly with
--gpus N > 1 it starts the N ranks itself, as fresh child processes, before it
has touched the GPU, and exits with their status.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # 37 variables, 10^6 rows, BDeu, <= 3 parents: the configuration the metric is quoted on
    "alarm37_1e6_bdeu_mp3": dict(gen="alarm", N=1_000_000, max_parents=3, score="bdeu"),
    "alarm37_1e6_bic_mp4": dict(gen="alarm", N=1_000_000, max_parents=4, score="bic"),
    "hepar70_1e7_bdeu_mp3": dict(gen="hepar2", N=10_000_000, max_parents=3, score="bdeu"),
    "uniform128x4_1e7_bdeu_mp3": dict(gen="uniform128", N=10_000_000, max_parents=3, score="bdeu"),
    "asia8_1e5_bdeu_mp2": dict(gen="asia", N=100_000, max_parents=2, score="bdeu"),
    # the 128-variable shape at a tenth of the rows (development probe of the large-tile kernel)
    "uniform128x4_1e6_bdeu_mp3": dict(gen="uniform128", N=1_000_000, max_parents=3, score="bdeu"),
    # development probes for the path-selection threshold
    "uniform20x6_1e6_bdeu_mp3": dict(gen="uniform20x6", N=1_000_000, max_parents=3, score="bdeu"),
    "uniform20x5_1e6_bdeu_mp3": dict(gen="uniform20x5", N=1_000_000, max_parents=3, score="bdeu"),
    "uniform20x8_1e6_bdeu_mp3": dict(gen="uniform20x8", N=1_000_000, max_parents=3, score="bdeu"),
}
HBM_PEAK_GBS = 8000.0
# dense FP4 matrix-core peak: 1024 SIMDs x (32 x 32 x 64 multiply-adds per 32 cycles) x 2.4 GHz
MFMA_FP4_PEAK_TFLOPS = 1024 * 32 * 32 * 64 * 2 / 32 * 2.4e9 / 1e12
# what the instruction sustains when every SIMD issues nothing else (tools/mfma_probe.hip:
# 16.5 ns per 32 x 32 x 64 FP4 instruction and SIMD; the clock gives way under that load)
MFMA_FP4_SUSTAINED_TFLOPS = 1024 * 32 * 32 * 64 * 2 / 16.5e-9 / 1e12


def host_cpu():
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"model": model, "os_cpu_count": os.cpu_count(),
            "sched_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")}


def make_data(spec):
    from openbnsl_amd import synth
    if spec["gen"] == "alarm":
        return synth.alarm_shaped(spec["N"], seed=0)
    if spec["gen"] == "hepar2":
        return synth.hepar2_shaped(spec["N"], seed=0)
    if spec["gen"] == "uniform128":
        return synth.uniform(128, 4, spec["N"], seed=0)
    if spec["gen"].startswith("uniform20x"):
        return synth.uniform(20, int(spec["gen"][len("uniform20x"):]), spec["N"], seed=0)
    if spec["gen"] == "asia":
        return synth.asia(spec["N"], seed=0)
    raise ValueError(spec["gen"])


def cpu_baseline(codes, card, names, max_parents, budget_s=20.0, timed_table=None):
    """Reference C++ (oracle/_ref) timed single-threaded, as the reference runs
    it, on a bounded random sample of this workload's own size-(max_parents+1)
    subsets (drawn over 16 randomly chosen columns, so that the string-typed
    frame the reference ingests stays small).  timed_table: the host copy of the
    score table of that size as the TIMED pass wrote it; the sampled subsets'
    rows are compared with the reference's scores (the in-run parity gate)."""
    import oracle
    from openbnsl_amd import engine, synth
    k = max_parents + 1
    n = codes.shape[0]
    rng = np.random.default_rng(0)
    ncols = min(n, 16)
    cols = sorted(rng.choice(n, size=ncols, replace=False).tolist())
    subs_all = oracle.enumerate_subsets(ncols, min(k, ncols))
    subs_all = subs_all[rng.permutation(len(subs_all))]
    ref = oracle.ref()
    out = {}
    if ref is not None:
        sub_codes = np.ascontiguousarray(codes[cols])
        df = synth.to_dataframe(sub_codes, [names[c] for c in cols])
        dfw = ref.DataframeWrapper(df)
        probe = [list(map(int, subs_all[0]))]
        secs, _ = ref.score_subsets_timed(dfw, probe, 1.0)
        m = int(max(1, min(len(subs_all), budget_s / max(secs, 1e-6))))
        pick = [list(map(int, s)) for s in subs_all[:m]]
        secs, ref_scores = ref.score_subsets_timed(dfw, pick, 1.0)
        parity = None
        if timed_table is not None and len(pick[0]) == k:
            # rows of the table the timed pass wrote (variable ids mapped back to the full matrix)
            ranks = [engine.rank([cols[i] for i in s]) for s in pick]
            have = timed_table.reshape(-1, k)[ranks]
            want = np.array(ref_scores, dtype=np.float64)
            parity = float(np.max(np.abs(have - want) / np.abs(want)))
        out = {"value": m * len(pick[0]) / secs, "unit": "scores/s", "cores": 1, "kind": "reference",
               "parity_max_rel_err": parity, "parity_tables": m,
               "parity_source": "rows of the score table written by the timed pass",
               "sample": f"{m} random size-{len(pick[0])} subsets over {ncols} random columns of the {n}, "
                         f"all {codes.shape[1]} rows, reference ContingencyTable + calculate_local_score, {secs:.1f} s"}
    else:
        sub_codes = np.ascontiguousarray(codes[cols[:6]])
        t0 = time.perf_counter()
        sc = oracle.score_all(sub_codes, card[cols[:6]], min(k, 6), nthreads=1)
        secs = time.perf_counter() - t0
        out = {"value": sc.size / secs, "unit": "scores/s", "cores": 1, "kind": "port",
               "sample": f"{sc.shape[0]} size-{sc.shape[1]} subsets over 6 columns, all rows, oracle.c, {secs:.1f} s"}
    # the same restatement spread over every host core this job is given: OpenMP's own
    # default (omp_get_max_threads(), i.e. OMP_NUM_THREADS where the box sets it -- a GPU box
    # shares its host's cores between jobs -- else every core)
    threads = int(oracle.lib.oracle_max_threads())
    wide = sorted(rng.choice(n, size=min(n, 20), replace=False).tolist())
    sub_codes = np.ascontiguousarray(codes[wide])
    t0 = time.perf_counter()
    sc = oracle.score_all(sub_codes, card[wide], min(k, len(wide)), nthreads=threads)
    secs = time.perf_counter() - t0
    omp = {"value": sc.size / secs, "unit": "scores/s", "cores": threads, "kind": "port", "host": host_cpu(),
           "sample": f"{sc.shape[0]} size-{sc.shape[1]} subsets over {len(wide)} columns, all rows, "
                     f"oracle.c with OpenMP over subsets, {secs:.1f} s"}
    return out, omp


def stream_read_gbs(torch, engine, dev, nbytes=2 << 30, reps=5):
    """Read bandwidth the memory system gives a plain streaming kernel (16 bytes
    per lane) over a buffer far larger than the 256 MiB Infinity Cache."""
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    buf.random_(0, 1 << 20)
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    engine.stream_read(buf.data_ptr(), nbytes, stream, sink.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.stream_read(buf.data_ptr(), nbytes, stream, sink.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def spawn_ranks(n):
    """Starts this script once per rank (fresh processes; this parent has not
    touched the GPU and never does), waits for them, returns the worst status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    for pr in procs:
        rc = pr.wait()
        if rc != 0:
            worst = worst or rc
            for other in procs:        # a dead rank leaves the others waiting in a collective
                if other.poll() is None:
                    other.terminate()
    return worst


def two_data_sets_in_flight(torch, HotPath, hot, codes, card, mp, kind, dev_index, steps, warmup):
    """Secondary figure (never `value`): the same passes alternating between two resident
    copies of the data set, each on a stream of its own, so that the start of one pass
    (row bitmaps, count launches) fills the GPU while the previous pass ends (its last
    scorer and the arg-max leave most of the chip idle).  Every pass does the whole work
    of a step; only their ends overlap."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hot2 = HotPath(codes, card, mp, kind=kind, ess=1.0, rank=0, world=1, device_index=dev_index,
                       exchange=False)
    pipes = [(hot, torch.cuda.current_stream()), (hot2, side)]

    def run(i):
        h, st = pipes[i & 1]
        with torch.cuda.stream(st):
            h.step()

    steps = max(2, steps // 2 * 2)
    for i in range(2 * max(1, warmup)):
        run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    same = bool((hot.best == hot2.best).all().item())
    hot2.close()
    return {"value": hot.scores_per_pass * steps / elapsed, "unit": "scores/s", "steps": steps,
            "ms_per_step": elapsed / steps * 1e3, "same_best_parent_sets": same,
            "note": "passes alternate between two resident copies of the data set on two streams; "
                    "each pass is a whole step, the end of one overlaps the start of the next"}


def exhaustive_search_wallclock(engine, with_cpu):
    """Second half of the metric: wall-clock of the whole exhaustive search
    (scores + subset DP + back-track, data resident) on the 8-variable ASIA-shaped
    case (10^5 rows, <= 2 parents) and on a 20-variable case; the oracle's restatement
    of the reference search is timed beside the former (the reference's own search
    cannot be compiled here)."""
    from openbnsl_amd import synth
    res = []
    cases = [("asia8_1e5_bdeu_mp2", synth.asia(100_000, 0)[:2], 2),
             ("random20_1e4_bdeu_mp3", synth.random_bn(([2, 3, 2, 4, 2, 3] * 4)[:20], 10_000, seed=20)[:2], 3)]
    for name, (codes, card), mp in cases:
        d = engine.Data(codes, card)
        d.exhaustive_search(mp)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            d.pack()    # new row bitmaps: no count table of the previous call is reused
            parents, total = d.exhaustive_search(mp)
            ts.append(time.perf_counter() - t0)
        d.close()
        rec = {"workload": name, "ms": float(np.median(ts)) * 1e3, "total_score": total,
               "edges": int(sum(len(p) for p in parents))}
        if with_cpu and codes.shape[0] <= 8:
            import oracle
            t0 = time.perf_counter()
            _, ototal = oracle.exhaustive_search(codes, card, mp)
            rec["cpu_port_ms"] = (time.perf_counter() - t0) * 1e3
            rec["cpu_total_score"] = ototal
        res.append(rec)
    return res


def simulated_annealing_wallclock(engine, data, with_cpu):
    """Wall-clock of simulated_annealing (a caller of the same table + score pair)
    on the resident benchmark data set and on the 8-variable ASIA-shaped data, where the
    reference's own implementation (oracle/_ref) is timed beside it on a bounded
    number of steps."""
    from openbnsl_amd import synth
    res = []
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        data.pack()     # new row bitmaps: no count table of the previous call is reused
        parents, total = data.simulated_annealing(3, 10000, seed=0, num_chains=4, is_deterministic=True)
        ts.append(time.perf_counter() - t0)
    res.append({"workload": "resident data set, max_parents 3, 10000 steps x 4 chains",
                "ms": float(np.median(ts)) * 1e3, "total_score": total,
                "edges": int(sum(len(p) for p in parents))})
    codes, card, names = synth.asia(100_000, 0)
    d = engine.Data(codes, card)
    d.simulated_annealing(3, 1000, seed=0, num_chains=1, is_deterministic=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        d.pack()
        parents, total = d.simulated_annealing(3, 1000, seed=0, num_chains=1, is_deterministic=True)
        ts.append(time.perf_counter() - t0)
    d.close()
    rec = {"workload": "asia8_1e5_bdeu_mp3, 1000 steps x 1 chain", "ms": float(np.median(ts)) * 1e3,
           "total_score": total, "edges": int(sum(len(p) for p in parents))}
    if with_cpu:
        import oracle
        ref = oracle.ref()
        if ref is not None:
            dfw = ref.DataframeWrapper(synth.to_dataframe(codes, names))
            secs, want = ref.simulated_annealing(dfw, 1.0, 3, 1000, 1.0, 0.9995, True, 0, 1)
            rec["cpu_reference_ms"] = secs * 1e3
            rec["same_graph_as_reference"] = [list(map(int, p)) for p in want] == parents
    res.append(rec)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="alarm37_1e6_bdeu_mp3", choices=sorted(WORKLOADS))
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="the W + K passes are timed twice: first from the idle device (reported as "
                         "cold_start), then again after this many milliseconds of untimed passes, when "
                         "the device has reached its steady speed (0: only the first measurement)")
    ap.add_argument("--sharding", default="auto", choices=["auto", "subsets", "rows"],
                    help="how several ranks split a pass (auto: the rows of a short pass, else the subsets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-hot-path", action="store_true",
                    help="skip the exhaustive-search / annealing wall-clock legs and the CPU baseline "
                         "(profiling runs: every kernel launch then belongs to the timed pass)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from openbnsl_amd import engine
    from openbnsl_amd.pipeline import HotPath

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal on a box with fewer GPUs than ranks: OBNSL_BENCH_BACKEND=gloo maps
    # every rank to an existing device and exchanges through host memory
    backend = os.environ.get("OBNSL_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # OBNSL_BENCH_FORCE_EXCHANGE=1: create the process group and run the all-gather
    # and combine even with one rank (exercises the RCCL path on a one-GPU box)
    exchange = world > 1 or os.environ.get("OBNSL_BENCH_FORCE_EXCHANGE") == "1"
    if exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")

    spec = WORKLOADS[args.workload]
    codes, card, names = make_data(spec)
    n, N = codes.shape
    mp = spec["max_parents"]
    kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC

    hot = HotPath(codes, card, mp, kind=kind, ess=1.0, rank=rank, world=world, device_index=dev_index,
                  exchange=exchange, sharding=args.sharding)
    data = hot.data
    # the count launch is timed with HIP events on its stream inside the timed region -- in a
    # sample of the passes: an event pair costs the stream ~10 us, 2 % of a short pass
    timing_every = max(1, min(5, args.steps // 4))
    data.kernel_timing(timing_every)
    rows_mode = hot.sharding == "rows"
    rows_launch = data.N      # rows one count launch of this rank contracts over (its slice when the rows are split)
    tables = hot.tables
    n_scores = hot.scores_per_pass
    step = hot.step

    def barrier():
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
            torch.cuda.synchronize()

    def measure():
        """W untimed passes, then exactly K passes between barriers; the slowest rank's time."""
        for _ in range(args.warmup):
            step()
        data.kernel_time(mp + 1, reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if exchange:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # An MI355X that has been idle takes ~30 ms of load to reach its steady speed (the time
    # per pass falls by ~10 % over the first 50 passes of the 37-variable workload,
    # tools/step_ramp.py).  The W + K passes are therefore measured twice: from the idle
    # device as the command line says (cold_start), and again after settle-ms of untimed
    # passes -- the steady rate a resident service sees, which is `value`.
    elapsed = measure()
    cold = None
    if args.settle_ms > 0:
        cold = {"ms_per_step": elapsed / args.steps * 1e3, "value": n_scores * args.steps / elapsed,
                "unit": "scores/s",
                "note": "the same W warmup + K timed passes started on the idle device, before the settle passes"}
        # (a pass count every rank agrees on: from the slowest rank's cold time)
        settle_steps = max(8, int(args.settle_ms / cold["ms_per_step"]))
        for i in range(settle_steps):
            step()
            if i % 16 == 15:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        cold["settle_ms"] = args.settle_ms
        cold["settle_steps"] = settle_steps
        elapsed = measure()

    ms_per_step = elapsed / args.steps * 1e3
    value = n_scores * args.steps / elapsed
    # host copy of the top-size score table exactly as the timed pass left it (the
    # parity gate of the cpu_baseline leg reads its rows)
    timed_top = tables[mp].detach().cpu() if (rank == 0 and world == 1) else None

    # dominant kernel: the counting launch of the largest subset size, timed live
    # with HIP events on the stream it runs on (obnsl_kernel_time).
    k_top = mp + 1
    total_ms, launches = data.kernel_time(k_top)
    launches_per_step = max(1, round(launches / max(1, args.steps / timing_every)))
    kernel_ms = total_ms / max(1, launches)       # average duration of one launch
    path = data.count_path(k_top)
    # scores of this rank's shard of the top size (count the finite entries)
    top_scores_all = data.num_subsets(k_top) * k_top
    top_scores_rank = int(torch.isfinite(tables[k_top - 1]).sum().item())
    # algorithmic bytes: N bytes per (child, parent set) score (one uint8
    # column of N rows per score) x the scores one launch yields
    # (ranks that split the rows: every score of the size, over the rank's rows)
    alg_bytes_per_launch = (top_scores_all if rows_mode else top_scores_rank) * rows_launch / launches_per_step
    hbm_alg = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None

    # HBM traffic of that kernel: PMC counters need rocprofv3 passes of their own
    # (tools/profile_bench.sh); the committed summary is quoted only when it was
    # taken on this workload with this kernel, else null
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path) and world == 1:
        with open(pmc_path) as f:
            pmc = json.load(f)
        if pmc.get("workload") == args.workload and pmc.get("count_path") == path:
            traffic = pmc["traffic_bytes_per_launch"] / 1e9
            traffic_src = ("profiles/pmc_summary.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                           "passes over this command, GB per launch, FETCH_SIZE doubled as on gfx950)")

    if path == "mfma" and kernel_ms > 0:
        # what binds the launch: the matrix cores.  Algorithmic work = one
        # multiply-add per (counted cell, sample): the cells are the joint states
        # with every variable in a non-reference state (the others follow by
        # inclusion-exclusion in the scorer), each FP4 multiply-add is one AND of
        # two sample bits.  Peak = dense FP4 rate of the chip.
        # (the launch's own figures: a short pass counts every subset size in this one launch,
        # and a shard counts its share of the cells)
        info = data.count_launch_info(k_top)
        cells = float(info["cells_asked"]) or None
        flops = 2.0 * cells * rows_launch / launches_per_step if cells else None
        ach = flops / (kernel_ms * 1e-3) / 1e12 if flops else None
        roofline = {"bound": "mfma", "achieved": ach, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_FP4_PEAK_TFLOPS if ach else None, "traffic": traffic,
                    "traffic_unit": "GB per launch", "traffic_source": traffic_src,
                    "frac_of_sustained": ach / MFMA_FP4_SUSTAINED_TFLOPS if ach else None,
                    "sustained_peak": MFMA_FP4_SUSTAINED_TFLOPS,
                    "kernel": f"bitgemm_mfma_kernel (subset size {k_top}, FP4 32x32x64)",
                    "kernel_ms": kernel_ms, "launches_per_step": launches_per_step,
                    "timed_launches": launches, "timed_every_nth_step": timing_every,
                    "algorithmic_flops_per_launch": flops,
                    "algorithmic_cells_per_launch": cells,
                    "subset_sizes_in_launch": info["sizes_counted"],
                    "tiles": info["tiles"], "work_units": info["units"], "workgroups": info["workgroups"],
                    "mfma_cells_executed_over_algorithmic": (info["cells_computed"] / cells) if cells else None,
                    "note": "2 flops per (counted cell, sample); peak = 1024 SIMDs x 65536 FP4 "
                            "multiply-adds per 32 cycles at 2.4 GHz; sustained_peak = the same at the "
                            "16.5 ns per instruction a SIMD sustains with nothing else to issue"}
    else:
        roofline = {"bound": "hbm", "achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (hbm_alg / HBM_PEAK_GBS) if hbm_alg else None, "traffic": traffic,
                    "traffic_unit": "GB per launch", "traffic_source": traffic_src,
                    "kernel": f"counting kernel of subset size {k_top} ({path} path)",
                    "kernel_ms": kernel_ms, "launches_per_step": launches_per_step,
                    "algorithmic_bytes_per_launch": alg_bytes_per_launch}

    out = {
        "metric": "parent-set scores/sec",
        "value": value,
        "unit": "scores/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "fp4 0/1 multiply-adds with exact f32 counts (u32 tables) + f64 lgamma"
                 if path == "mfma" else "u32 counts + f64 lgamma",
        "data": "synthetic",
        "config": {"workload": args.workload, "num_vars": int(n), "num_rows": int(N),
                   "max_parents": mp, "score": spec["score"], "scores_per_step": int(n_scores),
                   "world_size": int(dist.get_world_size()) if exchange else 1,
                   "backend": (dist.get_backend() if exchange else None),
                   "sharding": (f"rows split over {world} ranks ({rows_launch} on rank 0); one all-reduce of all "
                                "count tables; tables scored in shares; per-node bests all-gathered" if rows_mode else
                                f"subsets split over {world} rank(s) by cost; per-node bests all-gathered")},
        "cold_start": cold,
        "roofline": roofline,
        # the per-score definition above, kept beside the binding roofline: algorithmic
        # column bytes per second against the HBM peak (the row bitmaps are cache
        # resident and 8x denser than the columns, so this exceeds 1 by design)
        "roofline_hbm_algorithmic": {"achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": (hbm_alg / HBM_PEAK_GBS) if hbm_alg else None,
                                     "algorithmic_bytes_per_launch": alg_bytes_per_launch},
    }
    if rank == 0 and world == 1 and not args.only_hot_path:
        out["two_data_sets_in_flight"] = two_data_sets_in_flight(torch, HotPath, hot, codes, card, mp, kind,
                                                                   dev_index, args.steps, args.warmup)
    if rank == 0 and world == 1 and not args.only_hot_path:
        out["hbm_stream_read_gbs"] = stream_read_gbs(torch, engine, dev)
    if rank == 0 and world == 1 and not args.only_hot_path:
        out["exhaustive_search"] = exhaustive_search_wallclock(engine, not args.no_cpu_baseline)
        out["simulated_annealing"] = simulated_annealing_wallclock(engine, data, not args.no_cpu_baseline)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.only_hot_path:
        try:
            base, omp = cpu_baseline(codes, card, names, mp,
                                     timed_table=timed_top.numpy() if timed_top is not None else None)
            out["cpu_baseline"] = base
            out["cpu_baseline_omp"] = omp
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    hot.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
