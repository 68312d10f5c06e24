"""Copies the summaries of one tools/profile_bench.sh run from gpurun_out/ into
profiles/ and derives profiles/<name>_pmc_summary.json (per launch of the
dominant kernel: HBM-side traffic, instruction counts, matrix-pipe busy cycles,
clock).  With --default it also writes profiles/pmc_summary.json, which
bench.py quotes as `roofline.traffic` when workload and counting path match.

    python tools/summarize_profile.py <tag> <name> <dominant kernel substring> <workload> [--default]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE
counts 64 B per 128-B request of 16-B-per-lane coalesced loads, so it is doubled;
WRITE_SIZE is exact for one-dword-per-lane atomics and 16-B stores."""
import csv
import glob
import json
import os
import shutil
import sys

tag, name, dominant, workload = sys.argv[1:5]
default = "--default" in sys.argv


def one(pattern):
    hits = glob.glob(pattern)
    return max(hits, key=os.path.getmtime) if hits else None


def per_launch(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if dominant in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(per.values())
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


shutil.copy(one(f"gpurun_out/{tag}_trace/*/*kernel_stats.csv"), f"profiles/{name}_kernel_stats.csv")
shutil.copy(f"gpurun_out/{tag}_bench_under_trace.json", f"profiles/{name}_bench_under_trace.json")
summary = {"workload": workload, "kernel": dominant,
           "count_path": "mfma" if "mfma" in dominant else "histogram" if "hist" in dominant else "valu"}
for r in csv.DictReader(open(f"profiles/{name}_kernel_stats.csv")):
    if dominant in r["Name"]:
        summary["kernel_trace"] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                   "percentage": float(r["Percentage"])}
fetch_csv, write_csv = (one(f"gpurun_out/{tag}_{x}/*/*counter_collection.csv") for x in ("fetch", "write"))
if fetch_csv and write_csv:
    shutil.copy(fetch_csv, f"profiles/{name}_pmc_fetch_size.csv")
    shutil.copy(write_csv, f"profiles/{name}_pmc_write_size.csv")
    f_kib, nf = per_launch(fetch_csv, "FETCH_SIZE")
    w_kib, nw = per_launch(write_csv, "WRITE_SIZE")
    summary.update({"launches_sampled": [nf, nw],
                    "fetch_bytes_per_launch_corrected": f_kib * 1024 * 2,
                    "write_bytes_per_launch": w_kib * 1024,
                    "traffic_bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024})
counters = {}
for part in ("sq", "sq2"):
    path = one(f"gpurun_out/{tag}_{part}/*/*counter_collection.csv")
    if not path:
        continue
    shutil.copy(path, f"profiles/{name}_pmc_{part}.csv")
    names = {r["Counter_Name"] for r in csv.DictReader(open(path))}
    for c in sorted(names):
        v, _ = per_launch(path, c)
        if v is not None:
            counters[c] = v
    if part == "sq2":
        # launch duration under the counters, from the same rows
        dur, seen = [], set()
        for r in csv.DictReader(open(path)):
            if dominant in r["Kernel_Name"] and "Start_Timestamp" in r and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        if dur and "GRBM_GUI_ACTIVE" in counters:
            ns = sum(dur) / len(dur)
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            summary["clock"] = {"launch_ns_under_counters": ns,
                                "effective_clock_ghz": counters["GRBM_GUI_ACTIVE"] / 8.0 / ns}
summary["counters_per_launch"] = counters
if counters.get("SQ_INSTS_MFMA") and counters.get("SQ_VALU_MFMA_BUSY_CYCLES") and counters.get("SQ_BUSY_CYCLES"):
    summary["mfma"] = {
        "instructions_per_launch": counters["SQ_INSTS_MFMA"],
        "busy_cycles_per_instruction": counters["SQ_VALU_MFMA_BUSY_CYCLES"] / counters["SQ_INSTS_MFMA"],
    }
summary["note"] = ("rocprofv3 --pmc passes, each in a run of its own (FETCH_SIZE; WRITE_SIZE; SQ counters; GRBM + SQ wait "
                   "counters); sizes in KiB per dispatch; FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request of "
                   "16-B-per-lane coalesced loads)")
paths = [f"profiles/{name}_pmc_summary.json"] + (["profiles/pmc_summary.json"] if default else [])
for path in paths:
    with open(path, "w") as f:
        json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1))
for r in csv.DictReader(open(f"profiles/{name}_kernel_stats.csv")):
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
