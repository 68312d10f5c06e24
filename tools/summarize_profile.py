"""Copies the summaries of one tools/profile_bench.sh run from gpurun_out/ into
profiles/ and derives profiles/pmc_summary.json (HBM traffic per launch of the
dominant kernel).

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE
counts 64 B per 128-B request of 16-B-per-lane coalesced loads, so it is doubled;
WRITE_SIZE is exact for one-dword-per-lane atomics and 16-B stores."""
import csv
import glob
import json
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]          # e.g. prof r1g
dominant = sys.argv[3] if len(sys.argv) > 3 else "bitgemm_kernel<2, 2, 4>"
workload = sys.argv[4] if len(sys.argv) > 4 else "alarm37_1e6_bdeu_mp3"


def one(pattern):
    # gpurun_out/ keeps the files of earlier runs: take the newest
    import os
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(one(f"gpurun_out/{tag}_trace/*/*kernel_stats.csv"), f"profiles/{name}_bench_kernel_stats.csv")
shutil.copy(f"gpurun_out/{tag}_bench_under_trace.json", f"profiles/{name}_bench_under_trace.json")


def per_launch(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if dominant in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(per.values())
    return sum(vals) / len(vals), len(vals)


fetch_csv = one(f"gpurun_out/{tag}_fetch/*/*counter_collection.csv")
write_csv = one(f"gpurun_out/{tag}_write/*/*counter_collection.csv")
shutil.copy(fetch_csv, f"profiles/{name}_pmc_fetch_size.csv")
shutil.copy(write_csv, f"profiles/{name}_pmc_write_size.csv")
f_kib, nf = per_launch(fetch_csv, "FETCH_SIZE")
w_kib, nw = per_launch(write_csv, "WRITE_SIZE")
sq = {}
sq_hits = glob.glob(f"gpurun_out/{tag}_sq/*/*counter_collection.csv")
if sq_hits:
    sq_csv = one(f"gpurun_out/{tag}_sq/*/*counter_collection.csv")
    shutil.copy(sq_csv, f"profiles/{name}_pmc_sq.csv")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"):
        try:
            sq[c] = per_launch(sq_csv, c)[0]
        except ZeroDivisionError:
            pass
clock = None
grbm_hits = glob.glob(f"gpurun_out/{tag}_grbm/*/*counter_collection.csv")
if grbm_hits:
    grbm_csv = one(f"gpurun_out/{tag}_grbm/*/*counter_collection.csv")
    shutil.copy(grbm_csv, f"profiles/{name}_pmc_grbm.csv")
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the launch duration comes from the same rows
    cyc, dur = [], []
    seen = set()
    for r in csv.DictReader(open(grbm_csv)):
        if dominant in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc.append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if cyc and dur and len(seen):
        clock = {"grbm_gui_active_per_launch": sum(cyc) / len(seen),
                 "launch_ns_under_counters": sum(dur) / len(dur),
                 "effective_clock_ghz": (sum(cyc) / len(seen)) / 8.0 / (sum(dur) / len(dur))}
summary = {
    "workload": workload,
    "kernel": dominant,
    "launches_sampled": [nf, nw],
    "fetch_bytes_per_launch_corrected": f_kib * 1024 * 2,
    "write_bytes_per_launch": w_kib * 1024,
    "traffic_bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024,
    "instructions_per_launch": sq,
    "clock": clock,
    "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{name}_pmc_*_size.csv); "
            "counters are in KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B-per-lane coalesced loads)",
}
for path in ("profiles/pmc_summary.json", f"profiles/{name}_pmc_summary.json"):
    with open(path, "w") as f:
        json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1))
for r in csv.DictReader(open(f"profiles/{name}_bench_kernel_stats.csv")):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
