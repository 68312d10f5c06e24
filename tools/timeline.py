"""Development helper: kernel timeline of the last bench step from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_planes" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 40]:
    nm = r["Kernel_Name"]
    nm = nm.split("::")[-1][:44] if "::" in nm else nm[:44]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} us  q={r.get('Queue_Id','?')} grid={r.get('Grid_Size','?'):>8} {nm}")
