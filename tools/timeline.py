"""Prints the kernel timeline of the last timed pass of a rocprofv3 --kernel-trace run
(CSV): start offset from the pass's first kernel, duration, queue, kernel name.
   python tools/timeline.py <dir with *_kernel_trace.csv> [first kernel substring]"""
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "pack_planes"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
end = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f"{s/1000:9.1f} us  +{(e-s)/1000:8.1f} us  q{r.get('Queue_Id','?'):>3}  {name}")
print(f"pass: {end/1000:.1f} us")
