"""Development helper: time per matrix-core instruction and SIMD of the top-size count
launch on uniform data (n variables, r states, N rows), from the unit list the engine
prints under OBNSL_DUMP_UNITS.
    python tools/mfma_rate.py n r N [k]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("MFMA_RATE_CHILD"):
    import torch
    from openbnsl_amd import engine, synth
    n, r, N, k = (int(x) for x in sys.argv[1:5])
    codes, card, _ = synth.uniform(n, r, N, seed=1)
    d = engine.Data(codes, card)
    buf = torch.empty(d.num_subsets(k) * k, dtype=torch.float64, device="cuda")
    d.score_all_device(k, buf.data_ptr())
    torch.cuda.synchronize()
    d.kernel_timing(1)
    d.kernel_time(k, reset=True)
    for _ in range(3):
        d.pack()
        d.score_all_device(k, buf.data_ptr())
    torch.cuda.synchronize()
    ms, launches = d.kernel_time(k)
    print("KERNEL_MS", ms / launches)
    sys.exit(0)

n, r, N = (int(x) for x in sys.argv[1:4])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 4
env = dict(os.environ, MFMA_RATE_CHILD="1", OBNSL_DUMP_UNITS="1")
out = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(r), str(N), str(k)], env=env,
                     capture_output=True, text=True)
ms = float(re.search(r"KERNEL_MS (\S+)", out.stdout).group(1))
# the last "units:" block belongs to the top size
blocks = out.stderr.split("units:")[-1]
head = blocks.splitlines()[0]
mfma = 0
nunits = 0
for m in re.finditer(r"blocks (\d+) steps (\d+): (\d+) units", blocks):
    b, s, u = (int(x) for x in m.groups())
    mfma += b * s * u * 16 * 4
    nunits += u
per_simd = mfma / 1024
print(f"n={n} r={r} N={N} k={k}:{head} -> {nunits} units, {ms:.3f} ms, {per_simd:.0f} MFMA per SIMD, "
      f"{ms * 1e6 / per_simd:.1f} ns per MFMA per SIMD")
