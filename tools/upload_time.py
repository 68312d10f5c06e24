"""Development helper: time of obnsl_data_create from a host matrix (allocation,
host-to-device copy over PCIe from pageable memory, first pack) per workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openbnsl_amd import engine
for name in ("alarm37_1e6_bdeu_mp3", "hepar70_1e7_bdeu_mp3"):
    codes, card, _ = bench.make_data(bench.WORKLOADS[name])
    engine.Data(codes, card).close()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); d = engine.Data(codes, card); ts.append(time.perf_counter() - t0); d.close()
    print(f"{name}: {codes.nbytes/1e6:.0f} MB, create {min(ts)*1e3:.1f} ms (best of 5) = {codes.nbytes/min(ts)/1e9:.1f} GB/s", flush=True)
