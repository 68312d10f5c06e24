"""Per-pass device time of the first passes after start-up (events around every pass,
no host synchronisation in between): how long the device takes to reach its steady
speed.   python tools/step_ramp.py [workload] [passes]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from openbnsl_amd import engine  # noqa: E402


def main():
    import torch
    from openbnsl_amd.pipeline import HotPath
    wl = sys.argv[1] if len(sys.argv) > 1 else "alarm37_1e6_bdeu_mp3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    spec = bench.WORKLOADS[wl]
    codes, card, _ = bench.make_data(spec)
    kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
    hot = HotPath(codes, card, spec["max_parents"], kind=kind, ess=1.0)
    hot.step()                      # plans, buffers
    torch.cuda.synchronize()
    import time
    time.sleep(float(os.environ.get("RAMP_IDLE_S", "0.5")))   # let the device fall idle
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        hot.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print("per-pass ms:", " ".join(f"{x:.3f}" for x in ms))
    print("first 5 %.4f  next 20 %.4f  last 20 %.4f" % (sum(ms[:5]) / 5, sum(ms[5:25]) / 20, sum(ms[-20:]) / 20))
    hot.close()


if __name__ == "__main__":
    main()
