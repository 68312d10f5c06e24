"""Development soak (GPU + oracle): random data shapes through the batch path -- every
subset size up to kmax, sampled tables of the batch kernel's own counters and their
scores against the oracle, repeated passes, and the same pass in shards.  Prints one line
per case and a summary.
    python tools/soak_counts.py [cases] [first seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from openbnsl_amd import engine, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(seed0, seed0 + cases):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 44))
    cards = rng.integers(2, 6, size=n).tolist()
    N = int(rng.choice([97, 1000, 4097, 33_000, 70_001, 250_000, 1_000_003]))
    kmax = int(rng.integers(2, 6)) if n <= 30 else int(rng.integers(2, 5))
    kmax = min(kmax, n)
    try:
        codes, card, _, _ = synth.random_bn(cards, N, seed=seed)
    except AssertionError:      # the generator insists on every state being observed
        print(f"seed {seed}: n={n} N={N}: a state is unobserved, skipped", flush=True)
        continue
    d = engine.Data(codes, card)
    worst, paths = 0.0, []
    ok = True
    kept = {}
    for k in range(1, kmax + 1):
        scores = d.score_all(k)
        kept[k] = scores
        paths.append(d.count_path(k))
        for _ in range(6):
            vs = sorted(rng.choice(n, size=k, replace=False).tolist())
            want = oracle.count(codes, card, vs)
            got = d.full_table_from_reduced(vs) if d.count_path(k) != "histogram" else d.count(vs)
            if not np.array_equal(got, want):
                ok = False
                print("  COUNT MISMATCH", seed, k, vs, flush=True)
            row = scores[engine.rank(vs)]
            for ci in range(k):
                w = oracle.local_score(want, card[vs], ci)
                worst = max(worst, abs(row[ci] - w) / abs(w))
    if worst > 1e-9:
        ok = False
    # a second and a third pass over fresh bitmaps (the tile images of the split launches
    # are cleared by the pack in between): bit-identical scores, largest size first once
    for order in (range(1, kmax + 1), range(kmax, 0, -1)):
        d.pack()
        for k in order:
            if not np.array_equal(d.score_all(k), kept[k]):
                ok = False
                print("  PASS MISMATCH", seed, k, flush=True)
    # the same sizes in shards (a random world size, one data set object per rank, lower
    # sizes first as a pass does): every score exactly once, bit-identical to the whole pass
    import torch
    world = int(rng.choice([2, 3, 5, 8]))
    merged = {k: torch.full((kept[k].size,), float("-inf"), dtype=torch.float64, device="cuda")
              for k in kept}
    for r in range(world):
        dr = engine.Data(codes, card)
        dr.precount(kmax, r, world)
        for k in range(1, kmax + 1):
            buf = torch.full((kept[k].size,), float("-inf"), dtype=torch.float64, device="cuda")
            dr.score_all_device(k, buf.data_ptr(), shard=r, num_shards=world)
            torch.cuda.synchronize()
            fin = torch.isfinite(buf)
            if bool((fin & torch.isfinite(merged[k])).any()):
                ok = False
                print("  SHARD OVERLAP", seed, k, r, world, flush=True)
            merged[k] = torch.where(fin, buf, merged[k])
        dr.close()
    for k in kept:
        if not np.array_equal(merged[k].cpu().numpy(), kept[k].reshape(-1)):
            ok = False
            print("  SHARD MISMATCH", seed, k, world, flush=True)
    # the pass as one call pair -- obnsl_precount (every size in one launch where the pass is
    # short) + obnsl_score_pass (the sizes share their scorer launches): whole, then in shards
    # with the second-largest count table summed over the ranks as a multi-GPU pass does;
    # bit-identical tables, the per-node records equal to those of the per-size calls
    d.pack()
    d.precount(kmax)
    tabs = [torch.full((kept[k].size,), float("-inf"), dtype=torch.float64, device="cuda") for k in range(1, kmax + 1)]
    ptrs = [t.data_ptr() for t in tabs]
    best = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    d.score_pass(ptrs, kmax - 1, best.data_ptr())
    want_best = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    d.best_parents_device(ptrs, kmax - 1, want_best.data_ptr())
    torch.cuda.synchronize()
    for k in kept:
        if not np.array_equal(tabs[k - 1].cpu().numpy(), kept[k].reshape(-1)):
            ok = False
            print("  PASS-CALL MISMATCH", seed, k, flush=True)
    if not np.array_equal(best.cpu().numpy(), want_best.cpu().numpy()):
        ok = False
        print("  PASS-CALL BEST MISMATCH", seed, flush=True)
    ranks = [engine.Data(codes, card) for _ in range(world)]
    sums = [dr.precount_sharded(kmax, r, world) for r, dr in enumerate(ranks)]
    if sums[0][0]:
        size, cells = sums[0]
        total = torch.zeros(cells, dtype=torch.int32, device="cuda")
        bufs = []
        for dr in ranks:
            b = torch.zeros(cells, dtype=torch.int32, device="cuda")
            dr.counts_export(size, b.data_ptr())
            torch.cuda.synchronize()
            total += b
        for dr in ranks:
            dr.counts_import(size, total.data_ptr())
    merged = {k: torch.full((kept[k].size,), float("-inf"), dtype=torch.float64, device="cuda") for k in kept}
    for r, dr in enumerate(ranks):
        tabs = [torch.full((kept[k].size,), float("-inf"), dtype=torch.float64, device="cuda") for k in range(1, kmax + 1)]
        dr.score_pass([t.data_ptr() for t in tabs], kmax - 1, None, shard=r, num_shards=world)
        torch.cuda.synchronize()
        for k in kept:
            fin = torch.isfinite(tabs[k - 1])
            if bool((fin & torch.isfinite(merged[k])).any()):
                ok = False
                print("  PASS-CALL SHARD OVERLAP", seed, k, r, world, flush=True)
            merged[k] = torch.where(fin, tabs[k - 1], merged[k])
        dr.close()
    for k in kept:
        if not np.array_equal(merged[k].cpu().numpy(), kept[k].reshape(-1)):
            ok = False
            print("  PASS-CALL SHARD MISMATCH", seed, k, world, flush=True)
    bad += not ok
    print(f"seed {seed}: n={n} N={N} kmax={kmax} paths={paths} worst rel {worst:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    d.close()
print("cases", cases, "failed", bad)
sys.exit(1 if bad else 0)
