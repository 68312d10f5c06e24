"""CPU, world_size 2 over gloo: the multi-GPU step's only exchange -- the
all-gather of per-node best parent sets -- and its deterministic combine.
Each rank holds the bests of its own half of the score entries (taken from the
oracle's score tables); the combined result must equal the global argmax."""
import os
import socket

import numpy as np
import pytest

import oracle
from openbnsl_amd import distributed as obd, engine, synth


def _local_best(tables, n, mask_fn):
    best = {}
    for k, tab in enumerate(tables, start=1):
        subs = oracle.enumerate_subsets(n, k)
        for t, s in enumerate(subs):
            if not mask_fn(k, t):
                continue
            for i, v in enumerate(s):
                ps = [int(x) for x in s if x != v]
                key = (tab[t, i], -(k - 1), -engine.rank(ps))
                if int(v) not in best or key > best[int(v)]:
                    best[int(v)] = key
    sc = np.full(n, -np.inf)
    sz = np.zeros(n, dtype=np.uint32)
    rk = np.zeros(n, dtype=np.uint64)
    for v, (s, negsz, negrk) in best.items():
        sc[v], sz[v], rk[v] = s, -negsz, -negrk
    return sc, sz, rk


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    codes, card, _, _ = synth.random_bn([2, 3, 2, 4, 3, 2, 3], 3_000, seed=9)
    n = 7
    tables = [oracle.score_all(codes, card, k) for k in (1, 2, 3)]
    mine = _local_best(tables, n, lambda k, t: (t + k) % world == rank)
    # the other exchange of a pass: a count table of which each rank holds a slice
    import torch
    whole = oracle.count(codes, card, [0, 2, 3, 5]).astype(np.int64)
    cut = len(whole) // 3
    part = np.zeros_like(whole)
    if rank == 0:
        part[:cut] = whole[:cut]
    else:
        part[cut:] = whole[cut:]
    summed = obd.sum_count_table(torch.from_numpy(part.astype(np.int32)))
    counts_ok = np.array_equal(summed.numpy().astype(np.int64), whole)
    got = obd.allgather_best(*mine)
    want = _local_best(tables, n, lambda k, t: True)
    ok = counts_ok and all(np.array_equal(a, b) for a, b in zip(got, want))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_allgather_best_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_combine_best_tie_rules():
    parts = np.array([
        [[-5.0, 2, 7], [-1.0, 1, 3], [-np.inf, 0, 0]],
        [[-5.0, 1, 9], [-1.0, 1, 2], [-2.0, 3, 1]],
    ])
    sc, sz, rk = obd.combine_best(parts)
    assert sc.tolist() == [-5.0, -1.0, -2.0]
    assert sz.tolist() == [1, 1, 3]          # fewer parents wins the tie on node 0
    assert rk.tolist() == [9, 2, 1]          # lower rank wins the tie on node 1
    import torch
    t = obd.combine_best_device(torch.from_numpy(parts))
    assert t.tolist() == [[-5.0, 1.0, 9.0], [-1.0, 1.0, 2.0], [-2.0, 3.0, 1.0]]
