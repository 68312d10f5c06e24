"""GPU, several ranks on ONE device: the sharded hot path end to end, with the ranks
splitting the subsets (each counts and scores its shard of every subset size) or the rows
(each counts every size over its rows, one all-reduce of all count tables, each scores its
share).  The ranks exchange over gloo (host tensors; RCCL needs one GPU per rank) and the
combined result must equal the single-rank result and the oracle's."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, sharding):
    import torch.distributed as dist
    import oracle
    from openbnsl_amd import synth
    from openbnsl_amd.pipeline import HotPath
    from helpers import best_parents_reference
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    codes, card, _, _ = synth.random_bn([2, 3, 4, 2, 3, 4, 3, 2, 4, 3, 2], 30_000, seed=21)
    mp = 3
    hot = HotPath(codes, card, mp, rank=rank, world=world, device_index=0, sharding=sharding)
    assert hot.sharding == sharding
    got = None
    for _ in range(2):                       # second pass exercises the cached plans
        got = hot.to_host(hot.step())
    mine = [hot.shard_scores(k) for k in range(1, mp + 2)]
    tables = [oracle.score_all(codes, card, k, nthreads=2) for k in range(1, mp + 2)]
    want = best_parents_reference(tables, len(card))
    ok = (np.max(np.abs(got[0] - want[0]) / np.abs(want[0])) <= 1e-9
          and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]))
    dist.barrier()
    hot.close()
    dist.destroy_process_group()
    q.put((rank, bool(ok), mine))


@pytest.mark.parametrize("sharding", ["subsets", "rows"])
def test_two_ranks_one_gpu(sharding):
    import torch.multiprocessing as mp
    from openbnsl_amd import engine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sharding)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    # the two shards together hold every score of every size exactly once
    n = 11
    for k in range(1, 5):
        assert res[0][2][k - 1] + res[1][2][k - 1] == engine.binomial(n, k) * k
        assert min(res[0][2][k - 1], res[1][2][k - 1]) >= 0


@pytest.mark.parametrize("sharding", ["subsets", "rows"])
def test_bench_exchange_path_over_rccl_with_one_rank(sharding):
    """bench.py with the process group, the all-reduce of the count tables in place in the
    engine's memory, the all-gather (backend nccl = RCCL) and the combine kernel switched on
    in a world of one rank: the code path the multi-GPU runs take, as far as one GPU can
    exercise it, for both ways of splitting a pass."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OBNSL_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "asia8_1e5_bdeu_mp2",
                          "--steps", "3", "--warmup", "1", "--only-hot-path", "--sharding", sharding],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["scores_per_step"] == 232
    assert rec["config"]["sharding"].startswith(sharding)


def test_large_tile_instances_still_count_exactly():
    """The 128 x 128 tile instances of the counting kernel are not the default;
    OBNSL_TILE=128 selects them (read once per process, hence the child process).
    The smoke check compares scores of subset sizes 1..4 and one table with the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OBNSL_TILE="128")
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout


def test_wide_addressing_still_counts_exactly():
    """The 128 x 128 matrix-core kernels address the row bitmaps with 32-bit byte offsets
    where those fit 4 GB (every data set of this suite); OBNSL_WIDE_LOADS=1 selects the
    64-bit form that larger data sets take (read once per process, hence the child process).
    The smoke check compares scores of subset sizes 1..4 and one table with the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OBNSL_WIDE_LOADS="1")
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout


def _full_size_worker(rank, world, port, q, workload, sharding):
    """One rank of a benchmark workload at its own size (ranks share the one GPU,
    gloo on host copies): the sharded pass twice, then rank 0 runs the same pass
    unsharded and compares the combined per-node best parent sets."""
    import torch
    import torch.distributed as dist
    import bench
    from openbnsl_amd import engine
    from openbnsl_amd.pipeline import HotPath
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = bench.WORKLOADS[workload]
    codes, card, _ = bench.make_data(spec)
    kind = engine.BDEU if spec["score"] == "bdeu" else engine.BIC
    mp = spec["max_parents"]
    hot = HotPath(codes, card, mp, kind=kind, rank=rank, world=world, device_index=0, sharding=sharding)
    assert hot.sharding == sharding
    best = None
    for _ in range(2):
        best = hot.step()
    torch.cuda.synchronize()
    got = HotPath.to_host(best)
    covered = torch.tensor([sum(hot.shard_scores(k) for k in range(1, mp + 2))], dtype=torch.int64)
    dist.all_reduce(covered)
    hot.close()
    ok = True
    if rank == 0:
        one = HotPath(codes, card, mp, kind=kind, rank=0, world=1, device_index=0)
        want = HotPath.to_host(one.step())
        torch.cuda.synchronize()
        ok = all((a == b).all() for a, b in zip(got, want)) and int(covered.item()) == one.scores_per_pass
        one.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("workload,world,sharding", [("alarm37_1e6_bdeu_mp3", 2, "subsets"),
                                                     ("alarm37_1e6_bdeu_mp3", 8, "subsets"),
                                                     ("alarm37_1e6_bic_mp4", 4, "subsets"),
                                                     ("alarm37_1e6_bdeu_mp3", 2, "rows"),
                                                     ("alarm37_1e6_bdeu_mp3", 8, "rows"),
                                                     ("alarm37_1e6_bic_mp4", 4, "rows")])
def test_benchmark_workload_sharded_equals_unsharded(workload, world, sharding):
    """Every score covered exactly once over the ranks; combined best parent sets
    identical (scores bit for bit, sizes, ranks) to the unsharded pass."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_full_size_worker, args=(r, world, port, q, workload, sharding)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]

