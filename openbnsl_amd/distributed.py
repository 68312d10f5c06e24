"""Multi-GPU plumbing for the sharded hot path: one process per GPU, each rank
scores its shard of the subsets, reduces it to per-node best parent sets, and
the ranks all-gather those small records (n x 3 doubles per rank) over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" on
CPU).  There is no other exchange: the data matrix is replicated."""
import numpy as np


def pack_best(score, size, rank):
    """float64 [n, 3]: score, number of parents, colex rank of the parent set
    (ranks stay far below 2^53, so float64 carries them exactly)."""
    return np.stack([np.asarray(score, dtype=np.float64),
                     np.asarray(size, dtype=np.float64),
                     np.asarray(rank, dtype=np.float64)], axis=1)


def combine_best(parts):
    """parts: float64 [world, n, 3].  Per node: highest score; ties go to fewer
    parents, then to the lower rank (the engine's own tie rule)."""
    parts = np.asarray(parts, dtype=np.float64)
    world, n, _ = parts.shape
    order = np.lexsort((parts[:, :, 2], parts[:, :, 1], -parts[:, :, 0]), axis=0)[0]
    sel = parts[order, np.arange(n)]
    return sel[:, 0].copy(), sel[:, 1].astype(np.uint32), sel[:, 2].astype(np.uint64)


def combine_best_device(parts):
    """torch form of combine_best: parts [world, n, 3] (any device) -> [n, 3]."""
    import torch
    score = parts[:, :, 0]
    best = score.max(dim=0).values
    m = score == best
    inf = torch.full_like(score, float("inf"))
    size = torch.where(m, parts[:, :, 1], inf)
    bsize = size.min(dim=0).values
    m2 = m & (size == bsize)
    brank = torch.where(m2, parts[:, :, 2], inf).min(dim=0).values
    return torch.stack([best, bsize, brank], dim=1)


def allgather_best(score, size, rank, device=None):
    """All-gathers this rank's per-node bests and returns the combined result.
    device: torch device of the exchange buffer (cuda for nccl, None/cpu for gloo)."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(pack_best(score, size, rank))
    if device is not None:
        mine = mine.to(device)
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1).contiguous())
    return combine_best(out.cpu().numpy())


def sum_count_table(table):
    """All-reduce(sum) of a count table of which every rank holds a disjoint
    slice (zero elsewhere): int32 torch tensor, summed in place.  Device tensor
    over nccl; over gloo the sum runs on a host copy."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" or not table.is_cuda:
        dist.all_reduce(table)
        return table
    host = table.cpu()
    dist.all_reduce(host)
    table.copy_(host)
    return table
