"""The whole hot path as one object: what one rank does per pass.

  ranks split the SUBSETS (long passes: many tiles per subset size)
    pack -> count (largest size: this rank's shard; second largest: this rank's
    slice, then all-reduce of that count table; smaller sizes whole) -> for
    k = 1..max_parents+1 score this rank's shard of the size-k subsets ->
    per-node best parent sets of the shard -> (world > 1) all-gather of those
    records and the deterministic combine.

  ranks split the ROWS (short passes: the count tables are small, and a split of
  the subsets would leave every rank with the same fixed part and some without work)
    pack this rank's rows -> count every size over them in one launch -> ONE
    all-reduce of all the count tables -> score this rank's share of the tables ->
    per-node bests -> all-gather and combine as above.

Used by bench.py (timed) and by the multi-rank tests.  PyTorch provides the
device score tables, the stream and torch.distributed; every kernel launched is
the engine's (include/obnsl_engine.h)."""
import numpy as np

from . import distributed as obd
from . import engine


class _DeviceInts:
    """int32 view of `n` device words at `ptr` for torch.as_tensor."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


ROWS_MAX_TABLE_BYTES = 64 << 20    # all count tables of a pass, for one all-reduce
ROWS_MIN_PER_RANK = 32 * 1024      # 32 contraction steps


def row_slice(num_rows, rank, world):
    """[lo, hi) of rank's rows: boundaries on multiples of 2048 rows (64 bitmap words,
    the padding unit of a data set), the last rank takes what is left."""
    def edge(r):
        return num_rows if r == world else min(num_rows, (num_rows * r // world + 1024) // 2048 * 2048)
    return edge(rank), edge(rank + 1)


def count_table_words(card, kmax):
    """uint32 counters of all count tables of subset sizes 1..kmax (states 1..r-1 of every
    variable: the elementary symmetric polynomials of r_v - 1), and the largest full table."""
    e = [1] + [0] * kmax
    for r in card:
        for j in range(kmax, 0, -1):
            e[j] += e[j - 1] * (int(r) - 1)
    top = sorted((int(r) for r in card), reverse=True)[:kmax]
    return sum(e[1:]), int(np.prod(top, dtype=np.int64))


def choose_sharding(card, num_rows, max_parents, world):
    """'rows' or 'subsets': a function of the problem only, so every rank takes the same
    decision without a collective (obnsl_group_* applies the same rule, group.hip).

    Splitting the rows costs one all-reduce of ALL count tables (estimated at 0.02 ms +
    bytes / 150 GB/s; splitting the subsets sums the second-largest table only) and saves
    what a split of the subsets cannot shard or balance: the pack of all rows, the small
    sizes, and the uneven cost of its shards -- put at 0.10 ms + 10 % of a rank's share of
    the counting time (cells x 32-row words x 2.0e-5 ns, the engine's own cost model).
    Measured per-rank times behind the rule (slowest rank, no collective, MI355X):
    37 variables x 10^6 rows, <= 3 parents (3.1 MB of tables), 8 ranks: 0.145 ms rows,
    0.284 subsets; <= 4 parents (38 MB): 0.94 against 1.03; 70 variables x 10^7 rows
    (16 MB): 2.46 against 3.90."""
    kmax = max_parents + 1
    if world <= 1 or kmax > 8 or kmax > len(card):
        return "subsets"
    words, top_cells = count_table_words(card, kmax)
    if top_cells > 4096 or words * 4 > ROWS_MAX_TABLE_BYTES or num_rows // world < ROWS_MIN_PER_RANK:
        return "subsets"
    allreduce_ms = 0.02 + words * 4 / 150e9 * 1e3
    count_ms = words * ((num_rows + 31) // 32) * 2.0e-5 * 1e-6
    return "rows" if allreduce_ms <= 0.10 + 0.10 * count_ms / world else "subsets"


class HotPath:
    def __init__(self, codes, card, max_parents, kind=engine.BDEU, ess=1.0, rank=0, world=1,
                 device_index=0, exchange=None, fused=True, sharding="auto"):
        """exchange: run the all-gather + combine even in a world of one rank
        (default: only when world > 1).  fused: score every subset size and reduce to the
        per-node bests with one obnsl_score_pass call (False: one obnsl_score_all per size
        and obnsl_best_parents_device, the same results).  sharding: 'subsets', 'rows' or
        'auto' (choose_sharding); `codes` is the whole matrix either way -- in 'rows' mode
        only this rank's slice of it goes to the device."""
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        self.mp, self.kind, self.ess = max_parents, kind, ess
        self.fused = bool(fused)
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        total_rows = int(codes.shape[1])
        if sharding == "auto":
            sharding = choose_sharding(card, total_rows, max_parents, world)
        if sharding not in ("subsets", "rows"):
            raise ValueError("sharding must be 'subsets', 'rows' or 'auto'")
        self.sharding = sharding      # (an explicit 'rows' holds in a world of one too: the same calls on all rows)
        if sharding == "rows":
            lo, hi = row_slice(total_rows, rank, world)
            if hi <= lo:
                raise ValueError("fewer rows than ranks can share")
            self.data = engine.Data(codes[:, lo:hi], card, device=device_index, total_rows=total_rows)
            # the engine's own choice of counting path (made from total_rows: the same on every rank)
            if any(self.data.count_path(k) == "histogram" for k in range(1, max_parents + 2)):
                self.data.close()
                self.sharding = sharding = "subsets"
        if sharding == "subsets":
            self.data = engine.Data(codes, card, device=device_index)
        self.data.set_stream(torch.cuda.current_stream().cuda_stream)
        self.n, self.N = self.data.n, total_rows
        # one score table per subset size; entries of other ranks' shards stay -inf
        self.tables = [torch.full((self.data.num_subsets(k) * k,), float("-inf"),
                                  dtype=torch.float64, device=self.dev)
                       for k in range(1, max_parents + 2)]
        self.ptrs = [t.data_ptr() for t in self.tables]
        self.scores_per_pass = sum(self.data.num_subsets(k) * k for k in range(1, max_parents + 2))
        self.best = torch.empty((self.n, 3), dtype=torch.float64, device=self.dev)
        self.exchange = world > 1 if exchange is None else bool(exchange)
        self.gathered = (torch.empty((world, self.n, 3), dtype=torch.float64, device=self.dev)
                         if self.exchange else None)
        self.combined = torch.empty((self.n, 3), dtype=torch.float64, device=self.dev)
        self._count_views = {}
        self._count_buf = None

    def step(self):
        """One pass, queued asynchronously on the current stream (no host
        synchronisation when world == 1 or the backend is nccl).  Returns a device
        tensor [n, 3]: per node the best score, the number of parents and the colex
        rank of the parent set, combined over all ranks when world > 1."""
        torch = self.torch
        d = self.data
        d.pack()
        if self.sharding == "rows":
            ptr, words = d.precount_rows(self.mp + 1)
            if self.exchange and words:
                self._sum_range(ptr, words)
            d.counts_mark_summed(0)
        elif self.world > 1 and self.exchange:
            # the second-largest size is counted in slices too; its table is summed
            # over the ranks before anything is scored from it
            size, cells = d.precount_sharded(self.mp + 1, self.rank, self.world)
            if size:
                self._sum_counts(size, cells)
        else:
            d.precount(self.mp + 1, self.rank, self.world)
            if self.exchange and self.mp >= 1:
                # a world of one rank with the exchange forced on: the same export ->
                # all-reduce -> import round trip (the sum of one table is itself)
                self._sum_counts(self.mp, d.counts_cells(self.mp))
        if self.fused:
            d.score_pass(self.ptrs, self.mp, self.best.data_ptr(), kind=self.kind, ess=self.ess,
                         shard=self.rank, num_shards=self.world)
        else:
            for k in range(1, self.mp + 2):
                d.score_all_device(k, self.ptrs[k - 1], kind=self.kind, ess=self.ess,
                                   shard=self.rank, num_shards=self.world)
            d.best_parents_device(self.ptrs, self.mp, self.best.data_ptr())
        if not self.exchange:
            return self.best
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self.gathered.view(-1), self.best.view(-1))
        else:
            # gloo: host tensors only
            mine = self.best.cpu()
            out = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype)
            dist.all_gather_into_tensor(out.view(-1), mine.view(-1))
            self.gathered.copy_(out)
        d.combine_best_device(self.gathered.data_ptr(), self.world, self.combined.data_ptr())
        return self.combined

    def _view(self, ptr, cells):
        """int32 torch view of engine memory (False where torch cannot alias it)."""
        torch = self.torch
        view = self._count_views.get((ptr, cells))
        if view is None:
            try:
                view = torch.as_tensor(_DeviceInts(ptr, cells), device=self.dev)
                if view.data_ptr() != ptr or view.device != self.dev:
                    view = False
            except Exception:  # noqa: BLE001  (no array-interface support: take the copying path)
                view = False
            self._count_views[(ptr, cells)] = view
        return view

    def _sum_range(self, ptr, words):
        """All-reduce, in place, of the one range that holds every count table."""
        view = self._view(ptr, words)
        if view is False:
            raise RuntimeError("torch cannot alias the engine's count tables on this build")
        obd.sum_count_table(view)

    def _sum_counts(self, size, cells):
        """All-reduce of the engine's count table.  In place when torch can alias
        the table's device memory (CUDA array interface); otherwise through a
        buffer of torch's own and two device copies."""
        torch = self.torch
        ptr, cells = self.data.counts_pointer(size)
        view = self._view(ptr, cells)
        if view is not False:
            obd.sum_count_table(view)   # disjoint slices, zero elsewhere: the sum is the whole table
            self.data.counts_mark_summed(size)
            return
        if self._count_buf is None or self._count_buf.numel() < cells:
            self._count_buf = torch.empty(cells, dtype=torch.int32, device=self.dev)
        buf = self._count_buf[:cells]
        self.data.counts_export(size, buf.data_ptr())
        obd.sum_count_table(buf)
        self.data.counts_import(size, buf.data_ptr())

    @staticmethod
    def to_host(best):
        """(best_score, best_size, best_rank) numpy arrays from step()'s tensor."""
        b = best.detach().cpu().numpy()
        return b[:, 0].copy(), b[:, 1].astype(np.uint32), b[:, 2].astype(np.uint64)

    def shard_scores(self, k):
        """Number of finite entries this rank holds in the size-k table."""
        return int(self.torch.isfinite(self.tables[k - 1]).sum().item())

    def parents_of(self, size, rank):
        return [engine.unrank(int(r), int(s)) for s, r in zip(size, rank)]

    def close(self):
        self.data.close()
