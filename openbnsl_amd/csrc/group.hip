// group.hip -- one host process driving several GPUs (include/obnsl_engine.h,
// obnsl_group_*).
//
// The reference is a single-threaded, single-device program
// (src/structure_learning/exhaustive_search.cpp:47-130 has no parallel loop).
// Here the work of a pass is split over the devices of a group; rank r
// (= devices[r]) holds a replica of the data matrix and works on its own
// stream.  One pass, ranks splitting the SUBSETS:
//
//   every rank : pack -> count (largest size: its shard; second largest: its
//                slice; smaller sizes whole)                       [asynchronous]
//   exchange 1 : all-reduce(sum, uint32) of the second-largest count table
//   every rank : score its shard of every size -> per-node bests of the shard
//   exchange 2 : all-gather of the per-node records (3 doubles per node)
//   every rank : deterministic combine (score, then fewer parents, then rank)
//
// A pass whose count tables (sizes 1..max_parents+1, every size on the bit-plane
// contraction) are cheap to all-reduce next to its counting time (rows_pass below; at most
// 64 MiB, at least 32768 rows per rank) splits the ROWS instead: a
// split of the subsets leaves every rank the same fixed part -- the pack of all rows,
// the small sizes, launch latencies -- and ranks without a tile of the largest size.
//
//   every rank : pack ITS rows -> count every size over them (one launch)
//   exchange 1 : ONE all-reduce(sum, uint32) of all the count tables
//   every rank : score its share of every size's tables -> as above
//
// Every rank has a host thread of its own (its current HIP device set once, to the
// rank's): the launches of a phase are queued for all ranks at the same time instead
// of device after device from the caller's thread, whose current device is never
// touched.  The exchanges go through RCCL (ncclCommInitAll over the group's devices, one
// communicator per rank, group calls from the caller's thread).  RCCL is opened at
// run time (dlopen), so the library loads where it is absent and binds to the
// copy a host program already holds (PyTorch ships its own).  RCCL refuses a
// device that appears twice; such a group -- several ranks sharing one GPU, which
// is how a box with one GPU rehearses the sharded pass -- and a group of one
// rank exchange by device copies and an add kernel instead ("copy" exchange; it
// can also be forced with OBNSL_GROUP_EXCHANGE=copy).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"
#include "engine_state.h"
#include "obnsl_engine.h"

namespace {

using obnsl_internal::fail;

// Restores the caller's current HIP device when a group call returns: every call
// below moves through the group's devices with hipSetDevice.
struct DeviceGuard {
  int saved = -1;
  DeviceGuard() {
    if (hipGetDevice(&saved) != hipSuccess) saved = -1;
  }
  ~DeviceGuard() {
    if (saved >= 0) (void)hipSetDevice(saved);
  }
};

// ---- the few RCCL entry points used, bound at run time
typedef void* nccl_comm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok() const { return lib != nullptr; }
};
// rccl.h: ncclSum = 0, ncclMax = 2; ncclUint32 = 3, ncclFloat64 = 8
constexpr int kSum = 0, kMax = 2, kU32 = 3, kF64 = 8;

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    // the copy the process already holds first (PyTorch's), then the system's
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    if (x.CommInitAll && x.CommDestroy && x.AllReduce && x.AllGather && x.GroupStart && x.GroupEnd)
      x.lib = h;
    return x;
  }();
  return r;
}

__global__ void add_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    dst[i] += src[i];
}
__global__ void max_f64_kernel(double* __restrict__ dst, const double* __restrict__ src, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    dst[i] = fmax(dst[i], src[i]);
}
__global__ void fill_f64_kernel(double* __restrict__ dst, double v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    dst[i] = v;
}

// A rank's host thread: runs one task at a time with the rank's device current.
struct Worker {
  std::thread thread;
  std::mutex mu;
  std::condition_variable cv;
  std::function<obnsl_status()> task;
  bool has_task = false, done = true, stop = false;
  obnsl_status result = OBNSL_OK;
  std::string error;
  double busy_ms = 0.0;   // time the last task took on the host

  void start(int device) {
    thread = std::thread([this, device] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> lock(mu);
      for (;;) {
        cv.wait(lock, [this] { return has_task || stop; });
        if (stop) return;
        has_task = false;
        std::function<obnsl_status()> fn = std::move(task);
        lock.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        const obnsl_status st = fn();
        const std::string msg = st == OBNSL_OK ? std::string() : std::string(obnsl_last_error());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lock.lock();
        result = st;
        error = msg;
        busy_ms = ms;
        done = true;
        cv.notify_all();
      }
    });
  }
  void post(std::function<obnsl_status()> fn) {
    std::lock_guard<std::mutex> lock(mu);
    task = std::move(fn);
    has_task = true;
    done = false;
    cv.notify_all();
  }
  obnsl_status wait(std::string* msg) {
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [this] { return done; });
    if (result != OBNSL_OK && msg) *msg = error;
    return result;
  }
  void shutdown() {
    if (!thread.joinable()) return;
    {
      std::lock_guard<std::mutex> lock(mu);
      stop = true;
      cv.notify_all();
    }
    thread.join();
  }
};

struct Rank {
  int device = 0;
  std::unique_ptr<Worker> worker;
  obnsl_data* data = nullptr;
  obnsl_data* rows = nullptr;    // this rank's slice of the rows (passes that split the rows)
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  nccl_comm_t comm = nullptr;
  std::vector<double*> tables;   // [k-1]: C(n,k) * k doubles, device
  std::vector<uint64_t> table_len;
  double* best = nullptr;        // n * 3
  double* gathered = nullptr;    // world * n * 3
  double* combined = nullptr;    // n * 3
  uint32_t* stage = nullptr;     // copy exchange across devices: a peer's table lands here
  uint64_t stage_words = 0;
};

}  // namespace

struct obnsl_group {
  uint32_t n = 0;
  uint64_t N = 0;
  std::vector<uint32_t> card;
  bool no_rows = false;     // OBNSL_GROUP_SHARDING=subsets
  bool force_rows = false;  // OBNSL_GROUP_SHARDING=rows (wherever the engine can serve it)
  int rows_served[OBNSL_MAX_SUBSET + 1];   // by kmax: -1 not asked yet, 0 a size is counted per row, 1 yes
  obnsl_group() { std::fill(rows_served, rows_served + OBNSL_MAX_SUBSET + 1, -1); }
  std::vector<Rank> ranks;
  bool use_rccl = false;
  uint32_t tables_for = 0;  // score tables are allocated for subset sizes 1..tables_for
  double enqueue_ms = 0.0;  // host time of the per-rank phases of the last call (all ranks in parallel)
  ~obnsl_group() {
    for (Rank& r : ranks)
      if (r.worker) r.worker->shutdown();
    for (Rank& r : ranks) {
      (void)hipSetDevice(r.device);
      if (r.stream) (void)hipStreamSynchronize(r.stream);
      if (r.comm && rccl().ok()) (void)rccl().CommDestroy(r.comm);
      if (r.rows) (void)obnsl_data_destroy(r.rows);
      if (r.data) (void)obnsl_data_destroy(r.data);
      for (double* t : r.tables)
        if (t) (void)hipFree(t);
      if (r.best) (void)hipFree(r.best);
      if (r.gathered) (void)hipFree(r.gathered);
      if (r.combined) (void)hipFree(r.combined);
      if (r.stage) (void)hipFree(r.stage);
      if (r.ready) (void)hipEventDestroy(r.ready);
      if (r.done) (void)hipEventDestroy(r.done);
      if (r.stream) (void)hipStreamDestroy(r.stream);
    }
  }
};

namespace {

// Runs fn(r) on every rank's own thread, all at once; the first failure wins (its
// message becomes the caller's last error).
obnsl_status on_ranks(obnsl_group* g, const std::function<obnsl_status(uint32_t)>& fn) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t r = 0; r < g->ranks.size(); ++r) g->ranks[r].worker->post([&fn, r] { return fn(r); });
  obnsl_status first = OBNSL_OK;
  std::string first_msg;
  for (Rank& rk : g->ranks) {
    std::string msg;
    const obnsl_status st = rk.worker->wait(&msg);
    if (st != OBNSL_OK && first == OBNSL_OK) {
      first = st;
      first_msg = msg;
    }
  }
  g->enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return first == OBNSL_OK ? OBNSL_OK : fail(first, first_msg);
}

obnsl_status nccl_check(int rc, const char* what) {
  if (rc == 0) return OBNSL_OK;
  const char* msg = rccl().GetErrorString ? rccl().GetErrorString(rc) : "error";
  return fail(OBNSL_ERR_RUNTIME, std::string(what) + ": " + msg);
}

// One collective per rank between ncclGroupStart and ncclGroupEnd, from this one
// thread.  The group is always ended, also when a call inside it fails (a thread
// left inside an open group defers every later collective); the first error wins.
template <class PerRank>
obnsl_status grouped(size_t world, PerRank per_rank) {
  OB_TRY(nccl_check(rccl().GroupStart(), "ncclGroupStart"));
  obnsl_status first = OBNSL_OK;
  std::string first_msg;
  for (size_t r = 0; r < world && first == OBNSL_OK; ++r) {
    first = per_rank(r);
    if (first != OBNSL_OK) first_msg = obnsl_last_error();
  }
  const obnsl_status end = nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  if (first != OBNSL_OK) return fail(first, first_msg);
  return end;
}

obnsl_status ensure_tables(obnsl_group* g, uint32_t kmax) {
  if (kmax <= g->tables_for) return OBNSL_OK;
  for (Rank& r : g->ranks) {
    HIP_TRY(hipSetDevice(r.device));
    r.tables.resize(kmax, nullptr);
    r.table_len.resize(kmax, 0);
    for (uint32_t k = g->tables_for + 1; k <= kmax; ++k) {
      const uint64_t len = obnsl_binomial(g->n, k) * k;
      r.table_len[k - 1] = len;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.tables[k - 1]), std::max<uint64_t>(len, 1) * sizeof(double)));
    }
  }
  g->tables_for = kmax;
  return OBNSL_OK;
}

// rank 0's stream waits for everything queued on the other ranks' streams, or
// the other way round
obnsl_status fan_in(obnsl_group* g) {
  for (size_t r = 1; r < g->ranks.size(); ++r) {
    HIP_TRY(hipSetDevice(g->ranks[r].device));
    HIP_TRY(hipEventRecord(g->ranks[r].ready, g->ranks[r].stream));
    HIP_TRY(hipSetDevice(g->ranks[0].device));
    HIP_TRY(hipStreamWaitEvent(g->ranks[0].stream, g->ranks[r].ready, 0));
  }
  return OBNSL_OK;
}
obnsl_status fan_out(obnsl_group* g) {
  HIP_TRY(hipSetDevice(g->ranks[0].device));
  HIP_TRY(hipEventRecord(g->ranks[0].done, g->ranks[0].stream));
  for (size_t r = 1; r < g->ranks.size(); ++r) {
    HIP_TRY(hipSetDevice(g->ranks[r].device));
    HIP_TRY(hipStreamWaitEvent(g->ranks[r].stream, g->ranks[0].done, 0));
  }
  return OBNSL_OK;
}

// Sum of the ranks' uint32 tables (`words` each, at ptr[r]), left in every rank's table.
obnsl_status all_reduce_sum_u32(obnsl_group* g, const std::vector<uint32_t*>& ptr, uint64_t words) {
  const size_t W = g->ranks.size();
  if (W == 1 && !g->use_rccl) return OBNSL_OK;
  if (g->use_rccl) {
    return grouped(W, [&](size_t r) {
      return nccl_check(rccl().AllReduce(ptr[r], ptr[r], words, kU32, kSum, g->ranks[r].comm, g->ranks[r].stream),
                        "ncclAllReduce");
    });
  }
  // copy exchange: rank 0 adds the others' tables into its own, then hands the sum back
  OB_TRY(fan_in(g));
  Rank& r0 = g->ranks[0];
  HIP_TRY(hipSetDevice(r0.device));
  for (size_t r = 1; r < W; ++r) {
    const uint32_t* src = ptr[r];
    if (g->ranks[r].device != r0.device) {
      if (r0.stage_words < words) {
        if (r0.stage) HIP_TRY(hipFree(r0.stage));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r0.stage), words * sizeof(uint32_t)));
        r0.stage_words = words;
      }
      HIP_TRY(hipMemcpyPeerAsync(r0.stage, r0.device, ptr[r], g->ranks[r].device, words * sizeof(uint32_t), r0.stream));
      src = r0.stage;
    }
    hipLaunchKernelGGL(add_u32_kernel, dim3(1024), dim3(256), 0, r0.stream, ptr[0], src, words);
    HIP_TRY(hipGetLastError());
  }
  for (size_t r = 1; r < W; ++r)
    HIP_TRY(hipMemcpyPeerAsync(ptr[r], g->ranks[r].device, ptr[0], r0.device, words * sizeof(uint32_t), r0.stream));
  return fan_out(g);
}

// Every rank's `count` doubles at send[r] -> all of them, rank after rank, at recv[r].
obnsl_status all_gather_f64(obnsl_group* g, const std::vector<double*>& send, const std::vector<double*>& recv,
                            uint64_t count) {
  const size_t W = g->ranks.size();
  if (g->use_rccl) {
    return grouped(W, [&](size_t r) {
      return nccl_check(rccl().AllGather(send[r], recv[r], count, kF64, g->ranks[r].comm, g->ranks[r].stream),
                        "ncclAllGather");
    });
  }
  OB_TRY(fan_in(g));
  Rank& r0 = g->ranks[0];
  HIP_TRY(hipSetDevice(r0.device));
  for (size_t r = 0; r < W; ++r)
    HIP_TRY(hipMemcpyPeerAsync(recv[0] + r * count, r0.device, send[r], g->ranks[r].device, count * sizeof(double),
                               r0.stream));
  for (size_t r = 1; r < W; ++r)
    HIP_TRY(hipMemcpyPeerAsync(recv[r], g->ranks[r].device, recv[0], r0.device, W * count * sizeof(double), r0.stream));
  return fan_out(g);
}

// Element-wise maximum of the ranks' double tables, left in every rank's table
// (score tables whose foreign entries are -inf: the maximum is the whole table).
obnsl_status all_reduce_max_f64(obnsl_group* g, const std::vector<double*>& ptr, uint64_t count) {
  const size_t W = g->ranks.size();
  if (W == 1 && !g->use_rccl) return OBNSL_OK;
  if (g->use_rccl) {
    return grouped(W, [&](size_t r) {
      return nccl_check(rccl().AllReduce(ptr[r], ptr[r], count, kF64, kMax, g->ranks[r].comm, g->ranks[r].stream),
                        "ncclAllReduce");
    });
  }
  OB_TRY(fan_in(g));
  Rank& r0 = g->ranks[0];
  HIP_TRY(hipSetDevice(r0.device));
  for (size_t r = 1; r < W; ++r) {
    const double* src = ptr[r];
    if (g->ranks[r].device != r0.device) {
      const uint64_t words = count * 2;
      if (r0.stage_words < words) {
        if (r0.stage) HIP_TRY(hipFree(r0.stage));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r0.stage), words * sizeof(uint32_t)));
        r0.stage_words = words;
      }
      HIP_TRY(hipMemcpyPeerAsync(r0.stage, r0.device, ptr[r], g->ranks[r].device, count * sizeof(double), r0.stream));
      src = reinterpret_cast<const double*>(r0.stage);
    }
    hipLaunchKernelGGL(max_f64_kernel, dim3(1024), dim3(256), 0, r0.stream, ptr[0], src, count);
    HIP_TRY(hipGetLastError());
  }
  for (size_t r = 1; r < W; ++r)
    HIP_TRY(hipMemcpyPeerAsync(ptr[r], g->ranks[r].device, ptr[0], r0.device, count * sizeof(double), r0.stream));
  return fan_out(g);
}

// Whether a pass over subset sizes 1..kmax splits the rows: a function of the problem
// only (the same rule as openbnsl_amd/pipeline.py's choose_sharding).
bool rows_pass(const obnsl_group* g, uint32_t kmax) {
  const uint64_t W = g->ranks.size();
  if (W <= 1 || g->no_rows || kmax > 8 || kmax > g->n) return false;
  std::vector<uint64_t> e(kmax + 1, 0);   // elementary symmetric polynomials of r_v - 1
  e[0] = 1;
  for (uint32_t r : g->card)
    for (uint32_t j = kmax; j >= 1; --j) {
      e[j] += e[j - 1] * (r - 1);
      if (e[j] > (1ull << 40)) return false;
    }
  uint64_t words = 0;
  for (uint32_t j = 1; j <= kmax; ++j) words += e[j];
  std::vector<uint32_t> sorted(g->card);
  std::sort(sorted.rbegin(), sorted.rend());
  uint64_t top_cells = 1;
  for (uint32_t j = 0; j < kmax; ++j) top_cells *= sorted[j];
  if (top_cells > 4096) return false;
  if (g->force_rows) return g->N / W >= 1;
  if (words * 4 > (64ull << 20) || g->N / W < 32 * 1024) return false;
  // one all-reduce of all count tables (0.02 ms + bytes / 150 GB/s) against what a split of
  // the subsets cannot shard or balance (0.10 ms + 10 % of a rank's share of the counting
  // time, cells x 32-row words x 2.0e-5 ns)
  const double allreduce_ms = 0.02 + (double)words * 4.0 / 150e9 * 1e3;
  const double count_ms = (double)words * (double)((g->N + 31) / 32) * 2.0e-5 * 1e-6;
  return allreduce_ms <= 0.10 + 0.10 * count_ms / (double)W;
}

// The slice data sets, made on first use from each rank's resident copy of the matrix
// (boundaries on multiples of 2048 rows: 64 bitmap words).
obnsl_status ensure_row_slices(obnsl_group* g) {
  const uint64_t W = g->ranks.size();
  if (g->ranks[0].rows) return OBNSL_OK;
  auto edge = [&](uint64_t r) -> uint64_t {
    if (r == W) return g->N;
    return std::min<uint64_t>(g->N, (g->N * r / W + 1024) / 2048 * 2048);
  };
  return on_ranks(g, [&](uint32_t r) -> obnsl_status {
    Rank& rk = g->ranks[r];
    const uint64_t lo = edge(r), hi = edge(r + 1);
    if (hi <= lo) return fail(OBNSL_ERR_INVALID_ARGUMENT, "fewer rows than the group's ranks can share");
    OB_TRY(obnsl_data_create(rk.data->codes.p + lo, hi - lo, rk.data->stride, g->n, g->card.data(), rk.device,
                             /*codes_on_device=*/1, &rk.rows));
    OB_TRY(obnsl_data_set_stream(rk.rows, rk.stream));
    return obnsl_data_set_total_rows(rk.rows, g->N);
  });
}

// pack (optional) -> sharded counting -> exchange 1 -> sharded scoring of sizes 1..kmax
// (and, with `best`, each rank's per-node records of its shard)
obnsl_status score_pass(obnsl_group* g, obnsl_score_kind kind, double ess, uint32_t kmax, int repack, bool best) {
  const uint32_t W = (uint32_t)g->ranks.size();
  g->enqueue_ms = 0.0;
  OB_TRY(ensure_tables(g, kmax));
  bool rows = rows_pass(g, kmax);
  if (rows) {
    OB_TRY(ensure_row_slices(g));
    if (g->rows_served[kmax] < 0) {
      // (the engine's own choice of counting path, made from the rows of the whole data set)
      g->rows_served[kmax] = 1;
      for (uint32_t j = 1; j <= kmax; ++j) {
        int path = OBNSL_PATH_HISTOGRAM;
        OB_TRY(obnsl_count_path(g->ranks[0].rows, j, &path));
        if (path == OBNSL_PATH_HISTOGRAM) g->rows_served[kmax] = 0;
      }
    }
    rows = g->rows_served[kmax] == 1;
  }
  if (rows) {
    std::vector<uint32_t*> ptr(W, nullptr);
    std::vector<uint64_t> words(W, 0);
    OB_TRY(on_ranks(g, [&](uint32_t r) -> obnsl_status {
      Rank& rk = g->ranks[r];
      if (repack) OB_TRY(obnsl_data_pack(rk.rows));
      for (uint32_t k = 1; k <= kmax; ++k) {
        hipLaunchKernelGGL(fill_f64_kernel, dim3(512), dim3(256), 0, rk.stream, rk.tables[k - 1], -INFINITY,
                           rk.table_len[k - 1]);
        HIP_TRY(hipGetLastError());
      }
      return obnsl_precount_rows(rk.rows, kmax, &ptr[r], &words[r]);
    }));
    for (uint32_t r = 1; r < W; ++r)
      if (words[r] != words[0]) return fail(OBNSL_ERR_RUNTIME, "internal: ranks disagree on the tables to sum");
    if (words[0]) OB_TRY(all_reduce_sum_u32(g, ptr, words[0]));
    for (uint32_t r = 0; r < W; ++r) OB_TRY(obnsl_counts_mark_summed(g->ranks[r].rows, 0));
    return on_ranks(g, [&](uint32_t r) -> obnsl_status {
      Rank& rk = g->ranks[r];
      return obnsl_score_pass(rk.rows, kind, ess, kmax - 1, r, W, rk.tables.data(), best ? rk.best : nullptr);
    });
  }
  std::vector<uint32_t> sum_size(W, 0);
  std::vector<uint64_t> sum_cells(W, 0);
  OB_TRY(on_ranks(g, [&](uint32_t r) -> obnsl_status {
    Rank& rk = g->ranks[r];
    if (repack) OB_TRY(obnsl_data_pack(rk.data));
    // entries of the other ranks' shards stay -inf
    for (uint32_t k = 1; k <= kmax; ++k) {
      hipLaunchKernelGGL(fill_f64_kernel, dim3(512), dim3(256), 0, rk.stream, rk.tables[k - 1], -INFINITY,
                         rk.table_len[k - 1]);
      HIP_TRY(hipGetLastError());
    }
    return obnsl_precount_sharded(rk.data, kmax, r, W, &sum_size[r], &sum_cells[r]);
  }));
  for (uint32_t r = 1; r < W; ++r)
    if (sum_size[r] != sum_size[0] || sum_cells[r] != sum_cells[0])
      return fail(OBNSL_ERR_RUNTIME, "internal: ranks disagree on the table to sum");
  if (sum_size[0]) {
    std::vector<uint32_t*> ptr(W);
    OB_TRY(on_ranks(g, [&](uint32_t r) -> obnsl_status {
      uint64_t cells = 0;
      return obnsl_counts_pointer(g->ranks[r].data, sum_size[0], &ptr[r], &cells);
    }));
    OB_TRY(all_reduce_sum_u32(g, ptr, sum_cells[0]));
    for (uint32_t r = 0; r < W; ++r) OB_TRY(obnsl_counts_mark_summed(g->ranks[r].data, sum_size[0]));
  }
  return on_ranks(g, [&](uint32_t r) -> obnsl_status {
    Rank& rk = g->ranks[r];
    return obnsl_score_pass(rk.data, kind, ess, kmax - 1, r, W, rk.tables.data(), best ? rk.best : nullptr);
  });
}

}  // namespace

extern "C" {

obnsl_status obnsl_group_create(const uint8_t* codes, uint64_t num_rows, uint64_t col_stride,
                                uint32_t num_vars, const uint32_t* cardinalities, const int* devices,
                                uint32_t num_devices, obnsl_group** out) {
  DeviceGuard device_guard;
  if (!out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!devices || num_devices == 0 || num_devices > 64)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "a group needs between 1 and 64 devices");
  auto g = std::make_unique<obnsl_group>();
  g->n = num_vars;
  g->N = num_rows;
  if (!cardinalities) return fail(OBNSL_ERR_INVALID_ARGUMENT, "empty data set");
  g->card.assign(cardinalities, cardinalities + num_vars);
  if (const char* sh = std::getenv("OBNSL_GROUP_SHARDING")) {
    g->no_rows = std::strcmp(sh, "subsets") == 0;
    g->force_rows = std::strcmp(sh, "rows") == 0;
  }
  g->ranks.resize(num_devices);
  std::set<int> distinct;
  for (uint32_t r = 0; r < num_devices; ++r) {
    Rank& rk = g->ranks[r];
    rk.device = devices[r];
    rk.worker = std::make_unique<Worker>();
    rk.worker->start(rk.device);
    distinct.insert(devices[r]);
    OB_TRY(obnsl_data_create(codes, num_rows, col_stride, num_vars, cardinalities, devices[r], 0, &rk.data));
    HIP_TRY(hipSetDevice(rk.device));
    HIP_TRY(hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&rk.ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&rk.done, hipEventDisableTiming));
    OB_TRY(obnsl_data_set_stream(rk.data, rk.stream));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rk.best), (size_t)num_vars * 3 * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rk.gathered), (size_t)num_devices * num_vars * 3 * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rk.combined), (size_t)num_vars * 3 * sizeof(double)));
  }
  const char* mode = std::getenv("OBNSL_GROUP_EXCHANGE");
  const bool want_copy = mode && std::strcmp(mode, "copy") == 0;
  const bool want_rccl = mode && std::strcmp(mode, "rccl") == 0;
  const bool can_rccl = distinct.size() == num_devices;   // RCCL refuses a device listed twice
  if (want_rccl && !can_rccl)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "OBNSL_GROUP_EXCHANGE=rccl needs every device of the group to be distinct");
  if (!want_copy && can_rccl && (num_devices > 1 || want_rccl)) {
    if (!rccl().ok()) return fail(OBNSL_ERR_RUNTIME, "librccl could not be opened for a multi-device group");
    std::vector<nccl_comm_t> comms(num_devices, nullptr);
    OB_TRY(nccl_check(rccl().CommInitAll(comms.data(), (int)num_devices, devices), "ncclCommInitAll"));
    for (uint32_t r = 0; r < num_devices; ++r) g->ranks[r].comm = comms[r];
    g->use_rccl = true;
  } else {
    // copy exchange between distinct devices reads a peer's memory
    for (uint32_t r = 1; r < num_devices; ++r)
      if (g->ranks[r].device != g->ranks[0].device) {
        HIP_TRY(hipSetDevice(g->ranks[0].device));
        (void)hipDeviceEnablePeerAccess(g->ranks[r].device, 0);   // already enabled is fine
        HIP_TRY(hipSetDevice(g->ranks[r].device));
        (void)hipDeviceEnablePeerAccess(g->ranks[0].device, 0);
        (void)hipGetLastError();
      }
  }
  *out = g.release();
  return OBNSL_OK;
}

obnsl_status obnsl_group_destroy(obnsl_group* g) {
  DeviceGuard device_guard;
  delete g;
  return OBNSL_OK;
}

obnsl_status obnsl_group_info(const obnsl_group* g, uint32_t* num_devices, int* uses_rccl) {
  if (!g) return fail(OBNSL_ERR_INVALID_ARGUMENT, "group is null");
  if (num_devices) *num_devices = (uint32_t)g->ranks.size();
  if (uses_rccl) *uses_rccl = g->use_rccl ? 1 : 0;
  return OBNSL_OK;
}

obnsl_status obnsl_group_host_time(const obnsl_group* g, double* enqueue_ms) {
  if (!g || !enqueue_ms) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  *enqueue_ms = g->enqueue_ms;
  return OBNSL_OK;
}

obnsl_status obnsl_group_best_parents(obnsl_group* g, obnsl_score_kind kind, double ess,
                                      uint32_t max_parents, int repack, double* best_score,
                                      uint32_t* best_size, uint64_t* best_rank) {
  DeviceGuard device_guard;
  if (!g || !best_score || !best_size || !best_rank) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (max_parents + 1 > g->n) return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  const uint32_t W = (uint32_t)g->ranks.size(), kmax = max_parents + 1;
  OB_TRY(score_pass(g, kind, ess, kmax, repack, true));
  std::vector<double*> send(W), recv(W);
  for (uint32_t r = 0; r < W; ++r) {
    send[r] = g->ranks[r].best;
    recv[r] = g->ranks[r].gathered;
  }
  OB_TRY(all_gather_f64(g, send, recv, (uint64_t)g->n * 3));
  OB_TRY(on_ranks(g, [&](uint32_t r) -> obnsl_status {
    Rank& rk = g->ranks[r];
    return obnsl_combine_best_device(rk.data, rk.gathered, W, rk.combined);
  }));
  // every rank holds the same combined record; read rank 0's, wait for all
  std::vector<double> h((size_t)g->n * 3);
  HIP_TRY(hipSetDevice(g->ranks[0].device));
  HIP_TRY(hipMemcpyAsync(h.data(), g->ranks[0].combined, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                         g->ranks[0].stream));
  for (uint32_t r = 0; r < W; ++r) {
    HIP_TRY(hipSetDevice(g->ranks[r].device));
    HIP_TRY(hipStreamSynchronize(g->ranks[r].stream));
  }
  for (uint32_t v = 0; v < g->n; ++v) {
    best_score[v] = h[(size_t)v * 3];
    best_size[v] = (uint32_t)h[(size_t)v * 3 + 1];
    best_rank[v] = (uint64_t)h[(size_t)v * 3 + 2];
  }
  return OBNSL_OK;
}

obnsl_status obnsl_group_score_tables(obnsl_group* g, obnsl_score_kind kind, double ess,
                                      uint32_t max_parents, int repack, double* const* tables_out) {
  DeviceGuard device_guard;
  if (!g || !tables_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (max_parents + 1 > g->n) return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  const uint32_t W = (uint32_t)g->ranks.size(), kmax = max_parents + 1;
  OB_TRY(score_pass(g, kind, ess, kmax, repack, false));
  for (uint32_t k = 1; k <= kmax; ++k) {
    std::vector<double*> ptr(W);
    for (uint32_t r = 0; r < W; ++r) ptr[r] = g->ranks[r].tables[k - 1];
    OB_TRY(all_reduce_max_f64(g, ptr, g->ranks[0].table_len[k - 1]));
  }
  Rank& r0 = g->ranks[0];
  HIP_TRY(hipSetDevice(r0.device));
  for (uint32_t k = 1; k <= kmax; ++k)
    if (tables_out[k - 1])
      HIP_TRY(hipMemcpyAsync(tables_out[k - 1], r0.tables[k - 1], r0.table_len[k - 1] * sizeof(double),
                             hipMemcpyDeviceToHost, r0.stream));
  for (uint32_t r = 0; r < W; ++r) {
    HIP_TRY(hipSetDevice(g->ranks[r].device));
    HIP_TRY(hipStreamSynchronize(g->ranks[r].stream));
  }
  return OBNSL_OK;
}

obnsl_status obnsl_group_exhaustive_search(obnsl_group* g, obnsl_score_kind kind, double ess,
                                           uint32_t max_parents, uint64_t* parents_out,
                                           double* total_score_out) {
  DeviceGuard device_guard;
  if (!g || !parents_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  // reference: exhaustive_search.cpp:18-23
  if (max_parents > g->n - 1) return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  if (g->n > 64) return fail(OBNSL_ERR_INVALID_ARGUMENT, "num_of_vars must be in [0, 64]");
  const uint32_t W = (uint32_t)g->ranks.size(), kmax = max_parents + 1;
  OB_TRY(score_pass(g, kind, ess, kmax, 0, false));
  for (uint32_t k = 1; k <= kmax; ++k) {
    std::vector<double*> ptr(W);
    for (uint32_t r = 0; r < W; ++r) ptr[r] = g->ranks[r].tables[k - 1];
    OB_TRY(all_reduce_max_f64(g, ptr, g->ranks[0].table_len[k - 1]));
  }
  // the subset dynamic program does not shard: rank 0 runs it on the whole tables
  Rank& r0 = g->ranks[0];
  HIP_TRY(hipSetDevice(r0.device));
  obnsl_status st = obnsl_exhaustive_search_tables(r0.data, r0.tables.data(), max_parents, parents_out, total_score_out);
  for (uint32_t r = 1; r < W; ++r) {
    HIP_TRY(hipSetDevice(g->ranks[r].device));
    HIP_TRY(hipStreamSynchronize(g->ranks[r].stream));
  }
  return st;
}

}  // extern "C"
