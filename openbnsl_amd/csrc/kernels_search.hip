// kernels_search.hip -- what consumes the score tables on the device: per-node best
// parent sets over every table (the partial result ranks exchange) and the subset
// dynamic program of exhaustive_search
// (src/structure_learning/exhaustive_search.cpp:75-148).
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>

#include "engine_state.h"

// Scratch of obnsl_best_parents_device_noalloc: the per-node arg-max keys
// [0, n) and ids [n, 2 n) that obnsl_best_parents_device keeps in memory it
// allocates.  One copy per GPU, held until the process ends; every call on a
// GPU shares it.  In host code the array's name is only a handle for
// hipGetSymbolAddress: kernels and copies take the per-device address that
// noalloc_scratch() below returns.
static __device__ __align__(256) unsigned long long scratch[2 * OBNSL_NOALLOC_MAX_VARS];

namespace obnsl_internal {
namespace {

std::mutex g_scratch_mu;
unsigned long long* g_scratch_ptr[MAX_DEVICES] = {nullptr};  // device address of `scratch`, per device
hipEvent_t g_scratch_done[MAX_DEVICES] = {nullptr};          // recorded behind the array's last user
std::mutex g_scratch_user_mu[MAX_DEVICES];                   // one host thread at a time queues a user

// ------------------------------------------------------------------ best parents
__device__ __forceinline__ uint64_t ordered_key(double x) {
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Largest v with C(v, j) <= rank; binom is [(n+1)][KMAX+1].
__device__ __forceinline__ uint32_t unrank_step(const uint64_t* __restrict__ binom, uint32_t n,
                                                uint32_t j, uint64_t rank) {
  uint32_t lo = j - 1, hi = n - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (binom[mid * (KMAX + 1) + j] <= rank) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// pass 0: best_key[v] = max ordered score; pass 1: best_id[v] = min id among
// the entries that reach it, id = (parents << 56) | colex rank of the parent set.
// Every score of a table competes for one of only n addresses, so each
// workgroup first reduces its share in LDS (n <= BEST_LDS_VARS) and then issues
// at most n global atomics.
template <int KCAP>
__global__ __launch_bounds__(256) void best_parents_kernel(
    BestArgs args, uint32_t n, const uint64_t* __restrict__ binom,
    unsigned long long* __restrict__ best_key, unsigned long long* __restrict__ best_id, int pass,
    double* __restrict__ decode_out, unsigned int* __restrict__ done_counter) {
  uint32_t k = 1;
#pragma unroll
  for (int j = 1; j < KMAX; ++j)
    if ((uint32_t)j < args.kmax && blockIdx.x >= args.block_start[j]) k = j + 1;
  const double* __restrict__ scores = nullptr;
  const uint16_t* __restrict__ vtab = nullptr;
  uint64_t nsubsets = 0;
  uint32_t first_block = 0, nblocks = 1;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if ((uint32_t)j + 1 == k) {
      scores = args.scores[j];
      vtab = args.vars[j];
      nsubsets = args.nsubsets[j];
      first_block = args.block_start[j];
      nblocks = args.block_start[j + 1] - args.block_start[j];
    }
  __shared__ unsigned long long s_best[BEST_LDS_VARS];
  const bool use_lds = n <= BEST_LDS_VARS;
  if (use_lds) {
    for (uint32_t v = threadIdx.x; v < n; v += 256) s_best[v] = pass == 0 ? 0ull : ~0ull;
    __syncthreads();
  }
  for (uint64_t t = (uint64_t)(blockIdx.x - first_block) * 256 + threadIdx.x; t < nsubsets;
       t += (uint64_t)nblocks * 256) {
    uint32_t vars[KCAP];
    if (vtab != nullptr) {
      for (uint32_t j = 0; j < k; ++j) vars[j] = vtab[t * k + j];
    } else {
      uint64_t rem = t;
      for (uint32_t j = k; j >= 1; --j) {
        const uint32_t v = unrank_step(binom, n, j, rem);
        vars[j - 1] = v;
        rem -= binom[v * (KMAX + 1) + j];
      }
    }
    for (uint32_t i = 0; i < k; ++i) {
      const double sc = scores[t * k + i];
      if (!(sc > -INFINITY)) continue;  // -inf fill (other shard) or NaN
      const unsigned long long key = ordered_key(sc);
      const uint32_t v = vars[i];
      if (pass == 0) {
        if (use_lds) atomicMax(&s_best[v], key);
        else atomicMax(&best_key[v], key);
      } else if (key == best_key[v]) {
        uint64_t prank = 0;  // colex rank of the parent set S \ {v_i}
        uint32_t pos = 1;
        for (uint32_t j = 0; j < k; ++j) {
          if (j == i) continue;
          prank += binom[vars[j] * (KMAX + 1) + pos];
          ++pos;
        }
        const unsigned long long id = ((unsigned long long)(k - 1) << 56) | prank;
        if (use_lds) atomicMin(&s_best[v], id);
        else atomicMin(&best_id[v], id);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < n; v += 256) {
      const unsigned long long x = s_best[v];
      if (pass == 0) {
        if (x) atomicMax(&best_key[v], x);
      } else if (x != ~0ull) {
        atomicMin(&best_id[v], x);
      }
    }
  }
  // The workgroup that finishes last writes the records (what best_decode_kernel does
  // in a launch of its own): every workgroup's atomics are visible once it has counted
  // itself in; the ids are read past this CU's cache.  The counter is left at zero.
  if (decode_out != nullptr && pass == 1) {
    __shared__ unsigned int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done_counter, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {
      __threadfence();
      for (uint32_t v = threadIdx.x; v < n; v += 256) {
        const unsigned long long kb = __hip_atomic_load(&best_key[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long idv = __hip_atomic_load(&best_id[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kb == 0) {  // nothing scored for this node in this shard
          decode_out[3 * v] = -INFINITY;
          decode_out[3 * v + 1] = 0.0;
          decode_out[3 * v + 2] = 0.0;
        } else {
          const unsigned long long bits = (kb >> 63) ? (kb & 0x7fffffffffffffffull) : ~kb;
          decode_out[3 * v] = __longlong_as_double((long long)bits);
          decode_out[3 * v + 1] = (double)(idv >> 56);
          decode_out[3 * v + 2] = (double)(idv & ((1ull << 56) - 1));
        }
      }
      if (threadIdx.x == 0) *done_counter = 0u;
    }
  }
}

// keys[0..n) = 0 (below every ordered score), ids[0..n) = all ones (above every id)
// Per node, the best of `world` records (score, number of parents, rank of the
// parent set): highest score, then fewer parents, then lower rank -- the rule
// best_parents_kernel applies inside one shard.
__global__ void best_combine_kernel(const double* __restrict__ parts, uint32_t world, uint32_t n,
                                    double* __restrict__ out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  double bs = parts[(size_t)v * 3], bz = parts[(size_t)v * 3 + 1], br = parts[(size_t)v * 3 + 2];
  for (uint32_t w = 1; w < world; ++w) {
    const double* p = parts + ((size_t)w * n + v) * 3;
    const double s = p[0], z = p[1], r = p[2];
    if (s > bs || (s == bs && (z < bz || (z == bz && r < br)))) {
      bs = s;
      bz = z;
      br = r;
    }
  }
  out[(size_t)v * 3] = bs;
  out[(size_t)v * 3 + 1] = bz;
  out[(size_t)v * 3 + 2] = br;
}

// The variables of every size-k subset, by colex rank: out[rank * k + j] (once per data set and size).
__global__ __launch_bounds__(256) void subset_vars_kernel(const uint64_t* __restrict__ binom, uint32_t n, uint32_t k,
                                                          uint64_t nsub, uint16_t* __restrict__ out) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < nsub; t += (uint64_t)gridDim.x * 256) {
    uint64_t rem = t;
    for (uint32_t j = k; j >= 1; --j) {
      const uint32_t v = unrank_step(binom, n, j, rem);
      out[t * k + j - 1] = (uint16_t)v;
      rem -= binom[v * (KMAX + 1) + j];
    }
  }
}

__global__ void best_init_kernel(unsigned long long* keys, unsigned long long* ids, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    keys[i] = 0ull;
    ids[i] = ~0ull;
  }
}

// (ordered key, id) -> (score, number of parents, colex rank of the parent set) as doubles
__global__ void best_decode_kernel(const unsigned long long* __restrict__ keys,
                                   const unsigned long long* __restrict__ ids, uint32_t n,
                                   double* __restrict__ out) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  const unsigned long long kb = keys[v];
  if (kb == 0) {  // nothing scored for this node in this shard
    out[3 * v] = -INFINITY;
    out[3 * v + 1] = 0.0;
    out[3 * v + 2] = 0.0;
    return;
  }
  const unsigned long long bits = (kb >> 63) ? (kb & 0x7fffffffffffffffull) : ~kb;
  out[3 * v] = __longlong_as_double((long long)bits);
  out[3 * v + 1] = (double)(ids[v] >> 56);
  out[3 * v + 2] = (double)(ids[v] & ((1ull << 56) - 1));
}

// ------------------------------------------------------------------ subset DP
// The dynamic program of exhaustive_search
// (src/structure_learning/exhaustive_search.cpp:47-148), one launch per subset
// size: all subsets of one size are independent (they read only entries of
// strictly smaller subsets), so each thread handles one mask S.
//   best-parent table (:75-87): start from the score of the full candidate set
//   (or -inf beyond max_parents); a sub-table entry replaces it when >= (ties go
//   to the smaller set, later parents override earlier ones);
//   best sink (:89-94): strict >, first child wins ties, 0 = uninitialised.
// bls/bps hold, per child c, the 2^(n-1) masks that do not contain c.
__device__ __forceinline__ uint64_t dp_squeeze(uint64_t mask, uint32_t c) {  // drop bit position c
  return (mask & ((1ull << c) - 1)) | ((mask >> (c + 1)) << c);
}
__global__ __launch_bounds__(256) void dp_level_kernel(
    uint32_t n, uint32_t level, DpScores sc, const uint64_t* __restrict__ binom,
    double* __restrict__ best_gs, signed char* __restrict__ best_ch, double* __restrict__ bls,
    uint32_t* __restrict__ bps) {
  const uint64_t combs = 1ull << n, half = combs >> 1;
  for (uint64_t S = (uint64_t)blockIdx.x * 256 + threadIdx.x; S < combs;
       S += (uint64_t)gridDim.x * 256) {
    if ((uint32_t)__popcll(S) != level) continue;
    const bool scored = level <= sc.max_scored;
    const double* row = nullptr;
    if (scored) {
      uint64_t rank = 0;
      uint32_t j = 0;
      for (uint64_t m = S; m; m &= m - 1) {
        ++j;
        rank += binom[(uint32_t)(__ffsll((long long)m) - 1) * (KMAX + 1) + j];
      }
      const double* base = nullptr;
#pragma unroll
      for (int q = 0; q < KMAX; ++q)
        if ((uint32_t)q + 1 == level) base = sc.tbl[q];
      row = base + rank * level;
    }
    double bgs = 0.0;
    int bch = -1;
    uint32_t ci = 0;
    for (uint64_t mc = S; mc; mc &= mc - 1, ++ci) {
      const uint32_t child = (uint32_t)(__ffsll((long long)mc) - 1);
      const uint64_t P = S & ~(1ull << child);
      double best = scored ? row[ci] : -INFINITY;
      uint64_t best_ps = P;
      for (uint64_t mp = P; mp; mp &= mp - 1) {
        const uint32_t pv = (uint32_t)(__ffsll((long long)mp) - 1);
        const uint64_t sub = dp_squeeze(P & ~(1ull << pv), child);
        const double l = bls[child * half + sub];
        if (l >= best) {
          best = l;
          best_ps = bps[child * half + sub];
        }
      }
      const uint64_t slot = child * half + dp_squeeze(P, child);
      bls[slot] = best;
      bps[slot] = (uint32_t)best_ps;
      const double gs = best_gs[P] + best;
      if (bgs >= 0 || gs > bgs) {
        bgs = gs;
        bch = (int)child;
      }
    }
    best_gs[S] = bgs;
    best_ch[S] = (signed char)bch;
  }
}

// Peels sinks off the full set (:132-148): parents[ch] = best parent set of ch
// within the remaining variables.
__global__ void dp_backtrack_kernel(uint32_t n, const double* __restrict__ best_gs,
                                    const signed char* __restrict__ best_ch,
                                    const uint32_t* __restrict__ bps, uint64_t* __restrict__ parents,
                                    double* __restrict__ total) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint64_t combs = 1ull << n, half = combs >> 1;
  uint64_t S = combs - 1;
  total[0] = best_gs[S];
  for (uint32_t v = 0; v < n; ++v) parents[v] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t ch = (uint32_t)best_ch[S];
    const uint64_t P = S & ~(1ull << ch);
    parents[ch] = bps[ch * half + dp_squeeze(P, ch)];
    S = P;
  }
}

}  // namespace

// Once per device, with that device current: where this GPU holds `scratch`.
obnsl_status noalloc_scratch(int device, unsigned long long** out) {
  *out = nullptr;
  if (device < 0 || device >= MAX_DEVICES) return fail(OBNSL_ERR_INVALID_ARGUMENT, "device index out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device) return fail(OBNSL_ERR_RUNTIME, "internal: scratch lookup on a device that is not current");
    unsigned long long* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g_scratch_done[device] = ev;
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return OBNSL_OK;
}

// Users of the array on one GPU are ordered without allocating anything: a user's stream
// first waits for the event recorded behind the previous user (whatever its stream or
// data set), and records it again behind its own last kernel.  begin .. end is a host-side
// critical section per device; the set-up (noalloc_scratch) must have run on the device.
obnsl_status noalloc_user_begin(int device, hipStream_t stream) {
  if (device < 0 || device >= MAX_DEVICES || !g_scratch_done[device])
    return fail(OBNSL_ERR_RUNTIME, "no scratch set up for this device");
  g_scratch_user_mu[device].lock();
  const hipError_t e = hipStreamWaitEvent(stream, g_scratch_done[device], 0);
  if (e != hipSuccess) {
    g_scratch_user_mu[device].unlock();
    return fail(OBNSL_ERR_RUNTIME, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
  }
  return OBNSL_OK;
}
obnsl_status noalloc_user_end(int device, hipStream_t stream) {
  const hipError_t e = hipEventRecord(g_scratch_done[device], stream);
  g_scratch_user_mu[device].unlock();
  if (e != hipSuccess) return fail(OBNSL_ERR_RUNTIME, std::string("hipEventRecord: ") + hipGetErrorString(e));
  return OBNSL_OK;
}

void launch_best_init(hipStream_t s, unsigned long long* keys, unsigned long long* ids, uint32_t n) {
  hipLaunchKernelGGL(best_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, ids, n);
}

void launch_best_parents(hipStream_t s, const BestArgs& args, uint32_t n, const uint64_t* binom,
                         unsigned long long* key, unsigned long long* id, int pass, double* decode_out,
                         unsigned int* done_counter) {
  const uint32_t grid = args.block_start[args.kmax];
  if (args.kmax <= (uint32_t)KFAST)
    hipLaunchKernelGGL(best_parents_kernel<KFAST>, dim3(grid), dim3(256), 0, s, args, n, binom, key, id, pass,
                       decode_out, done_counter);
  else
    hipLaunchKernelGGL(best_parents_kernel<KMAX>, dim3(grid), dim3(256), 0, s, args, n, binom, key, id, pass,
                       decode_out, done_counter);
}

void launch_subset_vars(hipStream_t s, const uint64_t* binom, uint32_t n, uint32_t k, uint64_t nsub, uint16_t* out) {
  const uint32_t grid = (uint32_t)std::min<uint64_t>((nsub + 255) / 256, 4096);
  hipLaunchKernelGGL(subset_vars_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, s, binom, n, k, nsub, out);
}

void launch_best_decode(hipStream_t s, const unsigned long long* keys, const unsigned long long* ids, uint32_t n,
                        double* out) {
  hipLaunchKernelGGL(best_decode_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, ids, n, out);
}

void launch_best_combine(hipStream_t s, const double* parts, uint32_t world, uint32_t n, double* out) {
  hipLaunchKernelGGL(best_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, s, parts, world, n, out);
}

void launch_dp_level(hipStream_t s, uint32_t grid, uint32_t n, uint32_t level, const DpScores& sc,
                     const uint64_t* binom, double* best_gs, signed char* best_ch, double* bls, uint32_t* bps) {
  hipLaunchKernelGGL(dp_level_kernel, dim3(grid), dim3(256), 0, s, n, level, sc, binom, best_gs, best_ch, bls, bps);
}

void launch_dp_backtrack(hipStream_t s, uint32_t n, const double* best_gs, const signed char* best_ch,
                         const uint32_t* bps, uint64_t* parents, double* total) {
  hipLaunchKernelGGL(dp_backtrack_kernel, dim3(1), dim3(64), 0, s, n, best_gs, best_ch, bps, parents, total);
}

}  // namespace obnsl_internal
