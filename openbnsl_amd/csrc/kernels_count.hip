// kernels_count.hip -- the kernels around the contraction: row bitmaps from the
// uint8 matrix, one-table and explicit-list counts, and the per-row histogram
// path (the reference's own loop, include/base/contingency_table.h:52-55, one
// workgroup per table and row chunk).
#include <hip/hip_runtime.h>

#include "engine_state.h"

namespace obnsl_internal {
namespace {

// ------------------------------------------------------------------ pack kernel
// One thread builds one 32-row word of every state plane of one variable.
// Bit b of word w is row 32*w + b.  Rows >= N contribute zero bits.
__global__ __launch_bounds__(256) void pack_planes_kernel(
    const uint8_t* __restrict__ codes, uint64_t N, uint64_t stride, uint32_t n,
    const uint32_t* __restrict__ card, const uint32_t* __restrict__ plane_base,
    uint32_t* __restrict__ planes, uint32_t wp, uint32_t ones_plane,
    uint32_t* __restrict__ bad_code, ZeroList zl) {
  if (zl.count) {
    const uint64_t nthreads = (uint64_t)gridDim.x * gridDim.y * 256;
    const uint64_t me = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    for (int b = 0; b < zl.count; ++b)
      for (uint64_t i = me; i < zl.n[b]; i += nthreads) zl.p[b][i] = 0;
  }
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w >= wp) return;
  const uint32_t v = blockIdx.y;
  const uint64_t row0 = (uint64_t)w * 32;
  uint32_t valid = 0;
  if (row0 + 32 <= N) valid = 0xffffffffu;
  else if (row0 < N) valid = (1u << (uint32_t)(N - row0)) - 1u;
  if (v == n) {  // the all-valid-rows plane
    planes[(size_t)ones_plane * wp + w] = valid;
    return;
  }
  const uint4* src = reinterpret_cast<const uint4*>(codes + (size_t)v * stride + row0);
  const uint4 lo = src[0], hi = src[1];
  const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  const uint32_t r = card[v], base = plane_base[v];
  uint32_t covered = 0;  // rows whose code is one of the r states
  for (uint32_t s = 0; s < r; ++s) {
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        word |= (uint32_t)(((x[i] >> (8 * b)) & 0xffu) == s) << (4 * i + b);
    }
    planes[(size_t)(base + s) * wp + w] = word & valid;
    covered |= word;
  }
  // a code >= the declared cardinality: flagged (when asked) with a plain store
  if (bad_code && (covered & valid) != valid) *bad_code = v + 1;
}

// ------------------------------------------------------------------ calibration
// Plain streaming read, 16 bytes per lane, grid-stride: what the memory system
// delivers to a kernel that does nothing else (the achievable figure beside the
// nominal 8 TB/s when a roofline fraction is quoted).  The XOR of everything
// read goes to `sink` so the loads cannot be dropped.
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* __restrict__ src, uint64_t n16,
                                                          uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {  // four loads in flight per lane
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], e = src[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ e.x ^ e.y ^ e.z ^ e.w;
  }
  for (; i < n16; i += stride) {
    const uint4 a = src[i];
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // practically never: keeps the loads alive
}

// ------------------------------------------------------------------ single table
// Dense counts of one explicit subset: counts[cell] += popc(AND of the cell's
// planes) over this block's share of the words.  Generic and simple: it serves
// the one-table calls (ContingencyTable(var_ids, df)), not the batch path.
template <int KCAP>
__global__ __launch_bounds__(256) void count_single_kernel(
    const uint32_t* __restrict__ planes, uint32_t wp, uint32_t k, OneTable ot, uint32_t cells,
    uint32_t* __restrict__ counts, uint32_t ones_plane) {
  const uint32_t* pbase = ot.pbase;
  const uint32_t* cards = ot.cards;
  __shared__ uint32_t red[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t cell = blockIdx.y; cell < cells; cell += gridDim.y) {
    uint32_t pl[KCAP];
    uint32_t rem = cell;
    for (int j = (int)k - 1; j >= 0; --j) {
      pl[j] = pbase[j] + rem % cards[j];
      rem /= cards[j];
    }
    uint32_t acc = 0;
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < wp; w += gridDim.x * 256) {
      uint32_t x = planes[(size_t)ones_plane * wp + w];
      for (uint32_t j = 0; j < k; ++j) x &= planes[(size_t)pl[j] * wp + w];
      acc += __popc(x);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = red[0] + red[1] + red[2] + red[3];
      if (tot) atomicAdd(&counts[cell], tot);
    }
    __syncthreads();
  }
}

// The same for a list of explicit subsets of one size in ONE launch: table
// blockIdx.x of the list (cells off[t] .. off[t+1] of `counts`), blockIdx.y
// strides over its cells, blockIdx.z over chunks of the rows.  The row chunk is
// the slowest grid dimension: the workgroups in flight together read the same
// words of (different) planes, a working set that stays in L2.
template <int KCAP>
__global__ __launch_bounds__(256) void count_list_kernel(
    const uint32_t* __restrict__ planes, uint32_t wp, uint32_t k,
    const uint32_t* __restrict__ pbase_all, const TableDesc* __restrict__ tds,
    uint32_t* __restrict__ counts, uint32_t ones_plane) {
  __shared__ uint32_t red[4];
  const uint64_t t = blockIdx.x;
  const uint32_t* __restrict__ pbase = pbase_all + t * KMAX;
  const uint32_t* __restrict__ cards = tds[t].cards;
  uint32_t cells = 1;
  for (uint32_t j = 0; j < k; ++j) cells *= cards[j];
  uint32_t* __restrict__ dst = counts + tds[t].off;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t cell = blockIdx.y; cell < cells; cell += gridDim.y) {
    uint32_t pl[KCAP];
    uint32_t rem = cell;
    for (int j = (int)k - 1; j >= 0; --j) {
      pl[j] = pbase[j] + rem % cards[j];
      rem /= cards[j];
    }
    uint32_t acc = 0;
    for (uint32_t w = blockIdx.z * 256 + threadIdx.x; w < wp; w += gridDim.z * 256) {
      uint32_t x = planes[(size_t)ones_plane * wp + w];
      for (uint32_t j = 0; j < k; ++j) x &= planes[(size_t)pl[j] * wp + w];
      acc += __popc(x);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = red[0] + red[1] + red[2] + red[3];
      if (tot) atomicAdd(&dst[cell], tot);
    }
    __syncthreads();
  }
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

// ------------------------------------------------------------------ histogram path
// Tables too large for the bit-plane form (its cost grows with the number of
// cells): the reference's own loop (include/base/contingency_table.h:52-55), one
// workgroup per (table, row chunk).  Each lane reads 16 consecutive rows of every
// column of the subset as one 16-byte load (coalesced, column-major), forms the
// mixed-radix keys in registers and counts them in an LDS histogram; the chunk's
// histogram is then added to the table's dense counters.  HBM/L2-bound: k bytes
// per row per table.
template <int KCAP>
__global__ __launch_bounds__(256) void count_hist_kernel(
    const uint8_t* __restrict__ codes, uint64_t stride, uint64_t num_rows, uint32_t n,
    const uint32_t* __restrict__ card, const uint64_t* __restrict__ binom, uint32_t k,
    uint64_t rank0, uint32_t nchunks, uint64_t rows_per_chunk,
    const uint64_t* __restrict__ tbl_off, uint32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
  const uint32_t table = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  // subset of this table from its colex rank
  uint32_t vars[KCAP], wgt[KCAP];
  {
    uint64_t rem = rank0 + table;
#pragma unroll
    for (int j = KCAP; j >= 1; --j) {
      vars[j - 1] = 0;
      if (j > (int)k) continue;
      const uint32_t v = unrank_step(binom, n, (uint32_t)j, rem);
      vars[j - 1] = v;
      rem -= binom[v * (KMAX + 1) + j];
    }
  }
  uint32_t cells = 1;
#pragma unroll
  for (int i = KCAP - 1; i >= 0; --i) {
    wgt[i] = 0;
    if (i >= (int)k) continue;
    wgt[i] = cells;  // the last variable has weight 1
    cells *= card[vars[i]];
  }
  const bool use_lds = cells <= HIST_LDS_CELLS;
  uint32_t* dst = counts + tbl_off[table];
  if (use_lds) {
    for (uint32_t c = threadIdx.x; c < cells; c += 256) hist[c] = 0;
    __syncthreads();
  }
  const uint64_t r0 = (uint64_t)chunk * rows_per_chunk;
  const uint64_t r1 = min(r0 + rows_per_chunk, num_rows);
  for (uint64_t base = r0 + (uint64_t)threadIdx.x * 16; base < r1; base += HIST_ROWS_PER_ITER) {
    uint32_t key[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) key[b] = 0;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
      if (i >= (int)k) continue;
      const uint4 x = *reinterpret_cast<const uint4*>(codes + (size_t)vars[i] * stride + base);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int b = 0; b < 16; ++b) key[b] += ((xs[b >> 2] >> (8 * (b & 3))) & 0xffu) * wgt[i];
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      if (base + b >= r1) continue;
      if (use_lds) atomicAdd(&hist[key[b]], 1u);
      else atomicAdd(&dst[key[b]], 1u);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < cells; c += 256) {
      const uint32_t v = hist[c];
      if (v) atomicAdd(&dst[c], v);
    }
  }
}

}  // namespace

void launch_pack(hipStream_t s, const uint8_t* codes, uint64_t N, uint64_t stride, uint32_t n, const uint32_t* card,
                 const uint32_t* plane_base, uint32_t* planes, uint32_t wp, uint32_t ones_plane, uint32_t* bad_code,
                 const ZeroList& zl) {
  dim3 grid((wp + 255) / 256, n + 1);
  hipLaunchKernelGGL(pack_planes_kernel, grid, dim3(256), 0, s, codes, N, stride, n, card, plane_base, planes, wp,
                     ones_plane, bad_code, zl);
}

void launch_stream_read(hipStream_t s, const uint4* src, uint64_t n16, uint32_t* sink) {
  hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 16), dim3(256), 0, s, src, n16, sink);
}

void launch_count_single(hipStream_t s, const uint32_t* planes, uint32_t wp, uint32_t k, const OneTable& ot,
                         uint32_t cells, uint32_t* counts, uint32_t ones_plane) {
  const uint32_t gx = std::min<uint32_t>((wp + 255) / 256, 64);
  const uint32_t gy = std::min<uint32_t>(cells, 1024);
  if (k <= (uint32_t)KFAST)
    hipLaunchKernelGGL(count_single_kernel<KFAST>, dim3(gx, gy), dim3(256), 0, s, planes, wp, k, ot, cells, counts,
                       ones_plane);
  else
    hipLaunchKernelGGL(count_single_kernel<KMAX>, dim3(gx, gy), dim3(256), 0, s, planes, wp, k, ot, cells, counts,
                       ones_plane);
}

void launch_count_list(hipStream_t s, dim3 grid, const uint32_t* planes, uint32_t wp, uint32_t k,
                       const uint32_t* pbase_all, const TableDesc* tds, uint32_t* counts, uint32_t ones_plane) {
  if (k <= (uint32_t)KFAST)
    hipLaunchKernelGGL(count_list_kernel<KFAST>, grid, dim3(256), 0, s, planes, wp, k, pbase_all, tds, counts,
                       ones_plane);
  else
    hipLaunchKernelGGL(count_list_kernel<KMAX>, grid, dim3(256), 0, s, planes, wp, k, pbase_all, tds, counts,
                       ones_plane);
}

void launch_count_hist(hipStream_t s, uint32_t grid, size_t lds_bytes, const uint8_t* codes, uint64_t stride,
                       uint64_t num_rows, uint32_t n, const uint32_t* card, const uint64_t* binom, uint32_t k,
                       uint64_t rank0, uint32_t nchunks, uint64_t rows_per_chunk, const uint64_t* tbl_off,
                       uint32_t* counts) {
  if (k <= (uint32_t)KFAST)
    hipLaunchKernelGGL(count_hist_kernel<KFAST>, dim3(grid), dim3(256), lds_bytes, s, codes, stride, num_rows, n,
                       card, binom, k, rank0, nchunks, rows_per_chunk, tbl_off, counts);
  else
    hipLaunchKernelGGL(count_hist_kernel<KMAX>, dim3(grid), dim3(256), lds_bytes, s, codes, stride, num_rows, n,
                       card, binom, k, rank0, nchunks, rows_per_chunk, tbl_off, counts);
}

}  // namespace obnsl_internal
