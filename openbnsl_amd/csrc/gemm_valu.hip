// gemm_valu.hip -- the bit-plane contraction as AND + popcount on the vector ALU
// (OBNSL_GEMM=valu; the matrix-core form in gemm_mfma.hip is the default).  The
// loop it replaces is the reference's row walk,
// include/base/contingency_table.h:52-55.
#include <hip/hip_runtime.h>

#include "engine_state.h"

namespace obnsl_internal {
namespace {

__device__ __forceinline__ uint4 and4(uint4 a, uint4 b) {
  return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w);
}

// acc += popc(x) as ONE instruction.  Written out because the optimiser otherwise
// may regroup a chain of such sums into popc(x) + 0 pairs joined by a three-way
// add: 25 % more vector instructions in a loop that is bound by their issue rate.
__device__ __forceinline__ void popc_acc(uint32_t& acc, uint32_t x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

// One 32-word slab of the tile product: acc[i][j] += popc(A row (tm + 16 i) & B
// column (tn + 16 j)) for i < TT and j < JM.  JM < TT skips the column groups of
// a tile on the edge of the staircase that hold no valid column.
template <int TT, int JM>
__device__ __forceinline__ void tile_mac(const uint32_t* __restrict__ sA,
                                         const uint32_t* __restrict__ sB, uint32_t tm,
                                         uint32_t tn, uint32_t (&acc)[TT][TT]) {
#pragma unroll 1
  for (int kq = 0; kq < KCW / 4; ++kq) {
    uint4 a[TT], b[JM];
#pragma unroll
    for (int i = 0; i < TT; ++i)
      a[i] = *reinterpret_cast<const uint4*>(&sA[(tm + 16 * i) * LDS_STRIDE + kq * 4]);
#pragma unroll
    for (int j = 0; j < JM; ++j)
      b[j] = *reinterpret_cast<const uint4*>(&sB[(tn + 16 * j) * LDS_STRIDE + kq * 4]);
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        popc_acc(acc[i][j], a[i].x & b[j].x);
        popc_acc(acc[i][j], a[i].y & b[j].y);
        popc_acc(acc[i][j], a[i].z & b[j].z);
        popc_acc(acc[i][j], a[i].w & b[j].w);
      }
  }
}

// C[row][col] += sum over this split's words of popc(A[row][w] & B[col][w]).
// 256 threads; a workgroup owns a (16 TT) x (16 TT) tile of C (TT = 8: 128 x 128,
// TT = 4: 64 x 64); thread (tm, tn) owns rows tm + 16 i and columns tn + 16 j
// (i, j < TT), which makes every ds_read_b128 of the compute phase
// bank-conflict free with the 36-word slab stride.
template <int KR, int KC, int TT>
__global__ __launch_bounds__(256, TT == 8 ? 2 : 1) void bitgemm_kernel(
    const uint32_t* __restrict__ planes, uint32_t wp, const RowDesc* __restrict__ rows,
    uint32_t nrows, const ColDesc* __restrict__ cols, uint32_t ncols_total,
    const uint2* __restrict__ items, uint32_t nitems, uint32_t* __restrict__ counts,
    uint32_t steps_total, uint32_t steps_per_split, int use_atomics) {
  constexpr int TB = 16 * TT;   // tile edge
  constexpr int NP = TB / 32;   // staging passes: 256 threads fill 32 slab rows per pass
  // two slabs of A and of B: slab (s & 1) is read while slab ((s + 1) & 1) is
  // filled, one barrier per step.  One array (a second __shared__ object beside
  // it could make the compiler drain loads before every LDS read).
  __shared__ __attribute__((aligned(16))) uint32_t smem[2 * 2 * TB * LDS_STRIDE];

  const uint32_t item_id = blockIdx.x % nitems;
  const uint32_t split = blockIdx.x / nitems;
  const uint2 it = items[item_id];
  const uint32_t row0 = it.x * TB, col0 = it.y * TB;
  const uint32_t tid = threadIdx.x;

  // staging role: 8 threads cover one 32-word slab row with one uint4 each
  const uint32_t q = tid & 7, rl = tid >> 3;
  uint32_t aoff[NP][KR], boff[NP][KC];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const uint32_t r = min(row0 + rl + 32 * p, nrows - 1);
    const uint32_t c = min(col0 + rl + 32 * p, ncols_total - 1);
#pragma unroll
    for (int j = 0; j < KR; ++j) aoff[p][j] = rows[r].plane[j] * wp + q * 4;
#pragma unroll
    for (int j = 0; j < KC; ++j) boff[p][j] = cols[c].plane[j] * wp + q * 4;
  }
  const uint32_t st_off = rl * LDS_STRIDE + q * 4;  // this thread's uint4 in a slab (row rl)

  const uint32_t tn = tid & 15, tm = tid >> 4;
  // rows are ordered by their number of valid columns: the tile's last row has the most
  const uint32_t lim = rows[min(row0 + TB, nrows) - 1].ncols;
  const uint32_t jm = (use_atomics & 2) ? (uint32_t)TT : lim > col0 ? min((lim - col0 + 15) / 16, (uint32_t)TT) : 1;
  uint32_t acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = 0;

  const uint32_t step_lo = split * steps_per_split;
  const uint32_t step_hi = min(step_lo + steps_per_split, steps_total);
  uint4 ga[NP][KR], gb[NP][KC];  // raw plane words of the next slab, in flight during compute

  auto issue_loads = [&](uint32_t step) {
    const uint32_t w0 = step * KCW;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int j = 0; j < KR; ++j)
        ga[p][j] = *reinterpret_cast<const uint4*>(planes + aoff[p][j] + w0);
#pragma unroll
      for (int j = 0; j < KC; ++j)
        gb[p][j] = *reinterpret_cast<const uint4*>(planes + boff[p][j] + w0);
    }
  };
  auto write_slab = [&](uint32_t buf) {
    uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
    uint32_t* sB = sA + TB * LDS_STRIDE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      uint4 va = ga[p][0], vb = gb[p][0];
#pragma unroll
      for (int j = 1; j < KR; ++j) va = and4(va, ga[p][j]);
#pragma unroll
      for (int j = 1; j < KC; ++j) vb = and4(vb, gb[p][j]);
      *reinterpret_cast<uint4*>(&sA[st_off + 32 * p * LDS_STRIDE]) = va;
      *reinterpret_cast<uint4*>(&sB[st_off + 32 * p * LDS_STRIDE]) = vb;
    }
  };

  if (step_lo < step_hi) {
    issue_loads(step_lo);
    write_slab(0);
  }
  __syncthreads();
  for (uint32_t step = step_lo; step < step_hi; ++step) {
    const uint32_t buf = (step - step_lo) & 1;
    const bool more = step + 1 < step_hi;
    if (more) issue_loads(step + 1);
    const uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
    const uint32_t* sB = sA + TB * LDS_STRIDE;
    // only the first jm column groups hold valid columns (uniform over the workgroup)
    if constexpr (TT == 4) {
      switch (jm) {
        case 1: tile_mac<TT, 1>(sA, sB, tm, tn, acc); break;
        case 2: tile_mac<TT, 2>(sA, sB, tm, tn, acc); break;
        case 3: tile_mac<TT, 3>(sA, sB, tm, tn, acc); break;
        default: tile_mac<TT, 4>(sA, sB, tm, tn, acc); break;
      }
    } else {
      switch ((jm + 1) >> 1) {
        case 1: tile_mac<TT, 2>(sA, sB, tm, tn, acc); break;
        case 2: tile_mac<TT, 4>(sA, sB, tm, tn, acc); break;
        case 3: tile_mac<TT, 6>(sA, sB, tm, tn, acc); break;
        default: tile_mac<TT, TT>(sA, sB, tm, tn, acc); break;
      }
    }
    if (more) write_slab(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const uint32_t r = row0 + tm + 16 * i;
    if (r >= nrows) continue;
    const RowDesc rd = rows[r];
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const uint32_t c = col0 + tn + 16 * j;
      if (c >= rd.ncols) continue;
      uint32_t* dst = counts + rd.out_base + (uint64_t)c * rd.M;
      if (use_atomics & 1) {
        if (acc[i][j]) atomicAdd(dst, acc[i][j]);
      } else {
        *dst = acc[i][j];
      }
    }
  }
}

template <int KR, int KC>
void launch_valu(uint32_t tile, hipStream_t s, const GemmOperands& op, const uint2* items, uint32_t nitems,
                 uint32_t grid, uint32_t* counts, uint32_t steps_total, uint32_t steps_per_split, int flags) {
  // the kernel's own flag bits: 1 = add with atomics, 2 = compute every column group
  const int f = ((flags & GEMM_ATOMIC) ? 1 : 0) | ((flags & GEMM_ALL_BLOCKS) ? 2 : 0);
  if (tile == 128)
    hipLaunchKernelGGL((bitgemm_kernel<KR, KC, 8>), dim3(grid), dim3(256), 0, s, op.planes, op.wp, op.rows,
                       op.nrows, op.cols, op.ncols_total, items, nitems, counts, steps_total, steps_per_split, f);
  else
    hipLaunchKernelGGL((bitgemm_kernel<KR, KC, 4>), dim3(grid), dim3(256), 0, s, op.planes, op.wp, op.rows,
                       op.nrows, op.cols, op.ncols_total, items, nitems, counts, steps_total, steps_per_split, f);
}

}  // namespace

void launch_gemm_valu(uint32_t kr, uint32_t kc, uint32_t tile, hipStream_t s, const GemmOperands& op,
                      const uint2* items, uint32_t nitems, uint32_t grid, uint32_t* counts, uint32_t steps_total,
                      uint32_t steps_per_split, int flags) {
#define OBNSL_VALU_CASE(KR, KC) \
  case KR * 10 + KC: launch_valu<KR, KC>(tile, s, op, items, nitems, grid, counts, steps_total, steps_per_split, flags); break
  switch (kr * 10 + kc) {
    OBNSL_VALU_CASE(1, 1);
    OBNSL_VALU_CASE(2, 1);
    OBNSL_VALU_CASE(2, 2);
    OBNSL_VALU_CASE(3, 2);
    OBNSL_VALU_CASE(3, 3);
    OBNSL_VALU_CASE(4, 3);
    OBNSL_VALU_CASE(4, 4);
    default: break;
  }
#undef OBNSL_VALU_CASE
}

}  // namespace obnsl_internal
