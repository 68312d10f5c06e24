// engine.hip -- MI355X (gfx950) local-score engine behind include/obnsl_engine.h.
//
// What the reference does per candidate (child, parent set): walk all N rows,
// build a sparse joint histogram of the subset (include/base/contingency_table.h
// :52-55), then evaluate BDeu from it (include/score/local_score.h:49-60).
//
// How this engine does it.  The uint8 data matrix is turned once into row
// bitmaps: one N-bit plane per (variable, state).  A cell count of a joint
// table is then popcount(AND of the cell's planes), and all tables of one subset
// size k are one GEMM-shaped contraction over the N rows of the data: the k
// variables of a subset are split into the kc smallest ("column side") and the
// kr = k - kc largest ("row side"); rows of the GEMM are the joint states of
// every kr-subset, columns the joint states of every kc-subset, and
// C[row][col] = sum over samples of A[row][sample] AND B[col][sample].  Only
// combinations with max(column vars) < min(row vars) are computed (rows are
// sorted by their smallest variable, so the valid region of each row tile is a
// prefix of the column list).  Only the joint states in which every variable is
// in one of its states 1..r-1 are counted; the scorer recovers the others by
// inclusion-exclusion from the tables of the sub-subsets.
//
// The contraction runs on the matrix cores (bitgemm_mfma_kernel): a sample bit
// becomes an FP4 nibble by masking, v_mfma_scale_f32_32x32x64_f8f6f4 multiplies
// and adds 32 x 32 x 64 of them per instruction, and the f32 accumulators hold
// exact integer counts.  A workgroup owns a 128 x 128 tile of C (256 x 256 in
// launches of very many tiles) and stages 32-word slabs of A and B through LDS
// (rows formed on the fly by ANDing the subset's planes).  A launch of few tiles
// is cut along the contraction into thousands of work units; their partial
// counts meet in row-major images of the tiles (coalesced atomics) that
// image_scatter_kernel moves into the count table.  bitgemm_kernel is the same
// contraction as AND + popcount on the vector ALU (OBNSL_GEMM=valu; 64 x 64
// tiles).  Tables with many cells, and subsets of more than 8 variables, are
// counted per row instead (count_hist_kernel).  Scores come from one functional
// per table, G(S) - G(P), evaluated in fp64 by score_reduced_kernel, two or four
// tables per wavefront.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "engine_internal.h"
#include "obnsl_engine.h"

// Scratch of obnsl_best_parents_device_noalloc: the per-node arg-max keys
// [0, n) and ids [n, 2 n) that obnsl_best_parents_device keeps in memory it
// allocates.  One copy per GPU, held until the process ends; every call on a
// GPU shares it.  In host code the array's name is only a handle for
// hipGetSymbolAddress: kernels and copies take the per-device address below.
static __device__ __align__(256) unsigned long long scratch[2 * OBNSL_NOALLOC_MAX_VARS];

namespace {

constexpr int KMAX = OBNSL_MAX_SUBSET;  // largest subset size (per-row counting path)
constexpr int KFAST = 8;                // largest subset size of the bit-plane contraction (2 KSIDE)
constexpr int KSIDE = 4;                // largest kr / kc
constexpr int KCW = 32;                 // 32-bit words of the contraction per LDS slab
constexpr int LDS_STRIDE = KCW + 4;     // padded slab row (words): conflict-free b128 reads
constexpr uint64_t BATCH_CELL_CAP = 1ull << 30;  // counters per batch (4 GiB)
constexpr uint64_t REDUCED_MAX_CELLS = 4096;     // largest full table the reduced-mode scorer expands in LDS

thread_local std::string g_err;

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned long long* g_scratch_ptr[MAX_DEVICES] = {nullptr};  // device address of `scratch`, per device

obnsl_status fail(obnsl_status st, const std::string& msg) {
  g_err = msg;
  return st;
}

}  // namespace

obnsl_status obnsl_internal::fail(obnsl_status st, const std::string& msg) { return ::fail(st, msg); }

namespace {

// Which contraction kernel counts the tables: the matrix-core form (default) or
// the vector-ALU and/popcount form (OBNSL_GEMM=valu; kept for comparison runs).
bool use_mfma_gemm() {
  static const bool on = [] {
    const char* e = std::getenv("OBNSL_GEMM");
    return !(e && std::strcmp(e, "valu") == 0);
  }();
  return on;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(OBNSL_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------------ descriptors
struct RowDesc {        // one row of A: a joint state of a kr-subset
  uint32_t plane[KSIDE];  // plane index per row-side variable (unused: ONES plane)
  uint64_t out_base;      // counter index of (this row, column 0)
  uint32_t M;             // rows in this row group = counter stride per column
  uint32_t ncols;         // valid columns: those of column groups below min(row vars)
};
struct ColDesc {        // one row of B: a joint state of a kc-subset
  uint32_t plane[KSIDE];
};
struct RowGroupDesc {   // scorer view of one kr-subset
  uint32_t vars[KSIDE];
  uint64_t out_off;     // first counter of the group: layout [column][M]
  uint64_t rank_part;   // sum_j C(u_j, kc + j): row-side part of the colex rank
  uint64_t tbl_start;   // tables of earlier groups in this batch
  uint32_t M;
  uint32_t ncolgroups;  // C(u_1, kc) column groups pair with this row group
};
struct ColGroupDesc {   // scorer view of one kc-subset
  uint32_t vars[KSIDE];
  uint32_t col_before;  // columns of earlier column groups
  uint32_t cells;
};
struct TableDesc {      // explicit table for the list-form / single-table scorer
  uint64_t off;
  uint32_t k;
  uint32_t cards[KMAX];
};

// Count tables that the pack kernel clears on the side (the split-contraction
// launches add into them with atomics and would otherwise each need a fill first).
constexpr int ZERO_MAX = 8;
struct ZeroList {
  uint32_t* p[ZERO_MAX];
  uint64_t n[ZERO_MAX];
  int count;
};

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

// ------------------------------------------------------------------ bit-GEMM
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

// ------------------------------------------------------------------ bit-GEMM on the matrix cores
// The same contraction with v_mfma_scale_f32_32x32x64_f8f6f4 on FP4 (e2m1)
// operands.  A 32-bit word of 32 sample bits is used as eight FP4 nibbles four
// times over: w & 0x11111111, w & 0x22222222, w & 0x44444444 are words whose
// nibbles are 0 or 0b0001 / 0b0010 / 0b0100 = 0.5 / 1.0 / 2.0 in e2m1, and
// (w >> 3) & 0x11111111 brings the fourth bit of every nibble (the e2m1 sign)
// down to 0.5 as well.  The block scales (E8M0, one byte per operand) of each of
// the four instructions make the product of two set bits exactly 1.0, so the f32
// accumulator holds the integer count, exact while it stays below 2^24 (the
// host bounds the samples per workgroup accordingly).  The contraction is a sum
// over samples: which sample sits at which K position does not matter as long as
// A and B agree, and both are built by the same code from the same words.
//
// One instruction does 32 x 32 x 64 cell-samples in ~16 ns per SIMD (measured),
// 6.5 x the and/popcount pair stream of the vector ALU.  Operand build: 5 vector
// instructions per 32 rows x 32 samples of an operand, amortised over the
// wave's column blocks.
//
// A workgroup (4 waves) owns a 128 x 128 tile of C; wave w owns rows 32 w .. 32 w
// + 31 and the tile's first JM 32-column blocks (JM < 4 on the staircase edge:
// every wave then does the same, smaller, amount of work).  LDS holds the
// ANDed sample words exactly as in the vector-ALU kernel; a lane reads 16 bytes
// (128 samples of its row) per operand and round.
typedef int obnsl_v8i __attribute__((ext_vector_type(8)));
typedef float obnsl_v16f __attribute__((ext_vector_type(16)));

template <int J>
__device__ __forceinline__ obnsl_v8i fp4_operand(const uint4 w) {
  obnsl_v8i f;
  if constexpr (J < 3) {
    constexpr uint32_t m = 0x11111111u << J;
    f[0] = (int)(w.x & m);
    f[1] = (int)(w.y & m);
    f[2] = (int)(w.z & m);
    f[3] = (int)(w.w & m);
  } else {
    f[0] = (int)((w.x >> 3) & 0x11111111u);
    f[1] = (int)((w.y >> 3) & 0x11111111u);
    f[2] = (int)((w.z >> 3) & 0x11111111u);
    f[3] = (int)((w.w >> 3) & 0x11111111u);
  }
  f[4] = f[5] = f[6] = f[7] = 0;
  return f;
}

template <int J>
__device__ __forceinline__ obnsl_v16f fp4_mfma(const obnsl_v8i a, const obnsl_v8i b, const obnsl_v16f c) {
  // E8M0 scale bytes: 0.5-valued nibbles (J = 0, 3) x 2, 1.0 (J = 1) x 1, 2.0 (J = 2) x 1/2
  constexpr int sc = (J == 1) ? 0x7f7f7f7f : (J == 2) ? 0x7e7e7e7e : (int)0x80808080u;
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, sc, 0, sc);
}

// One (rd, J) block of a wave's NA x NB blocks: the NA A operands, then per
// column block its B operand and NA instructions.  The scheduling fences keep
// the compiler from building the operands of all four J up front (80 live
// registers, one wave per SIMD less).
template <int NA, int NB, int J>
__device__ __forceinline__ void mfma_block(const uint4 (&wa)[NA], const uint4 (&wb)[NB],
                                           obnsl_v16f (&acc)[NA][NB]) {
  obnsl_v8i a[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = fp4_operand<J>(wa[i]);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const obnsl_v8i b = fp4_operand<J>(wb[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i][j] = fp4_mfma<J>(a[i], b, acc[i][j]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// One slab (KCW words per row) of a wave's NA x NB blocks: A rows arow + 32 i, B
// rows brow + 32 j.  Operand cost is (NA + NB) builds per NA * NB instructions:
// 1 x 4 (a wave per 32 rows, every column block: equal work for every wave when
// the tile's last column blocks are empty) costs 6.25 vector instructions per
// MFMA, 2 x 2 (full tiles) 5 -- which is what lets a wave that is alone on its
// SIMD, while its neighbour stages or waits at the barrier, keep the matrix pipe fed.
template <int NA, int NB, class AfterRound>
__device__ __forceinline__ void tile_mfma(const uint32_t* __restrict__ sA,
                                          const uint32_t* __restrict__ sB, uint32_t arow,
                                          uint32_t brow, uint32_t hoff, obnsl_v16f (&acc)[NA][NB],
                                          AfterRound after_round) {
  // the words of round rd + 1 are read from LDS while round rd's instructions run
  const uint32_t* pa = &sA[arow * LDS_STRIDE + hoff];
  const uint32_t* pb = &sB[brow * LDS_STRIDE + hoff];
  uint4 wa[NA], wb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE);
#pragma unroll
  for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE);
  static_assert(KCW / 8 == 4, "two iterations of two rounds");
  // two rounds per iteration: the words of a round live in (wa, wb) and (na, nb) in turn
#pragma unroll 1
  for (int rp = 0; rp < 2; ++rp) {
    uint4 na[NA], nb[NB];
    {
      const int nx = (2 * rp + 1) * 8;
#pragma unroll
      for (int i = 0; i < NA; ++i) na[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + nx);
#pragma unroll
      for (int j = 0; j < NB; ++j) nb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + nx);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block<NA, NB, 0>(wa, wb, acc);
      mfma_block<NA, NB, 1>(wa, wb, acc);
      mfma_block<NA, NB, 2>(wa, wb, acc);
      mfma_block<NA, NB, 3>(wa, wb, acc);
      after_round(2 * rp);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      // (the last round reads its own words again: one address computation for every round)
      const int nx = min(2 * rp + 2, KCW / 8 - 1) * 8;
#pragma unroll
      for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + nx);
#pragma unroll
      for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + nx);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block<NA, NB, 0>(na, nb, acc);
      mfma_block<NA, NB, 1>(na, nb, acc);
      mfma_block<NA, NB, 2>(na, nb, acc);
      mfma_block<NA, NB, 3>(na, nb, acc);
      after_round(2 * rp + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Software pipelined form for NA = 2 (full tiles): the operands of block J + 1 are built
// between the instructions of block J (scheduling groups: one MFMA, then a share of the
// vector instructions), so a wave's matrix instructions are not held up by its own builds.
template <int J, int NA, int NB>
__device__ __forceinline__ void build_blocks(const uint4 (&wa)[NA], const uint4 (&wb)[NB], obnsl_v8i (&a)[NA],
                                             obnsl_v8i (&b)[NB]) {
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = fp4_operand<J>(wa[i]);
#pragma unroll
  for (int j = 0; j < NB; ++j) b[j] = fp4_operand<J>(wb[j]);
}
template <int J, int NA, int NB>
__device__ __forceinline__ void mfma_blocks(const obnsl_v8i (&a)[NA], const obnsl_v8i (&b)[NB],
                                            obnsl_v16f (&acc)[NA][NB]) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i][j] = fp4_mfma<J>(a[i], b[j], acc[i][j]);
}
template <int NA, int NB, class AfterRound>
__device__ __forceinline__ void tile_mfma_pipelined(const uint32_t* __restrict__ sA,
                                                    const uint32_t* __restrict__ sB, uint32_t arow,
                                                    uint32_t brow, uint32_t hoff, obnsl_v16f (&acc)[NA][NB],
                                                    AfterRound after_round) {
  const uint32_t* pa = &sA[arow * LDS_STRIDE + hoff];
  const uint32_t* pb = &sB[brow * LDS_STRIDE + hoff];
  uint4 wa[NA], wb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE);
#pragma unroll
  for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE);
  obnsl_v8i oa[NA], ob[NB];
  build_blocks<0, NA, NB>(wa, wb, oa, ob);
  static_assert(KCW / 8 == 4, "two iterations of two rounds");
  constexpr int VPM = (8 * (NA + NB) + NA * NB - 1) / (NA * NB);   // vector instructions per MFMA slot
  // two rounds per iteration: the words of a round live in (wa, wb) and (na, nb) in
  // turn, so no round ends with copying one set into the other
#pragma unroll 1
  for (int rp = 0; rp < 2; ++rp) {
    obnsl_v8i qa[NA], qb[NB];
    uint4 na[NA], nb[NB];
#define OBNSL_PIPE(J, JN, SRCA, SRCB)                                  \
    mfma_blocks<J, NA, NB>(oa, ob, acc);                               \
    build_blocks<JN, NA, NB>(SRCA, SRCB, qa, qb);                      \
    _Pragma("unroll") for (int q = 0; q < NA * NB; ++q) {              \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);             \
    }                                                                  \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) oa[i] = qa[i];      \
    _Pragma("unroll") for (int j = 0; j < NB; ++j) ob[j] = qb[j];
    {  // round 2 rp: its words are in (wa, wb); the next round's go to (na, nb)
      const int nx = (2 * rp + 1) * 8;
#pragma unroll
      for (int i = 0; i < NA; ++i) na[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + nx);
#pragma unroll
      for (int j = 0; j < NB; ++j) nb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + nx);
      OBNSL_PIPE(0, 1, wa, wb)
      OBNSL_PIPE(1, 2, wa, wb)
      OBNSL_PIPE(2, 3, wa, wb)
      OBNSL_PIPE(3, 0, na, nb)
      after_round(2 * rp);
    }
    {  // round 2 rp + 1: its words are in (na, nb); the next round's go to (wa, wb)
      // (the last round reads its own words again: one address computation for every round)
      const int nx = min(2 * rp + 2, KCW / 8 - 1) * 8;
#pragma unroll
      for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + nx);
#pragma unroll
      for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + nx);
      OBNSL_PIPE(0, 1, na, nb)
      OBNSL_PIPE(1, 2, na, nb)
      OBNSL_PIPE(2, 3, na, nb)
      OBNSL_PIPE(3, 0, wa, wb)
      after_round(2 * rp + 1);
    }
#undef OBNSL_PIPE
  }
}

constexpr int MFMA_TILE = 128;        // tile edge of the matrix-core kernel (256 for launches of many tiles)
constexpr uint64_t IMAGE_MAX_BYTES = 256ull << 20;  // tile images of one split launch (4096 tiles)
constexpr size_t BIG_TILE_FROM = 16384;  // 128 x 128 tiles from which a launch takes 256 x 256 tiles
constexpr uint32_t MFMA_MAX_STEPS = (1u << 24) / (KCW * 32);  // samples per workgroup < 2^24

// Everything after the tile is known, for one value of JM (a template
// parameter of the whole loop nest, so that the accumulators live in the
// accumulator registers from the first step to the epilogue).
template <int KR, int KC, int NA, int NB, int TB>
__device__ __forceinline__ void mfma_tile_body(
    uint32_t* __restrict__ smem, const uint32_t* __restrict__ planes, uint32_t wp,
    const RowDesc* __restrict__ rows, uint32_t nrows, const ColDesc* __restrict__ cols,
    uint32_t ncols_total, uint32_t row0, uint32_t col0, uint32_t* __restrict__ counts,
    uint32_t step_lo, uint32_t step_hi, bool atomics, uint32_t* __restrict__ image, bool probe_noload,
    bool probe_sameload = false, int probe_x = 0) {
  constexpr int PR = TB / 4;   // slab rows per staging pass: 2 TB threads, 8 per row
  constexpr int NP = TB / PR;  // staging passes
  static_assert(NP == KCW / 8, "one staging pass per round of the contraction step");
  const uint32_t tid = threadIdx.x;
  // Staging role: 8 threads cover one 32-word slab row with one uint4 each, TB / 4 rows
  // per pass.  The plane offsets of the tile's rows and columns sit in LDS behind
  // the slabs (a pass is chosen by a loop counter: no register array to index).
  const uint32_t q = tid & 7, rl = tid >> 3;
  uint32_t* sAo = smem + 2 * 2 * TB * LDS_STRIDE;
  uint32_t* sBo = sAo + TB * KR;
  if (TB != 128 && tid < (uint32_t)TB) {
    const uint32_t r = min(row0 + tid, nrows - 1);
    const uint32_t c = min(col0 + tid, ncols_total - 1);
#pragma unroll
    for (int j = 0; j < KR; ++j) sAo[tid * KR + j] = rows[r].plane[j] * wp;
#pragma unroll
    for (int j = 0; j < KC; ++j) sBo[tid * KC + j] = cols[c].plane[j] * wp;
  }
  const uint32_t st_off = rl * LDS_STRIDE + q * 4;
  // the epilogue's view of this thread's row (read now: no dependent load after the last step)
  uint64_t e_base = 0;
  uint32_t e_M = 0, e_nc = 0;   // 0 columns for a row past the end: nothing is written
  if (tid < (uint32_t)TB && row0 + tid < nrows) {
    e_base = rows[row0 + tid].out_base;
    e_M = rows[row0 + tid].M;
    e_nc = rows[row0 + tid].ncols;
  }

  // compute role: wave wv (of TB / 32) owns NA x NB blocks of 32 x 32 -- NA = 1: rows
  // 32 wv .., the tile's first NB column blocks; NA = 2: rows 64 (wv >> 1) .., columns
  // 32 NB (wv & 1) .. (a 64 x 64 quadrant of a 128 tile, a 64 x 128 eighth of a 256
  // tile).  Lane (lr, h) holds row / column lr of each block, samples 128 h ..
  // 128 h + 127 of each 256-sample round.
  static_assert(NA == 1 || 2 * 32 * NB == TB, "two waves side by side cover the tile's columns");
  const uint32_t lane = tid & 63, wv = tid >> 6;
  const uint32_t lr = lane & 31, h4 = (lane >> 5) * 4;
  const uint32_t wrow = NA == 1 ? wv * 32 : (wv >> 1) * 64;   // first row / column of the wave's blocks
  const uint32_t wcol = NA == 1 ? 0 : (wv & 1) * (32 * NB);
  obnsl_v16f acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][j][i] = 0.0f;

  if constexpr (TB == 128) {
    // The whole next slab is loaded into registers at the start of a step and ANDed
    // and stored into the other LDS buffer at its end (plane offsets in registers,
    // straight from the descriptors: a short unit does not pay a trip through LDS).
    uint4 ga[NP][KR], gb[NP][KC];
    uint32_t aoff[NP][KR], boff[NP][KC];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint32_t r = min(row0 + rl + PR * p, nrows - 1);
      const uint32_t c = min(col0 + rl + PR * p, ncols_total - 1);
#pragma unroll
      for (int j = 0; j < KR; ++j) aoff[p][j] = rows[r].plane[j] * wp + q * 4;
#pragma unroll
      for (int j = 0; j < KC; ++j) boff[p][j] = cols[c].plane[j] * wp + q * 4;
    }
    auto issue_loads = [&](uint32_t step) {
      const uint32_t w0 = (probe_sameload ? step_lo : step) * KCW;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int j = 0; j < KR; ++j) ga[p][j] = *reinterpret_cast<const uint4*>(planes + aoff[p][j] + w0);
#pragma unroll
        for (int j = 0; j < KC; ++j) gb[p][j] = *reinterpret_cast<const uint4*>(planes + boff[p][j] + w0);
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
        *reinterpret_cast<uint4*>(&sA[st_off + PR * p * LDS_STRIDE]) = va;
        *reinterpret_cast<uint4*>(&sB[st_off + PR * p * LDS_STRIDE]) = vb;
      }
    };
    if (step_lo < step_hi) {
      issue_loads(step_lo);
      write_slab(0);
    }
    __syncthreads();
    for (uint32_t step = step_lo; step < step_hi; ++step) {
      const uint32_t buf = (step - step_lo) & 1;
      const bool more = step + 1 < step_hi && !probe_noload;
      if (more && !(probe_x & 2)) issue_loads(step + 1);
      const uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
      const uint32_t* sB = sA + TB * LDS_STRIDE;
      if constexpr (NA == 2 && NB == 2) tile_mfma_pipelined<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {});
      else tile_mfma<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {});
      if (more && !(probe_x & 1)) write_slab(buf ^ 1);
      __syncthreads();
    }
  } else {
    __syncthreads();   // the offsets
    // One pass of the next slab is in flight at a time: its global loads are issued
    // after a round's instructions, and ANDed and stored into the other LDS buffer
    // after the next round's (16 staging registers instead of 64: what lets a wave
    // hold 2 x 4 blocks of accumulators at two waves per SIMD).
    uint4 ga[KR], gb[KC];
    auto issue_pass = [&](uint32_t step, int p) {
      const uint32_t w0 = (probe_sameload ? step_lo : step) * KCW + q * 4;
      const uint32_t r = rl + PR * p;
#pragma unroll
      for (int j = 0; j < KR; ++j) ga[j] = *reinterpret_cast<const uint4*>(planes + sAo[r * KR + j] + w0);
#pragma unroll
      for (int j = 0; j < KC; ++j) gb[j] = *reinterpret_cast<const uint4*>(planes + sBo[r * KC + j] + w0);
    };
    auto write_pass = [&](uint32_t buf, int p) {
      uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
      uint32_t* sB = sA + TB * LDS_STRIDE;
      uint4 va = ga[0], vb = gb[0];
#pragma unroll
      for (int j = 1; j < KR; ++j) va = and4(va, ga[j]);
#pragma unroll
      for (int j = 1; j < KC; ++j) vb = and4(vb, gb[j]);
      *reinterpret_cast<uint4*>(&sA[st_off + PR * p * LDS_STRIDE]) = va;
      *reinterpret_cast<uint4*>(&sB[st_off + PR * p * LDS_STRIDE]) = vb;
    };
    if (step_lo < step_hi) {
#pragma unroll 1
      for (int p = 0; p < NP; ++p) {
        issue_pass(step_lo, p);
        write_pass(0, p);
      }
    }
    __syncthreads();
    for (uint32_t step = step_lo; step < step_hi; ++step) {
      const uint32_t buf = (step - step_lo) & 1;
      const bool more = step + 1 < step_hi && !probe_noload;
      if (more && !(probe_x & 2)) issue_pass(step + 1, 0);
      const uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
      const uint32_t* sB = sA + TB * LDS_STRIDE;
      auto stage_round = [&](int rd) {
        if (more) {
          if (!(probe_x & 1)) write_pass(buf ^ 1, rd);
          if (rd + 1 < NP && !(probe_x & 2)) issue_pass(step + 1, rd + 1);
        }
      };
      if constexpr (NA == 2) tile_mfma_pipelined<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, stage_round);
      else tile_mfma<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, stage_round);
      __syncthreads();
    }
  }

  // A tile whose contraction is cut into several units adds its partial counts into
  // a row-major TB x TB image of the tile (32 neighbouring columns of a row per
  // half-wave and instruction: whole cache lines per atomic request, where the count
  // table itself would be hit at a stride of M cells); image_scatter_kernel moves
  // the finished images into the table.
  if (image != nullptr) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t rt = wrow + 32 * a + (i & 3) + 8 * (i >> 2) + h4;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t v = (uint32_t)acc[a][j][i];
          if (v && !(probe_x & 4)) atomicAdd(image + rt * TB + wcol + 32 * j + lr, v);
        }
      }
    return;
  }
  // The slabs are dead now (the loop ends on a barrier): park the tile's row
  // descriptors in LDS, one per thread, for the lanes that hold each row's counts.
  uint64_t* s_base = reinterpret_cast<uint64_t*>(smem);
  uint32_t* s_M = smem + 2 * TB;
  uint32_t* s_nc = smem + 3 * TB;
  if (tid < (uint32_t)TB) {
    s_base[tid] = e_base;
    s_M[tid] = e_M;
    s_nc[tid] = e_nc;
  }
  __syncthreads();
  // C/D layout of the 32 x 32 forms: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t rt = wrow + 32 * a + (i & 3) + 8 * (i >> 2) + h4;
      const uint32_t nc = s_nc[rt], M = s_M[rt];
      const uint64_t base = s_base[rt];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const uint32_t c = col0 + wcol + 32 * j + lr;
        if (c >= nc) continue;
        const uint32_t v = (uint32_t)acc[a][j][i];
        if ((probe_x & 4) && v != 0xffffffffu) continue;   // OBNSL_PROBE_NOSTORE: nothing is written
        uint32_t* dst = counts + base + (uint64_t)c * M;
        if (atomics) {
          if (v) atomicAdd(dst, v);
        } else {
          *dst = v;
        }
      }
    }
}

// A work unit of the matrix-core launch: one tile and a run of contraction
// steps.  The host cuts every tile into runs of (nearly) equal cost -- steps x
// column blocks -- so that all workgroups of a launch take the same time.
struct MfmaUnit {
  uint32_t row_tile, col_tile, step_lo, step_hi;
  uint32_t blocks;   // 32-column blocks of the tile that hold a valid column (>= 1)
  uint32_t tile;     // index of the tile in the launch's tile list (its image, see below)
};

// TB = 128: 4 waves, two workgroups per CU.  TB = 256: 8 waves, one workgroup per CU --
// half the plane words fetched and three quarters of the operand builds and LDS reads
// per instruction (a wave's full-tile share is 2 x 4 blocks), for launches of many tiles.
template <int KR, int KC, int TB>
__global__ __launch_bounds__(2 * TB, (TB == 128 && KR + KC <= 5) ? 2 : 1) void bitgemm_mfma_kernel(
    const uint32_t* __restrict__ planes, uint32_t wp, const RowDesc* __restrict__ rows,
    uint32_t nrows, const ColDesc* __restrict__ cols, uint32_t ncols_total,
    const MfmaUnit* __restrict__ units, uint32_t* __restrict__ counts, int use_atomics) {
  // two slabs of A and of B (one array: see bitgemm_kernel), and behind them the
  // plane offsets of the tile's rows and columns
  __shared__ __attribute__((aligned(16))) uint32_t smem[2 * 2 * TB * LDS_STRIDE + (TB == 128 ? 0 : TB * (KR + KC))];
  const MfmaUnit u = units[blockIdx.x];
  const uint32_t row0 = u.row_tile * TB, col0 = u.col_tile * TB;
  constexpr uint32_t JFULL = TB / 32;
  const uint32_t jm = (use_atomics & 2) ? JFULL : u.blocks;
  const bool atomics = (use_atomics & 1) != 0 && !(use_atomics & 8);
  // image mode: `counts` is the launch's array of tile images
  uint32_t* image = (use_atomics & 1024) ? counts + (size_t)u.tile * (TB * TB) : nullptr;
#define OBNSL_MFMA_BODY(NA, NB)                                                                          \
  mfma_tile_body<KR, KC, NA, NB, TB>(smem, planes, wp, rows, nrows, cols, ncols_total, row0, col0, counts, \
                                     u.step_lo, u.step_hi, atomics, image, (use_atomics & 4) != 0,       \
                                     (use_atomics & 32) != 0, (use_atomics >> 6) & 7)
  switch (jm) {  // uniform over the workgroup
    case 1: OBNSL_MFMA_BODY(1, 1); break;
    case 2: OBNSL_MFMA_BODY(1, 2); break;
    case 3: OBNSL_MFMA_BODY(1, 3); break;
    default:
      if constexpr (TB == 128) {
        if (use_atomics & 16) OBNSL_MFMA_BODY(1, 4);   // OBNSL_NO_2X2 (comparison runs)
        else OBNSL_MFMA_BODY(2, 2);
      } else {
        if (jm == 4) OBNSL_MFMA_BODY(1, 4);
        else OBNSL_MFMA_BODY(2, 4);
      }
      break;
  }
#undef OBNSL_MFMA_BODY
}

// Moves the finished tile images of a split-contraction launch into the count table:
// one workgroup per (tile, 8 rows), a half-wave per row, every valid (row, column) cell
// exactly once (plain stores: the table needs no clearing).
template <int TB>
__global__ __launch_bounds__(256) void image_scatter_kernel(
    const uint32_t* __restrict__ images, const uint2* __restrict__ items,
    const RowDesc* __restrict__ rows, uint32_t nrows, uint32_t* __restrict__ counts) {
  const uint32_t tile = blockIdx.x / (TB / 8), r8 = blockIdx.x % (TB / 8);
  const uint2 it = items[tile];
  const uint32_t rin = r8 * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
  const uint32_t r = it.x * TB + rin;
  if (r >= nrows) return;
  const RowDesc rd = rows[r];
  const uint32_t* src = images + (size_t)tile * (TB * TB) + rin * TB;
#pragma unroll
  for (int cb = 0; cb < TB / 32; ++cb) {
    const uint32_t c = it.y * TB + cb * 32 + lane;
    if (c < rd.ncols) counts[rd.out_base + (uint64_t)c * rd.M] = src[cb * 32 + lane];
  }
}

// ------------------------------------------------------------------ single table
// Dense counts of one explicit subset: counts[cell] += popc(AND of the cell's
// planes) over this block's share of the words.  Generic and simple: it serves
// the one-table calls (ContingencyTable(var_ids, df)), not the batch path.
struct OneTable {   // a single explicit subset, passed by value (no upload per call)
  uint32_t pbase[KMAX];
  uint32_t cards[KMAX];
};
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

// ------------------------------------------------------------------ scoring
// lgamma(x) for x > 0: shift the argument up to >= 16 with the recurrence
// lgamma(x) = lgamma(x + m) - ln(x (x+1) ... (x+m-1)), then the Stirling series
// through the 1/x^11 term (the next term is < 2e-18 at x = 16).  Agrees with
// glibc's lgamma to ~2e-14 absolute over the arguments the scores use (counts
// plus a positive prior); far smaller and lighter on registers than the
// general-purpose library routine, which matters for the scorer's occupancy.
__device__ __forceinline__ double lgamma_pos(double x) {
  double shift = 0.0;
  if (x < 16.0) {
    double p = 1.0;
    while (x < 16.0) {
      p *= x;
      x += 1.0;
    }
    shift = log(p);
  }
  const double inv = 1.0 / x, inv2 = inv * inv;
  const double ser =
      inv * (1.0 / 12.0 -
             inv2 * (1.0 / 360.0 -
                     inv2 * (1.0 / 1260.0 -
                             inv2 * (1.0 / 1680.0 - inv2 * (1.0 / 1188.0 - inv2 * (691.0 / 360360.0))))));
  return (x - 0.5) * log(x) - x + 0.91893853320467274178 + ser - shift;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// All k local scores of one dense table, by one wavefront (lane = 0..63).
// BDeu as the reference writes it (include/score/local_score.h:38-61): only
// non-zero cells and non-zero parent configurations contribute; alpha_ijk =
// ess / prod(card), alpha_ij = alpha_ijk * r_child.  The sums are taken
// lane-strided and combined by a fixed butterfly, so results are reproducible
// run to run (the reference adds the same terms one by one in key order).
// BIC (not in the reference): sum N_ijk ln N_ijk - sum N_ij ln N_ij
//                             - 0.5 ln(N) q (r - 1).
__device__ void score_table(const uint32_t* __restrict__ cnt, uint32_t k, const uint32_t* cards,
                            int kind, double ess, uint32_t lane, double* out) {
  uint32_t cells = 1;
  for (uint32_t i = 0; i < k; ++i) cells *= cards[i];
  double a_ijk = 0.0, lg_a_ijk = 0.0;
  if (kind == OBNSL_SCORE_BDEU) {
    a_ijk = ess / (double)cells;
    lg_a_ijk = lgamma_pos(a_ijk);
  }
  double t1 = 0.0, tot = 0.0;
  for (uint32_t c = lane; c < cells; c += 64) {
    const uint32_t nijk = cnt[c];
    if (nijk == 0) continue;
    if (kind == OBNSL_SCORE_BDEU) {
      t1 += lgamma_pos((double)nijk + a_ijk) - lg_a_ijk;
    } else {
      t1 += (double)nijk * log((double)nijk);
      tot += (double)nijk;
    }
  }
  t1 = wave_sum(t1);
  if (kind != OBNSL_SCORE_BDEU) tot = wave_sum(tot);
  uint32_t w = 1;  // radix weight of variable i
  for (int i = (int)k - 1; i >= 0; --i) {
    const uint32_t r = cards[i];
    const uint32_t q = cells / r;
    double a_ij = 0.0, lg_a_ij = 0.0;
    if (kind == OBNSL_SCORE_BDEU) {
      a_ij = a_ijk * (double)r;
      lg_a_ij = lgamma_pos(a_ij);
    }
    double t2 = 0.0;
    for (uint32_t j = lane; j < q; j += 64) {
      const uint32_t hi = j / w, lo = j - hi * w;
      const uint32_t base = hi * (r * w) + lo;
      uint32_t nij = 0;
      for (uint32_t s = 0; s < r; ++s) nij += cnt[base + s * w];
      if (nij == 0) continue;
      if (kind == OBNSL_SCORE_BDEU) t2 += lgamma_pos((double)nij + a_ij) - lg_a_ij;
      else t2 += (double)nij * log((double)nij);
    }
    t2 = wave_sum(t2);
    double sc = t1 - t2;
    if (kind == OBNSL_SCORE_BIC) sc -= 0.5 * log(tot) * (double)q * (double)(r - 1);
    if (lane == 0) out[i] = sc;
    w *= r;
  }
}

// One wavefront per table of a batch: table t pairs row group g (found by
// binary search on tbl_start) with column group t - tbl_start[g].
__global__ __launch_bounds__(256) void score_plan_kernel(
    const uint32_t* __restrict__ counts, const RowGroupDesc* __restrict__ rgs, uint32_t nrgs,
    const ColGroupDesc* __restrict__ cgs, const uint32_t* __restrict__ card, uint32_t kr,
    uint32_t kc, uint64_t ntables, int kind, double ess, double* __restrict__ scores) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t t = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntables) return;
  uint32_t lo = 0, hi = nrgs - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (rgs[mid].tbl_start <= t) lo = mid;
    else hi = mid - 1;
  }
  const RowGroupDesc rg = rgs[lo];
  const uint32_t gamma = (uint32_t)(t - rg.tbl_start);
  uint32_t cards[KFAST];
  uint32_t col_before = 0;
  if (kc > 0) {
    const ColGroupDesc cg = cgs[gamma];
    col_before = cg.col_before;
    for (uint32_t j = 0; j < kc; ++j) cards[j] = card[cg.vars[j]];
  }
  for (uint32_t j = 0; j < kr; ++j) cards[kc + j] = card[rg.vars[j]];
  const uint32_t k = kr + kc;
  const uint32_t* cnt = counts + rg.out_off + (uint64_t)col_before * rg.M;
  const uint64_t rank = rg.rank_part + gamma;
  score_table(cnt, k, cards, kind, ess, lane, scores + rank * k);
}

__global__ __launch_bounds__(256) void score_tables_kernel(
    const uint32_t* __restrict__ counts, const TableDesc* __restrict__ tables, uint64_t ntables,
    int kind, double ess, double* __restrict__ scores) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t t = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntables) return;
  const TableDesc td = tables[t];
  uint32_t cards[KMAX];
  for (uint32_t j = 0; j < td.k; ++j) cards[j] = td.cards[j];
  score_table(counts + td.off, td.k, cards, kind, ess, lane, scores + t * td.k);
}

// One table whose descriptor comes by value (obnsl_local_score).
__global__ __launch_bounds__(64) void score_one_kernel(const uint32_t* __restrict__ counts, TableDesc td,
                                                       int kind, double ess, double* __restrict__ scores) {
  score_table(counts, td.k, td.cards, kind, ess, threadIdx.x & 63, scores);
}

// ---- reduced ("non-reference states only") tables -> full tables -> scores
//
// In reduced mode the bit-GEMM counts, for every subset T, only the joint states
// in which every variable of T is in one of its states 1..r-1:
//   R_T(y) = #{rows : x_v = y_v for all v in T},  y_v >= 1.
// That is prod(r_v - 1) cells instead of prod(r_v).  The full table of a subset S
// follows from the reduced tables of all T subset-of S: write "any" for digit 0,
//   E_S(e) = R_supp(e)(e restricted to its non-zero digits),   E_S(all any) = N,
// then turn "any" into "state 0" one variable at a time,
//   F(.., 0, ..) = E(.., any, ..) - sum_{s>=1} E(.., s, ..),
// which is exact integer arithmetic (inclusion-exclusion).
// x / r for x < 2^20 and 1 <= r <= 4096 without the ~35-instruction integer
// division: floor((x + 0.5) * (1/r)) is exact there (the 0.5 keeps exact
// multiples away from the rounding edge; float carries x + 0.5 exactly).
__device__ __forceinline__ uint32_t fast_div(uint32_t x, float inv_r) {
  return (uint32_t)(((float)x + 0.5f) * inv_r);
}

struct SizeView {            // where the reduced tables of one subset size live
  const uint32_t* red;       // counters
  const uint64_t* rowoff;    // first counter of a row group, by colex rank of its kr-subset
  const ColGroupDesc* cgs;   // column groups in colex order (col_before in reduced cells)
  uint32_t kr, kc;
};
struct ReducedViews {
  SizeView v[KFAST + 1];
};

// One wavefront (one 64-thread workgroup) per table: gather E from the reduced
// tables of the 2^k - 1 non-empty sub-subsets, k in-place passes, then the k scores.
//
// Scores come from a per-table functional: with
//   G(T) = sum over non-zero cells c of T of [lgamma_pos(N_c + ess/cells(T)) - lgamma_pos(ess/cells(T))]
// the reference's BDeu local score (include/score/local_score.h:38-61) is
//   score(v | P) = G(P + v) - G(P),
// because the parent-configuration counts N_ij of the table over P + v are the
// cells of the table over P, and alpha_ij = alpha_ijk * r_v = ess / cells(P).
// (BIC: the same with G(T) = sum N_c ln N_c, minus the penalty.)  So this kernel
// evaluates ONE sum per table, stores it in G[rank], and lane i < k writes
// score i = G(S) - Gprev[rank(S without v_i)]; Gprev belongs to the next smaller
// subset size (g0 for the empty set).  scores may be null (G only).
// (templated on the subset size so that the per-variable loops unroll exactly)
template <int KT, int PARTS>
__global__ __launch_bounds__(64) void score_reduced_kernel(
    ReducedViews views, const RowGroupDesc* __restrict__ rgs, uint32_t nrgs,
    uint64_t ntables, uint32_t tables_per_wg, const uint32_t* __restrict__ card,
    const uint64_t* __restrict__ binom, uint32_t num_rows, int kind, double ess,
    double* __restrict__ G, const double* __restrict__ Gprev, double g0,
    double* __restrict__ scores, const double* __restrict__ lg_alpha, uint32_t e_stride,
    const uint32_t* __restrict__ wg_rg) {
  // The wavefront works on PARTS tables at a time (2 or 4: 32 or 16 lanes each):
  // a table is a chain of dependent loads (descriptors -> table bases -> counts
  // -> G of the subsets below), and several chains in flight per wave hide more
  // of that latency than one; four pay when the tables are small (few cells per
  // lane either way).  All parts pass the same barriers; a part without a table
  // idles through them.
  constexpr uint32_t SC_PARTS = PARTS, SC_LANES = 64 / PARTS;
  extern __shared__ __attribute__((aligned(16))) uint32_t E_all[];
  __shared__ SizeView sv[KFAST + 1];
  __shared__ uint32_t s_vars[SC_PARTS][KFAST], s_cards[SC_PARTS][KFAST];
  __shared__ uint64_t pat_base_all[SC_PARTS][1 << KT];  // first counter of the reduced table of a support pattern
  constexpr uint32_t k = KT;
  const uint32_t half = threadIdx.x / SC_LANES, lane = threadIdx.x % SC_LANES;
  uint32_t* E = E_all + (size_t)half * e_stride;
  uint64_t* pat_base = pat_base_all[half];
#pragma unroll
  for (int j = 0; j <= KFAST; ++j)
    if (threadIdx.x == (uint32_t)j) sv[j] = views.v[j];
  const uint64_t t0 = (uint64_t)blockIdx.x * tables_per_wg;
  const uint64_t t1 = min(t0 + (uint64_t)tables_per_wg, ntables);
  // row group of the workgroup's first table (from the host: one load instead of a
  // binary search of dependent loads); every part advances from there through the groups
  uint32_t g = wg_rg[blockIdx.x];
  RowGroupDesc rg = rgs[g];
  const uint32_t kc = k / 2, kr = k - kc;
  __syncthreads();
  const ColGroupDesc* __restrict__ own_cgs = sv[k].cgs;

  for (uint64_t tp = t0; tp < t1; tp += SC_PARTS) {
    const uint64_t t = tp + half;
    const bool active = t < t1;
    uint32_t vars[KFAST], cards[KFAST];
    float inv_card[KFAST];
    uint32_t cells = 1, gamma = 0;
#pragma unroll
    for (int i = 0; i < KFAST; ++i) {
      vars[i] = 0;
      cards[i] = 1;
      inv_card[i] = 1.0f;
    }
    if (active) {
      while (t >= rg.tbl_start + rg.ncolgroups) rg = rgs[++g];
      gamma = (uint32_t)(t - rg.tbl_start);
      // (fixed-trip loops guarded by i < k keep these small arrays in registers)
      ColGroupDesc cg;
      if (kc > 0) cg = own_cgs[gamma];
#pragma unroll
      for (int i = 0; i < KFAST; ++i) {
        if (i < KSIDE && i < (int)kc) vars[i] = cg.vars[i];
#pragma unroll
        for (int j = 0; j < KSIDE; ++j)
          if (i == (int)kc + j && j < (int)kr) vars[i] = rg.vars[j];
      }
#pragma unroll
      for (int i = 0; i < KFAST; ++i) {
        if (i < (int)k) {
          cards[i] = card[vars[i]];
          inv_card[i] = 1.0f / (float)cards[i];
          cells *= cards[i];
        }
      }
      // where the reduced table of each non-empty support pattern starts: the table
      // of T (|T| = j) lives in size j's layout, row side = its kr_j largest variables
      for (uint32_t pat = lane + 1; pat < (1u << k); pat += SC_LANES) {
        const SizeView s = sv[__popc(pat)];
        uint64_t colrank = 0, rowrank = 0;
        uint32_t mred = 1, p = 0;
#pragma unroll
        for (int i = 0; i < KFAST; ++i) {
          if (i >= (int)k || !(pat >> i & 1)) continue;
          ++p;
          if (p <= s.kc) {
            colrank += binom[vars[i] * (KMAX + 1) + p];
          } else {
            rowrank += binom[vars[i] * (KMAX + 1) + (p - s.kc)];
            mred *= cards[i] - 1;
          }
        }
        const uint32_t col_before = s.kc ? s.cgs[colrank].col_before : 0;
        pat_base[pat] = s.rowoff[rowrank] + (uint64_t)col_before * mred;
      }
    }
    __syncthreads();
    // E(e): digit 0 = "any".  Within a pattern the counter index is the mixed-radix
    // number of the non-zero digits (radix r - 1, first variable most significant).
    // Four cells per lane at a time: their counter loads are issued together, before
    // any of them is stored (a store into LDS between them would order the loads).
    if (active) {
      for (uint32_t cell0 = lane; cell0 < cells; cell0 += 4 * SC_LANES) {
        uint32_t val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t cell = cell0 + u * SC_LANES;
          val[u] = num_rows;
          if (cell < cells) {
            uint32_t rem = cell, pat = 0, key = 0, mul = 1;
#pragma unroll
            for (int i = KFAST - 1; i >= 0; --i) {
              if (i >= (int)k) continue;
              const uint32_t qt = fast_div(rem, inv_card[i]);
              const uint32_t e = rem - qt * cards[i];
              rem = qt;
              if (e) {
                pat |= 1u << i;
                key += (e - 1) * mul;
                mul *= cards[i] - 1;
              }
            }
            if (pat) val[u] = sv[__popc(pat)].red[pat_base[pat] + key];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t cell = cell0 + u * SC_LANES;
          if (cell < cells) E[cell] = val[u];
        }
      }
    }
    __syncthreads();
    uint32_t w = 1;
#pragma unroll
    for (int i = KFAST - 1; i >= 0; --i) {
      if (i >= (int)k) continue;
      if (active) {
        const uint32_t r = cards[i];
        const uint32_t lines = fast_div(cells, inv_card[i]);
        const float inv_w = 1.0f / (float)w;
        for (uint32_t l = lane; l < lines; l += SC_LANES) {
          const uint32_t h = fast_div(l, inv_w), lo_ = l - h * w;
          const uint32_t base = h * (r * w) + lo_;
          uint32_t sum = 0;
          for (uint32_t st = 1; st < r; ++st) sum += E[base + st * w];
          E[base] -= sum;
        }
        w *= r;
      }
      __syncthreads();
    }
    const uint64_t rank = rg.rank_part + gamma;
    double gsum = 0.0;
    if (active) {
      if (kind == OBNSL_SCORE_BDEU) {
        const double a = ess / (double)cells;
        const double lg_a = lg_alpha[cells];  // lgamma(a): the same for every table of this size
        for (uint32_t c = lane; c < cells; c += SC_LANES) {
          const uint32_t nc = E[c];
          if (nc) gsum += lgamma_pos((double)nc + a) - lg_a;
        }
      } else {
        for (uint32_t c = lane; c < cells; c += SC_LANES) {
          const uint32_t nc = E[c];
          if (nc) gsum += (double)nc * log((double)nc);
        }
      }
    }
#pragma unroll
    for (int o = SC_LANES / 2; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o);  // within the part
    if (active && lane == 0) G[rank] = gsum;
    if (scores != nullptr) {
#pragma unroll
      for (int i = 0; i < KFAST; ++i)
        if (lane == (uint32_t)i && i < (int)k) {
          s_vars[half][i] = vars[i];
          s_cards[half][i] = cards[i];
        }
      __syncthreads();
      if (active && lane < k) {
        double gp = g0;
        if (k > 1) {
          uint64_t sub = 0;
          uint32_t pos = 1;
          for (uint32_t j = 0; j < k; ++j) {
            if (j == lane) continue;
            sub += binom[s_vars[half][j] * (KMAX + 1) + pos];
            ++pos;
          }
          gp = Gprev[sub];
        }
        double sc = gsum - gp;
        if (kind == OBNSL_SCORE_BIC) {
          const uint32_t r = s_cards[half][lane];
          sc -= 0.5 * log((double)num_rows) * (double)(cells / r) * (double)(r - 1);
        }
        scores[rank * k + lane] = sc;
      }
    }
    __syncthreads();  // E, pat_base and s_vars are reused by the next pair of tables
  }
}

// ------------------------------------------------------------------ histogram path
// Tables too large for the bit-plane form (its cost grows with the number of
// cells): the reference's own loop (include/base/contingency_table.h:52-55), one
// workgroup per (table, row chunk).  Each lane reads 16 consecutive rows of every
// column of the subset as one 16-byte load (coalesced, column-major), forms the
// mixed-radix keys in registers and counts them in an LDS histogram; the chunk's
// histogram is then added to the table's dense counters.  HBM/L2-bound: k bytes
// per row per table.
__device__ __forceinline__ uint32_t unrank_step(const uint64_t* __restrict__ binom, uint32_t n,
                                                uint32_t j, uint64_t rank);
constexpr uint32_t HIST_LDS_CELLS = 16384;   // 64 KiB of LDS counters
constexpr uint32_t HIST_ROWS_PER_ITER = 256 * 16;
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
constexpr uint32_t BEST_LDS_VARS = 2048;
struct BestArgs {                 // every subset size in one launch
  const double* scores[KMAX];     // scores[k-1]: table of subset size k
  uint64_t nsubsets[KMAX];
  uint32_t block_start[KMAX + 1]; // workgroups [block_start[k-1], block_start[k]) serve size k
  uint32_t kmax;
};
template <int KCAP>
__global__ __launch_bounds__(256) void best_parents_kernel(
    BestArgs args, uint32_t n, const uint64_t* __restrict__ binom,
    unsigned long long* __restrict__ best_key, unsigned long long* __restrict__ best_id, int pass) {
  uint32_t k = 1;
#pragma unroll
  for (int j = 1; j < KMAX; ++j)
    if ((uint32_t)j < args.kmax && blockIdx.x >= args.block_start[j]) k = j + 1;
  const double* __restrict__ scores = nullptr;
  uint64_t nsubsets = 0;
  uint32_t first_block = 0, nblocks = 1;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if ((uint32_t)j + 1 == k) {
      scores = args.scores[j];
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
    uint64_t rem = t;
    for (uint32_t j = k; j >= 1; --j) {
      const uint32_t v = unrank_step(binom, n, j, rem);
      vars[j - 1] = v;
      rem -= binom[v * (KMAX + 1) + j];
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
struct DpScores {
  const double* tbl[KMAX];  // tbl[k-1][rank * k + i]
  uint32_t max_scored;      // max_parents + 1
};
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

// ------------------------------------------------------------------ host side
uint64_t binom_host(uint32_t n, uint32_t k) {
  if (k > n) return 0;
  if (k > n - k) k = n - k;
  unsigned __int128 r = 1;
  for (uint32_t i = 1; i <= k; ++i) {
    r = r * (n - k + i) / i;
    if (r > (unsigned __int128)UINT64_MAX) return UINT64_MAX;
  }
  return (uint64_t)r;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { reset(); }
  void reset() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t ensure(size_t count) {
    if (count <= n) return hipSuccess;
    reset();
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h) {
    hipError_t e = ensure(std::max<size_t>(h.size(), 1));
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  }
};

struct Batch {
  DevBuf<RowDesc> rows;
  DevBuf<uint2> items;
  std::vector<uint2> h_items;          // host copy of `items` (tile coordinates), largest first
  std::vector<uint32_t> h_item_blocks; // 32-column blocks of each item that hold a valid column
  DevBuf<MfmaUnit> units;              // matrix-core launch: work units for `units_steps` steps
  uint32_t nunits = 0, units_steps = 0, units_target = 0;
  bool units_split = false;            // some tile is cut into several units (atomics needed)
  std::vector<uint64_t> h_tbl_start;   // first table of every row group (host copy)
  DevBuf<uint32_t> wg_rg;              // scorer: row group of each workgroup's first table ...
  uint32_t wg_rg_tpw = 0;              // ... for this many tables per workgroup
  DevBuf<uint32_t> img;                // split launches: one row-major tile x tile image per item
  uint64_t img_zeroed_epoch = 0;       // data epoch at whose pack `img` was cleared
  bool img_wants_zero = false;         // the last launch added into `img`
  DevBuf<RowGroupDesc> rgs;
  uint32_t nrows = 0, nitems = 0, nrgs = 0;
  uint64_t ntables = 0, ncells = 0;
};

struct HistBatch {    // histogram path: a run of consecutive ranks whose dense tables fit one buffer
  uint64_t rank0 = 0, cells_total = 0;
  uint32_t ntables = 0, max_cells = 1;
  DevBuf<uint64_t> off;     // first counter of each table
  DevBuf<TableDesc> tds;
};

struct Plan {
  uint32_t k = 0, kr = 0, kc = 0;
  bool reduced = false;         // rows/columns cover states 1..r-1 only (see score_reduced_kernel)
  uint32_t tile = 128;          // C tile edge of the bit-GEMM launches of this plan
  bool mfma = false;            // counted by bitgemm_mfma_kernel (128 x 128 tiles, 32-column blocks)
  DevBuf<ColDesc> cols;
  DevBuf<ColGroupDesc> cgs;
  uint32_t ncols_total = 0;
  std::vector<std::unique_ptr<Batch>> batches;  // exactly one in reduced mode
  // reduced mode: the counters persist per plan, so that larger subsets can read them
  DevBuf<uint32_t> red;
  DevBuf<uint64_t> rowoff;      // by colex rank of the kr-subset
  std::vector<uint64_t> h_rowoff;       // host copies (obnsl_reduced_table)
  std::vector<uint32_t> h_col_before;   // by colex rank of the kc-subset
  uint64_t counted_epoch = 0;   // data epoch the counters in `red` belong to
  int pending_lane = -1;        // side lane still producing `red` (-1: none, or already joined)
  uint64_t zeroed_epoch = 0;    // data epoch at whose pack `red` was cleared
  bool wants_zero = false;      // the last count into `red` added with atomics (or filled a slice of it)
  uint64_t slice_off = 0;       // sharded plan: where its counters sit inside the whole plan's `red`
  uint64_t size_cells = 0;      // counters of ALL tables of this subset size (the same for every shard)
  bool awaits_sum = false;      // whole plan: `red` holds one rank's slice, the sum over ranks is pending
  uint64_t total_items = 0;
  // per-table functional G[colex rank] of this plan's tables (see score_reduced_kernel)
  DevBuf<double> G;
  uint64_t g_epoch = 0;
  int g_kind = -1;
  double g_ess = 0.0;
};

}  // namespace

struct obnsl_data {
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t N = 0, stride = 0;
  uint32_t n = 0, wp = 0, total_planes = 0, ones_plane = 0;
  std::vector<uint32_t> card, plane_base;
  DevBuf<uint8_t> codes;
  DevBuf<uint32_t> planes, d_card, d_plane_base, counts;
  DevBuf<unsigned long long> best_key;           // per-node arg-max scratch of obnsl_best_parents
  // scratch of the one-table / explicit-list entry points: grows, never shrinks
  DevBuf<uint32_t> one_cnt, list_pb, list_cnt;
  DevBuf<TableDesc> list_td;
  DevBuf<double> list_out;
  DevBuf<uint64_t> d_binom;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, bool>, std::unique_ptr<Plan>> plans;
  std::map<std::tuple<uint32_t, uint64_t, uint64_t>, std::vector<std::unique_ptr<HistBatch>>> hist_plans;
  uint64_t epoch = 1;                                 // bumped whenever the planes are rebuilt
  // obnsl_precount counts on two extra streams ("lanes", default priority): lane 0
  // the largest size, lane 1 the small sizes below the second largest
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t side_fork = nullptr, side_join[2] = {nullptr, nullptr};
  int count_lane = -1;                                // launch_count targets side[count_lane] (-1: `stream`)
  bool beside_large = false;                          // the launches of obnsl_precount run beside one of many tiles
  std::vector<Plan*> lane_plans[2];                   // plans whose counts a lane is still producing
  // per ess: lgamma(ess / c) for c = 1..REDUCED_MAX_CELLS (index c), from the host's libm
  std::map<double, std::unique_ptr<DevBuf<double>>> lg_alpha;
  hipStream_t count_stream() const {
    // OBNSL_SERIAL: every launch on the main stream (profiling: kernels do not overlap)
    static const bool serial = std::getenv("OBNSL_SERIAL") != nullptr;
    return count_lane >= 0 && !serial ? side[count_lane] : stream;
  }
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // around each bit-GEMM launch
  std::vector<uint32_t> ev_size;                      // subset size of each pending pair
  uint32_t ev_used = 0;                               // pending pairs
  double acc_ms[OBNSL_MAX_SUBSET + 1] = {0};          // per subset size, since the last reset
  uint32_t acc_launches[OBNSL_MAX_SUBSET + 1] = {0};
  ~obnsl_data() {
    for (auto& e : ev) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    if (side_fork) (void)hipEventDestroy(side_fork);
    for (int l = 0; l < 2; ++l) {
      if (side_join[l]) (void)hipEventDestroy(side_join[l]);
      if (side[l]) (void)hipStreamDestroy(side[l]);
    }
  }
};

namespace {

// Enumerate size-k subsets of {0..n-1} in colex order.
void enumerate_colex(uint32_t n, uint32_t k, std::vector<std::vector<uint32_t>>& out) {
  out.clear();
  if (k > n) return;
  std::vector<uint32_t> c(k);
  for (uint32_t i = 0; i < k; ++i) c[i] = i;
  if (k == 0) {
    out.push_back(c);
    return;
  }
  for (;;) {
    out.push_back(c);
    uint32_t i = 0;
    for (; i < k; ++i) {
      const uint32_t lim = (i + 1 < k) ? c[i + 1] : n;
      if (c[i] + 1 < lim) break;
    }
    if (i == k) break;
    ++c[i];
    for (uint32_t j = 0; j < i; ++j) c[j] = j;
  }
}

obnsl_status build_plan(obnsl_data* d, uint32_t k, uint32_t shard, uint32_t nshards,
                        bool reduced, Plan** out) {
  auto key = std::make_tuple(k, shard, nshards, reduced);
  auto itp = d->plans.find(key);
  if (itp != d->plans.end()) {
    *out = itp->second.get();
    return OBNSL_OK;
  }
  auto plan = std::make_unique<Plan>();
  plan->k = k;
  plan->kc = k / 2;
  plan->kr = k - plan->kc;
  plan->reduced = reduced;
  const uint32_t kr = plan->kr, kc = plan->kc, n = d->n;
  // states a row/column enumerates per variable, and the plane of the first one
  auto radix = [&](uint32_t v) { return reduced ? d->card[v] - 1 : d->card[v]; };
  const uint32_t first_state = reduced ? 1 : 0;

  // column side: every kc-subset in colex order, cells with the first variable
  // most significant
  std::vector<std::vector<uint32_t>> csub;
  enumerate_colex(n, kc, csub);
  std::vector<ColGroupDesc> cgs(csub.size());
  std::vector<ColDesc> cols;
  {
    uint64_t total = 0;
    for (size_t g = 0; g < csub.size(); ++g) {
      uint64_t cells = 1;
      for (uint32_t v : csub[g]) cells *= radix(v);
      total += cells;
      if (total > (1u << 27))
        return fail(OBNSL_ERR_UNSUPPORTED, "column side of the table batch is too large");
    }
    cols.reserve(total);
    uint32_t before = 0;
    for (size_t g = 0; g < csub.size(); ++g) {
      ColGroupDesc& cg = cgs[g];
      std::memset(&cg, 0, sizeof(cg));
      uint32_t cells = 1;
      for (uint32_t j = 0; j < kc; ++j) {
        cg.vars[j] = csub[g][j];
        cells *= radix(csub[g][j]);
      }
      cg.col_before = before;
      cg.cells = cells;
      for (uint32_t cell = 0; cell < cells; ++cell) {
        ColDesc cd;
        for (int j = 0; j < KSIDE; ++j) cd.plane[j] = d->ones_plane;
        uint32_t rem = cell;
        for (int j = (int)kc - 1; j >= 0; --j) {
          const uint32_t r = radix(csub[g][j]);
          cd.plane[j] = d->plane_base[csub[g][j]] + first_state + rem % r;
          rem /= r;
        }
        cols.push_back(cd);
      }
      before += cells;
    }
  }
  if (cols.empty()) {  // keeps the kernel's clamped column index valid; never a valid column
    ColDesc cd;
    for (int j = 0; j < KSIDE; ++j) cd.plane[j] = d->ones_plane;
    cols.push_back(cd);
  }
  plan->ncols_total = (uint32_t)cols.size();
  // columns usable by a row group whose smallest variable is u: those of the
  // first C(u, kc) column groups
  std::vector<uint32_t> ncols_below(n + 1, 0);
  for (uint32_t u = 0; u <= n; ++u) {
    const uint64_t g = kc == 0 ? 1 : binom_host(u, kc);
    if (g == 0) ncols_below[u] = 0;
    else if (g >= cgs.size()) ncols_below[u] = cgs.back().col_before + cgs.back().cells;
    else ncols_below[u] = cgs[g].col_before;
  }

  // row side: kr-subsets with at least kc variables below them, by smallest
  // variable (then colex), so that the valid columns of a row tile are a prefix
  std::vector<std::vector<uint32_t>> rsub_all, rsub;
  enumerate_colex(n, kr, rsub_all);
  for (auto& s : rsub_all)
    if (s[0] >= kc) rsub.push_back(s);
  std::stable_sort(rsub.begin(), rsub.end(),
                   [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
                     return a[0] < b[0];
                   });
  std::vector<uint64_t> cum(rsub.size() + 1, 0);
  std::vector<uint64_t> Ms(rsub.size());
  for (size_t g = 0; g < rsub.size(); ++g) {
    uint64_t M = 1;
    for (uint32_t v : rsub[g]) M *= radix(v);
    if (M > (1u << 24)) return fail(OBNSL_ERR_UNSUPPORTED, "row side of a table is too large");
    Ms[g] = M;
    cum[g + 1] = cum[g] + M * ncols_below[rsub[g][0]];
  }
  const uint64_t total_cost = cum[rsub.size()];
  // shard = contiguous run of row groups.  What a shard costs is what its
  // launch computes: row tiles of `edge` rows (the last one padded) times the
  // column limit of each tile's last row rounded up to the 16-column groups the
  // kernel skips by.  The cuts minimise the most expensive shard (a shard of a
  // few very wide rows pays a whole row tile for them, so equal counter shares
  // are not equal work).
  size_t g_lo = 0, g_hi = rsub.size();
  if (nshards > 1 && !rsub.empty()) {
    std::vector<uint64_t> row0(rsub.size() + 1, 0);
    for (size_t g = 0; g < rsub.size(); ++g) row0[g + 1] = row0[g] + Ms[g];
    static const uint64_t edge = [] {
      const char* e = std::getenv("OBNSL_TILE");
      return use_mfma_gemm() ? (uint64_t)MFMA_TILE : e && std::atoi(e) == 128 ? 128u : 64u;
    }();
    const uint64_t cgroup = use_mfma_gemm() ? 32 : 16;  // columns the kernel skips by
    auto range_cost = [&](size_t lo, size_t hi) {
      uint64_t c = 0;
      size_t g = lo;
      for (uint64_t r = row0[lo]; r < row0[hi]; r += edge) {
        const uint64_t last = std::min<uint64_t>(r + edge, row0[hi]) - 1;
        while (row0[g + 1] <= last) ++g;
        const uint64_t lim = ncols_below[rsub[g][0]];
        if (use_mfma_gemm()) {
          // per column tile 2 x (column blocks) + 1, as build_units weighs them
          const uint64_t rem = lim % edge;
          c += (lim / edge) * 9 + (rem ? 2 * ((rem + cgroup - 1) / cgroup) + 1 : 0);
        } else {
          c += edge * ((lim + cgroup - 1) / cgroup * cgroup);
        }
      }
      return c;
    };
    // cuts[s] = first group of shard s under bottleneck B; false if B is too small
    auto cut = [&](uint64_t B, std::vector<size_t>& cuts) {
      cuts.assign(nshards + 1, rsub.size());
      size_t lo = 0;
      for (uint32_t sh = 0; sh < nshards; ++sh) {
        cuts[sh] = lo;
        if (sh + 1 == nshards) return range_cost(lo, rsub.size()) <= B;
        size_t a = lo, b = rsub.size();  // largest hi with cost(lo, hi) <= B
        while (a < b) {
          const size_t mid = (a + b + 1) / 2;
          if (range_cost(lo, mid) <= B) a = mid; else b = mid - 1;
        }
        lo = a;
      }
      return true;
    };
    std::vector<size_t> cuts, best_cuts;
    uint64_t lo_b = 0, hi_b = range_cost(0, rsub.size());
    if (row0.back() / edge > (1u << 16)) {
      // very many row tiles: equal shares of the counters are close enough
      best_cuts.assign(nshards + 1, rsub.size());
      for (uint32_t sh = 0; sh < nshards; ++sh) {
        const uint64_t c = total_cost / nshards * sh + std::min<uint64_t>(sh, total_cost % nshards);
        best_cuts[sh] = sh == 0 ? 0 : std::lower_bound(cum.begin(), cum.end() - 1, c) - cum.begin();
      }
      lo_b = hi_b;
    } else {
      cut(hi_b, best_cuts);
    }
    while (lo_b < hi_b) {
      const uint64_t mid = lo_b + (hi_b - lo_b) / 2;
      if (cut(mid, cuts)) {
        hi_b = mid;
        best_cuts = cuts;
      } else {
        lo_b = mid + 1;
      }
    }
    g_lo = best_cuts[shard];
    g_hi = best_cuts[shard + 1];
  }

  plan->slice_off = rsub.empty() ? 0 : cum[std::min(g_lo, rsub.size())];
  plan->size_cells = total_cost;
  std::vector<uint64_t> rowoff;
  if (reduced) rowoff.assign(binom_host(n, kr), 0);
  const uint64_t cell_cap = reduced ? (1ull << 33) : BATCH_CELL_CAP;
  size_t g = g_lo;
  while (g < g_hi || (reduced && plan->batches.empty())) {
    auto b = std::make_unique<Batch>();
    std::vector<RowDesc> rows;
    std::vector<RowGroupDesc> rgs;
    uint64_t off = 0, tables = 0;
    while (g < g_hi) {
      const uint64_t M = Ms[g];
      const uint32_t u1 = rsub[g][0];
      const uint64_t sz = M * ncols_below[u1];
      if (off > 0 && off + sz > cell_cap) {
        if (reduced) return fail(OBNSL_ERR_UNSUPPORTED, "reduced count tables exceed 2^33 cells");
        break;
      }
      if (rows.size() + M > (1u << 27))
        return fail(OBNSL_ERR_UNSUPPORTED, "row side of the table batch is too large");
      RowGroupDesc rg;
      std::memset(&rg, 0, sizeof(rg));
      uint64_t rank_part = 0, own_rank = 0;
      for (uint32_t j = 0; j < kr; ++j) {
        rg.vars[j] = rsub[g][j];
        rank_part += binom_host(rsub[g][j], kc + j + 1);
        own_rank += binom_host(rsub[g][j], j + 1);
      }
      if (reduced) rowoff[own_rank] = off;
      rg.out_off = off;
      rg.rank_part = rank_part;
      rg.tbl_start = tables;
      rg.M = (uint32_t)M;
      rg.ncolgroups = kc == 0 ? 1 : (uint32_t)binom_host(u1, kc);
      rgs.push_back(rg);
      for (uint32_t cell = 0; cell < M; ++cell) {
        RowDesc rd;
        for (int j = 0; j < KSIDE; ++j) rd.plane[j] = d->ones_plane;
        uint32_t rem = cell;
        for (int j = (int)kr - 1; j >= 0; --j) {
          const uint32_t r = radix(rsub[g][j]);
          rd.plane[j] = d->plane_base[rsub[g][j]] + first_state + rem % r;
          rem /= r;
        }
        rd.out_base = off + cell;
        rd.M = (uint32_t)M;
        rd.ncols = ncols_below[u1];
        rows.push_back(rd);
      }
      off += sz;
      tables += rg.ncolgroups;
      ++g;
    }
    // tiles: (row tile i, column tile j) for j below the tile's column limit.
    // 64 x 64 tiles (16 counters per thread, 122 registers, 4 waves per SIMD).
    // The 128 x 128 instance (64 counters per thread) needs more than 256
    // registers: one wave per SIMD issues a vector instruction every 4 cycles
    // instead of every 2, and capped to 256 it spills; measured 25 % and 13 %
    // slower than 64 x 64 on 128 four-state variables, so it is kept only behind
    // OBNSL_TILE=128 for experiments.
    auto make_tiles = [&](uint32_t tb, std::vector<std::pair<uint32_t, uint2>>& out) {
      out.clear();
      const uint32_t ntr = (uint32_t)((rows.size() + tb - 1) / tb);
      for (uint32_t i = 0; i < ntr; ++i) {
        const size_t last = std::min<size_t>((size_t)(i + 1) * tb, rows.size()) - 1;
        const uint32_t lim = rows[last].ncols;
        for (uint32_t j = 0; j * tb < lim; ++j)
          out.push_back({std::min<uint32_t>(lim - j * tb, tb), make_uint2(i, j)});
      }
    };
    std::vector<std::pair<uint32_t, uint2>> tiles;
    static const int force_tile = [] {
      const char* e = std::getenv("OBNSL_TILE");
      return e ? std::atoi(e) : 0;
    }();
    plan->mfma = use_mfma_gemm();
    plan->tile = plan->mfma ? MFMA_TILE : force_tile == 128 ? 128 : 64;
    make_tiles(plan->tile, tiles);
    if (plan->mfma && (kr == 2 || (kr == 3 && kc == 2))) {
      // launches of many tiles take 256 x 256 tiles
      static const int big = [] {
        const char* e = std::getenv("OBNSL_MFMA_TILE");
        return e ? std::atoi(e) : 0;
      }();
      static const size_t big_from = [] {
        const char* e = std::getenv("OBNSL_BIG_TILE_FROM");
        return e ? (size_t)std::atol(e) : BIG_TILE_FROM;
      }();
      if (big == 256 || (big == 0 && tiles.size() >= big_from)) {
        plan->tile = 256;
        make_tiles(256, tiles);
      }
    }
    std::stable_sort(tiles.begin(), tiles.end(),
                     [](const std::pair<uint32_t, uint2>& a, const std::pair<uint32_t, uint2>& b) {
                       return a.first > b.first;
                     });
    std::vector<uint2> items(tiles.size());
    for (size_t i = 0; i < tiles.size(); ++i) items[i] = tiles[i].second;
    b->h_items = items;
    b->h_item_blocks.resize(tiles.size());
    for (size_t i = 0; i < tiles.size(); ++i) b->h_item_blocks[i] = (tiles[i].first + 31) / 32;
    b->nrows = (uint32_t)rows.size();
    b->nitems = (uint32_t)items.size();
    b->nrgs = (uint32_t)rgs.size();
    b->ntables = tables;
    b->ncells = off;
    HIP_TRY(b->rows.upload(rows));
    HIP_TRY(b->items.upload(items));
    HIP_TRY(b->rgs.upload(rgs));
    b->h_tbl_start.resize(rgs.size());
    for (size_t i = 0; i < rgs.size(); ++i) b->h_tbl_start[i] = rgs[i].tbl_start;
    plan->total_items += b->nitems;
    plan->batches.push_back(std::move(b));
  }
  if (reduced) {
    plan->h_rowoff = rowoff;
    plan->h_col_before.resize(cgs.size());
    for (size_t g2 = 0; g2 < cgs.size(); ++g2) plan->h_col_before[g2] = cgs[g2].col_before;
    HIP_TRY(plan->rowoff.upload(rowoff));
    // `red` itself is allocated where it is first counted into (ensure_counts, the
    // sliced count of obnsl_precount_sharded): a sharded plan whose counts land in
    // the whole plan's table never owns one
  }
  HIP_TRY(plan->cols.upload(cols));
  HIP_TRY(plan->cgs.upload(cgs));
  *out = plan.get();
  d->plans[key] = std::move(plan);
  return OBNSL_OK;
}

// Folds the event pairs recorded since the last call into the running
// per-subset-size totals (waits for them if they are still in flight).  Only
// obnsl_kernel_time calls this, so scoring calls never block on the device.
obnsl_status resolve_events(obnsl_data* d) {
  for (uint32_t i = 0; i < d->ev_used; ++i) {
    float t = 0;
    HIP_TRY(hipEventSynchronize(d->ev[i].second));
    HIP_TRY(hipEventElapsedTime(&t, d->ev[i].first, d->ev[i].second));
    d->acc_ms[d->ev_size[i]] += t;
    d->acc_launches[d->ev_size[i]] += 1;
  }
  d->ev_used = 0;
  return OBNSL_OK;
}

template <int KR, int KC>
void launch_bitgemm(obnsl_data* d, const Plan* p, const Batch* b, uint32_t* counts, uint32_t grid,
                    uint32_t steps_total, uint32_t steps_per_split, int atomics) {
  if (p->mfma && p->tile == 256) {
    if constexpr (KR == 2 || (KR == 3 && KC == 2))
      hipLaunchKernelGGL((bitgemm_mfma_kernel<KR, KC, 256>), dim3(b->nunits), dim3(512), 0, d->count_stream(),
                         d->planes.p, d->wp, b->rows.p, b->nrows, p->cols.p, p->ncols_total,
                         b->units.p, counts, atomics);
  } else if (p->mfma)
    hipLaunchKernelGGL((bitgemm_mfma_kernel<KR, KC, 128>), dim3(b->nunits), dim3(256), 0, d->count_stream(),
                       d->planes.p, d->wp, b->rows.p, b->nrows, p->cols.p, p->ncols_total,
                       b->units.p, counts, atomics);
  else if (p->tile == 128)
    hipLaunchKernelGGL((bitgemm_kernel<KR, KC, 8>), dim3(grid), dim3(256), 0, d->count_stream(),
                       d->planes.p, d->wp, b->rows.p, b->nrows, p->cols.p, p->ncols_total,
                       b->items.p, b->nitems, counts, steps_total, steps_per_split, atomics);
  else
    hipLaunchKernelGGL((bitgemm_kernel<KR, KC, 4>), dim3(grid), dim3(256), 0, d->count_stream(),
                       d->planes.p, d->wp, b->rows.p, b->nrows, p->cols.p, p->ncols_total,
                       b->items.p, b->nitems, counts, steps_total, steps_per_split, atomics);
}

// Work units of the matrix-core launch of one batch: every tile is cut along the
// contraction into runs whose cost (steps x column blocks) is close to
// total cost / target, so a launch is `target` workgroups of equal length
// (built once per batch and row count).
obnsl_status build_units(Batch* b, uint32_t steps_total, uint32_t target) {
  if (b->units_steps == steps_total && b->units_target == target && b->nunits) return OBNSL_OK;
  uint64_t total = 0;
  // a tile with fewer column blocks amortises its A operand over fewer instructions:
  // measured 31 / 24 / 20 ns per instruction for 1 / 2 / 4 blocks, i.e. cost ~ 2 blocks + 1
  auto weight = [](uint32_t blk) { return 2 * blk + 1; };
  for (uint32_t blk : b->h_item_blocks) total += (uint64_t)weight(blk) * steps_total;
  const double per_unit = std::max(1.0, (double)total / (double)std::max(1u, target));
  std::vector<MfmaUnit> units;
  std::vector<std::pair<uint32_t, MfmaUnit>> tmp;  // (cost, unit)
  bool split = false;
  for (size_t i = 0; i < b->h_items.size(); ++i) {
    const uint32_t blk = b->h_item_blocks[i];
    uint32_t ns = (uint32_t)std::max(1.0, std::floor((double)weight(blk) * steps_total / per_unit + 0.5));
    ns = std::min(ns, std::max(1u, steps_total / 4));            // at least 4 steps per unit
    ns = std::max(ns, (steps_total + MFMA_MAX_STEPS - 1) / MFMA_MAX_STEPS);  // f32 counts stay exact
    ns = std::min(ns, steps_total);
    if (ns > 1) split = true;
    for (uint32_t sidx = 0; sidx < ns; ++sidx) {
      MfmaUnit u;
      u.row_tile = b->h_items[i].x;
      u.col_tile = b->h_items[i].y;
      u.step_lo = (uint32_t)((uint64_t)steps_total * sidx / ns);
      u.step_hi = (uint32_t)((uint64_t)steps_total * (sidx + 1) / ns);
      u.blocks = blk;
      u.tile = (uint32_t)i;
      tmp.push_back({weight(blk) * (u.step_hi - u.step_lo), u});
    }
  }
  std::stable_sort(tmp.begin(), tmp.end(),
                   [](const std::pair<uint32_t, MfmaUnit>& x, const std::pair<uint32_t, MfmaUnit>& y) {
                     return x.first > y.first;
                   });
  units.reserve(tmp.size());
  for (auto& t : tmp) units.push_back(t.second);
  static const bool dump = std::getenv("OBNSL_DUMP_UNITS") != nullptr;   // tools/mfma_rate.py
  if (dump) {
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> hist;
    for (auto& t : tmp) hist[{t.second.blocks, t.second.step_hi - t.second.step_lo}]++;
    std::fprintf(stderr, "units: %zu tiles, %zu units, steps %u, target %u\n", b->h_items.size(),
                 tmp.size(), steps_total, target);
    for (auto& h : hist)
      std::fprintf(stderr, "  blocks %u steps %u: %u units\n", h.first.first, h.first.second, h.second);
  }
  HIP_TRY(b->units.upload(units));
  b->nunits = (uint32_t)units.size();
  b->units_steps = steps_total;
  b->units_target = target;
  b->units_split = split;
  return OBNSL_OK;
}

// A top size of at most this many tiles is counted whole on every rank rather
// than in shards (one 128 x 128 matrix-core tile is sixteen 64 x 64 ones).
uint64_t whole_below(const Plan* p) { return p->mfma ? 4 : 64; }

// Units per matrix-core launch by number of tiles: one or two tiles, up to 32, more
// without and with tile images (OBNSL_WANT="a,b,c[,d]" overrides: tuning runs).
uint32_t want_tier(int tier) {
  static const std::vector<uint32_t> t = [] {
    std::vector<uint32_t> v = {64u, 384u, 896u, 4096u};
    if (const char* e = std::getenv("OBNSL_WANT")) {
      unsigned a = 0, b = 0, c = 0, f = 0;
      const int got = std::sscanf(e, "%u,%u,%u,%u", &a, &b, &c, &f);
      if (got >= 3 && a && b && c) v = {a, b, c, got == 4 && f ? f : c};
    }
    return v;
  }();
  return t[tier];
}

// One bit-GEMM launch (timed with an event pair) filling `counts` for one batch.
obnsl_status launch_count(obnsl_data* d, Plan* plan, Batch* b, uint32_t* counts,
                          bool prezeroed = false) {
  if (b->nitems == 0) return OBNSL_OK;
  const uint32_t steps_total = d->wp / KCW;
  // split the contraction when there are too few tiles to fill the chip
  // (workgroups of equal length run in rounds of 2 per CU; many short ones keep
  // the last round from idling most of the chip)
  uint32_t ksplit = 1;
  static const uint32_t want_env = [] {
    const char* e = std::getenv("OBNSL_TARGET_WGS");
    return e ? (uint32_t)std::atoi(e) : 0u;
  }();
  // the matrix-core kernel runs two 128 x 128 workgroups per CU: a few rounds of
  // 512 are enough, and every extra split adds the whole table in atomics
  // (measured on the 37-variable workload, whole pass: one tile cut into ~64 units -- 16
  // units of ~64 steps when it runs beside a launch of many tiles, whose workgroup slots
  // it would take --, 17 tiles into ~384, more tiles into ~4096 units, at least two per
  // tile and up to 12288 of ~96 steps (70 variables, 10^7 rows: 306 tiles, +5 % from 4096
  // to 8192..16384 units; 37 variables, five-variable tables: 759 tiles, +4 % from 4096 to
  // 8192) when the units add into tile images, ~896 when they add into the
  // table with scattered atomics: short units pay their prologue and epilogue, long ones
  // leave CUs idle at the end of the launch)
  // split launches of 128 x 128 tiles add into tile images (up to IMAGE_MAX_BYTES of
  // them) instead of the count table itself -- not the launches of one or two tiles:
  // their few units' atomics cost less than the scatter launch; OBNSL_NO_IMAGE:
  // comparison runs
  static const bool no_image = std::getenv("OBNSL_NO_IMAGE") != nullptr;
  static const uint64_t image_max = [] {
    const char* e = std::getenv("OBNSL_IMAGE_MB");
    return e ? (uint64_t)std::atoi(e) << 20 : IMAGE_MAX_BYTES;
  }();
  const uint64_t img_cells = (uint64_t)b->nitems * plan->tile * plan->tile;
  const bool image_ok = plan->mfma && !no_image && b->nitems > 2 &&
                        img_cells * sizeof(uint32_t) <= image_max;
  const uint32_t want = want_env ? want_env
                        : !plan->mfma ? 8192u
                        : b->nitems <= 2 ? (d->beside_large ? std::min(want_tier(0), std::max(16u, steps_total / 64))
                                                            : want_tier(0))
                        : b->nitems <= 32 ? want_tier(1)
                        : image_ok ? std::max({want_tier(3), 2 * b->nitems,
                                               (uint32_t)std::min<uint64_t>((uint64_t)b->nitems * steps_total / 96, 12288)})
                                   : want_tier(2);
  static const uint32_t min_sps = [] {
    const char* e = std::getenv("OBNSL_MIN_SPS");
    return e ? std::max(1u, (uint32_t)std::atoi(e)) : 8u;
  }();
  const uint32_t max_split = std::max<uint32_t>(1, steps_total / min_sps);
  if (b->nitems < want) ksplit = std::min<uint32_t>((want + b->nitems - 1) / b->nitems, max_split);
  uint32_t sps = (steps_total + ksplit - 1) / ksplit;
  if (plan->mfma) sps = std::min(sps, MFMA_MAX_STEPS);  // f32 accumulators stay exact below 2^24
  ksplit = (steps_total + sps - 1) / sps;
  static const int nojskip = (std::getenv("OBNSL_NO_JSKIP") ? 2 : 0) | (std::getenv("OBNSL_PROBE_NOLOAD") ? 4 : 0) |
                             (std::getenv("OBNSL_PROBE_NOATOMIC") ? 8 : 0) |
                             (std::getenv("OBNSL_NO_2X2") ? 16 : 0) |
                             (std::getenv("OBNSL_PROBE_SAMELOAD") ? 32 : 0) |
                             (std::getenv("OBNSL_PROBE_NOWRITE") ? 64 : 0) | (std::getenv("OBNSL_PROBE_NOISSUE") ? 128 : 0) |
                             (std::getenv("OBNSL_PROBE_NOSTORE") ? 256 : 0);
  if (plan->mfma) {
    obnsl_status st = build_units(b, steps_total, std::max(want, b->nitems));
    if (st != OBNSL_OK) return st;
  }
  int atomics = ((plan->mfma ? b->units_split : ksplit > 1) ? 1 : 0) | nojskip;
  const bool image = image_ok && (atomics & 1);
  uint32_t* table = counts;
  if (image) {
    HIP_TRY(b->img.ensure(img_cells));
    if (b->img_zeroed_epoch != d->epoch)
      HIP_TRY(hipMemsetAsync(b->img.p, 0, img_cells * sizeof(uint32_t), d->count_stream()));
    b->img_zeroed_epoch = 0;   // about to be added into
    b->img_wants_zero = true;
    atomics |= 1024;
    counts = b->img.p;
    if (table == plan->red.p) plan->wants_zero = false;  // every cell is stored, none added into
  } else {
    if ((atomics & 1) && !prezeroed)
      HIP_TRY(hipMemsetAsync(counts, 0, b->ncells * sizeof(uint32_t), d->count_stream()));
    if (counts == plan->red.p) plan->wants_zero = (atomics & 1) != 0;
  }
  if (d->ev_used >= 4096) {  // bound the pool: fold what has been recorded so far
    obnsl_status st = resolve_events(d);
    if (st != OBNSL_OK) return st;
  }
  if (d->ev_used == d->ev.size()) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    d->ev.push_back({e0, e1});
    d->ev_size.push_back(0);
  }
  d->ev_size[d->ev_used] = plan->k;
  auto& ev = d->ev[d->ev_used++];
  HIP_TRY(hipEventRecord(ev.first, d->count_stream()));
  const uint32_t grid = b->nitems * ksplit;
  const uint32_t kcn = plan->kc == 0 ? 1 : plan->kc;  // kc == 0: the single ONES column
  switch (plan->kr * 10 + kcn) {
    case 11: launch_bitgemm<1, 1>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 21: launch_bitgemm<2, 1>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 22: launch_bitgemm<2, 2>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 32: launch_bitgemm<3, 2>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 33: launch_bitgemm<3, 3>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 43: launch_bitgemm<4, 3>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    case 44: launch_bitgemm<4, 4>(d, plan, b, counts, grid, steps_total, sps, atomics); break;
    default: return fail(OBNSL_ERR_RUNTIME, "internal: no bit-GEMM instance");
  }
  HIP_TRY(hipGetLastError());
  if (image) {
    if (plan->tile == 256)
      hipLaunchKernelGGL(image_scatter_kernel<256>, dim3(b->nitems * (256 / 8)), dim3(256), 0, d->count_stream(),
                         b->img.p, b->items.p, b->rows.p, b->nrows, table);
    else
      hipLaunchKernelGGL(image_scatter_kernel<128>, dim3(b->nitems * (128 / 8)), dim3(256), 0, d->count_stream(),
                         b->img.p, b->items.p, b->rows.p, b->nrows, table);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(ev.second, d->count_stream()));
  return OBNSL_OK;
}

// Makes the main stream wait for one lane's outstanding count launches.
obnsl_status join_lane(obnsl_data* d, int lane) {
  if (!d->lane_plans[lane].empty()) {
    HIP_TRY(hipStreamWaitEvent(d->stream, d->side_join[lane], 0));
    for (Plan* p : d->lane_plans[lane]) p->pending_lane = -1;
    d->lane_plans[lane].clear();
  }
  return OBNSL_OK;
}

obnsl_status join_side(obnsl_data* d) {
  obnsl_status st = join_lane(d, 0);
  return st == OBNSL_OK ? join_lane(d, 1) : st;
}

// Reduced mode: make plan->red hold the counts of the current planes.
obnsl_status ensure_counts(obnsl_data* d, Plan* plan) {
  if (plan->counted_epoch == d->epoch && plan->awaits_sum)
    return fail(OBNSL_ERR_INVALID_ARGUMENT,
                "the count table of subset size " + std::to_string(plan->k) +
                    " holds one rank's slice: sum it over the ranks (obnsl_counts_export / "
                    "obnsl_counts_import) before scoring");
  plan->awaits_sum = false;
  if (plan->counted_epoch == d->epoch) {
    if (plan->pending_lane >= 0) return join_lane(d, plan->pending_lane);
    return OBNSL_OK;
  }
  if (!plan->red.p) {
    HIP_TRY(plan->red.ensure(std::max<uint64_t>(plan->batches[0]->ncells, 1)));
    plan->zeroed_epoch = 0;  // fresh memory: not cleared by any pack
  }
  obnsl_status st = launch_count(d, plan, plan->batches[0].get(), plan->red.p,
                                 plan->zeroed_epoch == d->epoch);
  if (st != OBNSL_OK) return st;
  plan->counted_epoch = d->epoch;
  return OBNSL_OK;
}

// Reduced mode: expand + evaluate the functional G for every table of `plan`
// (and the scores, when `scores` is given) -- one score_reduced_kernel launch.
obnsl_status run_functional(obnsl_data* d, Plan* plan, const ReducedViews& views, uint32_t k,
                            uint64_t max_cells, obnsl_score_kind kind, double ess,
                            const double* gprev, double g0, double* scores) {
  HIP_TRY(plan->G.ensure(std::max<uint64_t>(binom_host(d->n, k), 1)));
  const double* lg_alpha = nullptr;
  if (kind == OBNSL_SCORE_BDEU) {
    auto& slot = d->lg_alpha[ess];
    if (!slot) {
      std::vector<double> h(REDUCED_MAX_CELLS + 1, 0.0);
      for (uint64_t c = 1; c <= REDUCED_MAX_CELLS; ++c) h[c] = std::lgamma(ess / (double)c);
      slot = std::make_unique<DevBuf<double>>();
      HIP_TRY(slot->upload(h));
    }
    lg_alpha = slot->p;
  }
  Batch* b = plan->batches[0].get();
  if (b->ntables) {
    // a few consecutive tables per wavefront amortise the descriptor loads and
    // the dispatch; keep at least ~16 workgroups per CU for latency hiding
    static const uint32_t tpw_env = [] {
      const char* e = std::getenv("OBNSL_TABLES_PER_WG");
      return e ? (uint32_t)std::atoi(e) : 0u;
    }();
    uint32_t tpw = (uint32_t)std::min<uint64_t>(4, std::max<uint64_t>(2, b->ntables / 2048));
    if (tpw_env) tpw = tpw_env;
    // four tables per wavefront when the tables are small (full tables of at most 256
    // cells, 32 counted cells per table on average over the whole size), else two.  A
    // property of the subset size alone: G of a size must come out bit for bit the same
    // whether it is computed for its own scores, for the next size's, whole or in shards
    // (the two forms sum a table's cells in different orders).
    std::vector<uint32_t> sorted_card(d->card);
    std::sort(sorted_card.rbegin(), sorted_card.rend());
    uint64_t own_max_cells = 1, all_tables = binom_host(d->n, k);
    for (uint32_t i = 0; i < k; ++i) own_max_cells *= sorted_card[i];
    const uint32_t parts = (own_max_cells <= 256 && plan->size_cells <= 32 * all_tables) ? 4 : 2;
    tpw = (tpw + parts - 1) / parts * parts;  // a wavefront takes its tables `parts` at a time
    const uint32_t grid = (uint32_t)((b->ntables + tpw - 1) / tpw);
    const uint32_t e_stride = (uint32_t)((max_cells + 3) & ~3ull);
    if (b->wg_rg_tpw != tpw) {   // row group of every workgroup's first table (once per plan)
      std::vector<uint32_t> h(grid);
      size_t g = 0;
      for (uint32_t w = 0; w < grid; ++w) {
        const uint64_t t0 = (uint64_t)w * tpw;
        while (g + 1 < b->h_tbl_start.size() && b->h_tbl_start[g + 1] <= t0) ++g;
        h[w] = (uint32_t)g;
      }
      HIP_TRY(b->wg_rg.upload(h));
      b->wg_rg_tpw = tpw;
    }
#define OBNSL_LAUNCH_SCORER_P(KT, P)                                                             \
  hipLaunchKernelGGL((score_reduced_kernel<KT, P>), dim3(grid), dim3(64),                        \
                     (size_t)P * e_stride * sizeof(uint32_t), d->stream, views, b->rgs.p,        \
                     b->nrgs, b->ntables, tpw, d->d_card.p, d->d_binom.p, (uint32_t)d->N,        \
                     (int)kind, ess, plan->G.p, gprev, g0, scores, lg_alpha, e_stride, b->wg_rg.p)
#define OBNSL_LAUNCH_SCORER(KT)            \
  do {                                     \
    if (parts == 4) OBNSL_LAUNCH_SCORER_P(KT, 4); \
    else OBNSL_LAUNCH_SCORER_P(KT, 2);     \
  } while (0)
    switch (k) {
      case 1: OBNSL_LAUNCH_SCORER(1); break;
      case 2: OBNSL_LAUNCH_SCORER(2); break;
      case 3: OBNSL_LAUNCH_SCORER(3); break;
      case 4: OBNSL_LAUNCH_SCORER(4); break;
      case 5: OBNSL_LAUNCH_SCORER(5); break;
      case 6: OBNSL_LAUNCH_SCORER(6); break;
      case 7: OBNSL_LAUNCH_SCORER(7); break;
      default: OBNSL_LAUNCH_SCORER(8); break;
    }
#undef OBNSL_LAUNCH_SCORER_P
#undef OBNSL_LAUNCH_SCORER
    HIP_TRY(hipGetLastError());
  }
  plan->g_epoch = d->epoch;
  plan->g_kind = (int)kind;
  plan->g_ess = ess;
  return OBNSL_OK;
}

// Histogram path for the size-k subsets with colex ranks [lo, hi): batches of
// tables, each batch = zero the dense counters, one count_hist_kernel launch,
// one score_tables_kernel launch.
obnsl_status run_hist_range(obnsl_data* d, uint32_t k, uint64_t lo, uint64_t hi,
                            obnsl_score_kind kind, double ess, double* d_scores) {
  const uint64_t cell_budget = 1ull << 28;  // counters per batch (1 GiB)
  auto key = std::make_tuple(k, lo, hi);
  auto it = d->hist_plans.find(key);
  if (it == d->hist_plans.end()) {
    // table descriptors of the range, cut into batches (built once, kept)
    std::vector<std::unique_ptr<HistBatch>> batches;
    std::vector<uint32_t> vars(k);
    uint64_t t = lo;
    while (t < hi) {
      auto hb = std::make_unique<HistBatch>();
      std::vector<uint64_t> off;
      std::vector<TableDesc> tds;
      hb->rank0 = t;
      while (t < hi && tds.size() < (1u << 22)) {
        obnsl_status st = obnsl_unrank(t, k, vars.data());
        if (st != OBNSL_OK) return st;
        TableDesc td;
        std::memset(&td, 0, sizeof(td));
        td.k = k;
        uint64_t cells = 1;
        for (uint32_t i = 0; i < k; ++i) {
          td.cards[i] = d->card[vars[i]];
          cells *= d->card[vars[i]];
          if (cells > (1ull << 26)) return fail(OBNSL_ERR_UNSUPPORTED, "table has more than 2^26 cells");
        }
        if (!tds.empty() && hb->cells_total + cells > cell_budget) break;
        td.off = hb->cells_total;
        off.push_back(hb->cells_total);
        tds.push_back(td);
        hb->cells_total += cells;
        hb->max_cells = std::max<uint32_t>(hb->max_cells, (uint32_t)cells);
        ++t;
      }
      hb->ntables = (uint32_t)tds.size();
      HIP_TRY(hb->off.upload(off));
      HIP_TRY(hb->tds.upload(tds));
      batches.push_back(std::move(hb));
    }
    it = d->hist_plans.emplace(key, std::move(batches)).first;
  }
  for (auto& hbp : it->second) {
    HistBatch* hb = hbp.get();
    const uint32_t nt = hb->ntables;
    if (nt == 0) continue;
    HIP_TRY(d->counts.ensure(hb->cells_total));
    HIP_TRY(hipMemsetAsync(d->counts.p, 0, hb->cells_total * sizeof(uint32_t), d->stream));
    // enough workgroups to fill the chip, chunks a multiple of one iteration's rows
    uint64_t nchunks = std::max<uint64_t>(1, std::min<uint64_t>(4096 / nt + 1,
                                                                (d->N + HIST_ROWS_PER_ITER - 1) / HIST_ROWS_PER_ITER));
    uint64_t rpc = (d->N + nchunks - 1) / nchunks;
    rpc = (rpc + HIST_ROWS_PER_ITER - 1) / HIST_ROWS_PER_ITER * HIST_ROWS_PER_ITER;
    nchunks = (d->N + rpc - 1) / rpc;
    const size_t lds = (size_t)std::min<uint32_t>(hb->max_cells, HIST_LDS_CELLS) * sizeof(uint32_t);
    if (d->ev_used >= 4096) {  // bound the pool: fold what has been recorded so far
      obnsl_status st = resolve_events(d);
      if (st != OBNSL_OK) return st;
    }
    if (d->ev_used == d->ev.size()) {
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      d->ev.push_back({e0, e1});
      d->ev_size.push_back(0);
    }
    d->ev_size[d->ev_used] = k;
    auto& ev = d->ev[d->ev_used++];
    HIP_TRY(hipEventRecord(ev.first, d->stream));
    if (k <= (uint32_t)KFAST)
      hipLaunchKernelGGL(count_hist_kernel<KFAST>, dim3(nt * (uint32_t)nchunks), dim3(256), lds, d->stream,
                         d->codes.p, d->stride, d->N, d->n, d->d_card.p, d->d_binom.p, k, hb->rank0,
                         (uint32_t)nchunks, rpc, hb->off.p, d->counts.p);
    else
      hipLaunchKernelGGL(count_hist_kernel<KMAX>, dim3(nt * (uint32_t)nchunks), dim3(256), lds, d->stream,
                         d->codes.p, d->stride, d->N, d->n, d->d_card.p, d->d_binom.p, k, hb->rank0,
                         (uint32_t)nchunks, rpc, hb->off.p, d->counts.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.second, d->stream));
    hipLaunchKernelGGL(score_tables_kernel, dim3((nt + 3) / 4), dim3(256), 0, d->stream, d->counts.p,
                       hb->tds.p, (uint64_t)nt, (int)kind, ess, d_scores + hb->rank0 * k);
    HIP_TRY(hipGetLastError());
  }
  return OBNSL_OK;
}

// Which counting path serves subset size k.  Reduced mode (count only states
// 1..r-1 with the bit-GEMM, expand in the scorer) whenever the largest full table
// of this size fits the scorer's LDS AND the cost model prefers it: the
// bit-plane form costs ~ (reduced cells) x (32-row words), the per-row histogram
// ~ (tables) x (rows); measured on MI355X 6.7e-5 ns per cell-word (at ~80 % tile
// fill) against 4.5e-4 ns per table-row, i.e. the histogram wins above ~215
// reduced cells per table.
obnsl_status choose_reduced(obnsl_data* d, uint32_t k, bool* use_reduced, uint64_t* max_cells_out) {
  std::vector<uint32_t> sorted_card(d->card);
  std::sort(sorted_card.rbegin(), sorted_card.rend());
  uint64_t max_cells = 1;
  for (uint32_t i = 0; i < k; ++i) max_cells *= sorted_card[i];
  if (max_cells_out) *max_cells_out = max_cells;
  static const bool allow_reduced = std::getenv("OBNSL_NO_REDUCED") == nullptr;
  static const bool force_hist = std::getenv("OBNSL_FORCE_HIST") != nullptr;
  static const bool force_bits = std::getenv("OBNSL_FORCE_BITGEMM") != nullptr;
  bool use = allow_reduced && max_cells <= REDUCED_MAX_CELLS && !force_hist && k <= (uint32_t)KFAST;
  if (use && !force_bits) {
    // reduced cells of all size-k tables = e_k(r_1 - 1, ..., r_n - 1), the elementary
    // symmetric polynomial of the reduced radices (no plan is built for an estimate)
    std::vector<double> e(k + 1, 0.0);
    e[0] = 1.0;
    for (uint32_t v = 0; v < d->n; ++v)
      for (uint32_t j = std::min(k, v + 1); j >= 1; --j) e[j] += e[j - 1] * (double)(d->card[v] - 1);
    const double est_bit = e[k] * (double)d->wp * (use_mfma_gemm() ? 2.0e-5 : 6.7e-5);
    const double est_hist = (double)binom_host(d->n, k) * (double)d->N * 4.5e-4;
    if (est_hist < est_bit) use = false;
  }
  *use_reduced = use;
  return OBNSL_OK;
}

obnsl_status check_kind(obnsl_score_kind kind) {
  if (kind == OBNSL_SCORE_BDEU || kind == OBNSL_SCORE_BIC) return OBNSL_OK;
  return fail(OBNSL_ERR_UNSUPPORTED, "Unsupported score type.");
}

obnsl_status set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return OBNSL_OK;
}

obnsl_status run_pack(obnsl_data* d, uint32_t* bad_code = nullptr) {
  obnsl_status js = join_side(d);  // a count launch may still be reading the planes
  if (js != OBNSL_OK) return js;
  // count tables that were filled with atomics for the planes being replaced will
  // be filled the same way for the new ones: clear them here instead of one fill each
  ZeroList zl;
  std::memset(&zl, 0, sizeof(zl));
  Plan* cleared[ZERO_MAX];
  int ncleared = 0;
  Batch* cleared_img[ZERO_MAX];
  int nimg = 0;
  for (auto& kv : d->plans) {
    Plan* p = kv.second.get();
    if (zl.count == ZERO_MAX) break;
    if (!p->reduced || p->counted_epoch != d->epoch) continue;
    Batch* b0 = p->batches.empty() ? nullptr : p->batches[0].get();
    if (b0 && b0->img_wants_zero && b0->img.p) {   // the tile images of its split launch
      cleared_img[nimg++] = b0;
      zl.p[zl.count] = b0->img.p;
      zl.n[zl.count] = (uint64_t)b0->nitems * p->tile * p->tile;
      ++zl.count;
      if (zl.count == ZERO_MAX) break;
    }
    if (!p->wants_zero || !p->red.p) continue;
    if (p->batches[0]->ncells > (1ull << 26)) continue;  // large tables keep their own fill
    cleared[ncleared++] = p;
    zl.p[zl.count] = p->red.p;
    zl.n[zl.count] = p->batches[0]->ncells;
    ++zl.count;
  }
  dim3 grid((d->wp + 255) / 256, d->n + 1);
  hipLaunchKernelGGL(pack_planes_kernel, grid, dim3(256), 0, d->stream, d->codes.p, d->N,
                     d->stride, d->n, d->d_card.p, d->d_plane_base.p, d->planes.p, d->wp,
                     d->ones_plane, bad_code, zl);
  HIP_TRY(hipGetLastError());
  d->epoch++;  // every cached count table is stale now
  for (int i = 0; i < ncleared; ++i) cleared[i]->zeroed_epoch = d->epoch;
  for (int i = 0; i < nimg; ++i) {
    cleared_img[i]->img_zeroed_epoch = d->epoch;
    cleared_img[i]->img_wants_zero = false;
  }
  return OBNSL_OK;
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

const char* obnsl_last_error(void) { return g_err.c_str(); }
int obnsl_abi_version(void) { return 1; }

obnsl_status obnsl_device_count(int* count) {
  if (!count) return fail(OBNSL_ERR_INVALID_ARGUMENT, "count is null");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *count = c;
  return OBNSL_OK;
}

uint64_t obnsl_binomial(uint32_t n, uint32_t k) { return binom_host(n, k); }

obnsl_status obnsl_unrank(uint64_t rank, uint32_t k, uint32_t* out) {
  if (k > 64 || !out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "bad unrank arguments");
  for (uint32_t j = k; j >= 1; --j) {
    uint32_t v = j - 1;
    while (binom_host(v + 1, j) <= rank) ++v;
    out[j - 1] = v;
    rank -= binom_host(v, j);
  }
  return OBNSL_OK;
}

uint64_t obnsl_rank(const uint32_t* v, uint32_t k) {
  uint64_t r = 0;
  for (uint32_t j = 0; j < k; ++j) r += binom_host(v[j], j + 1);
  return r;
}

obnsl_status obnsl_data_create(const uint8_t* codes, uint64_t num_rows, uint64_t col_stride,
                               uint32_t num_vars, const uint32_t* cardinalities, int device,
                               int codes_on_device, obnsl_data** out) {
  if (!out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!codes || !cardinalities || num_vars == 0 || num_rows == 0)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "empty data set");
  if (col_stride < num_rows)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "col_stride is smaller than num_rows");
  if (num_rows >= (1ull << 32))
    return fail(OBNSL_ERR_UNSUPPORTED, "more than 2^32-1 rows");
  if (num_vars > 65535) return fail(OBNSL_ERR_UNSUPPORTED, "more than 65535 variables");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(OBNSL_ERR_NO_DEVICE, "no HIP device available: the local-score engine has no CPU path");
  if (device < 0 || device >= ndev)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  if (device < MAX_DEVICES) {
    // once per device, with that device current: where this GPU holds `scratch`
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
      unsigned long long* ptr = nullptr;
      HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
      g_scratch_ptr[device] = ptr;
    }
  }

  auto d = std::make_unique<obnsl_data>();
  d->device = device;
  d->N = num_rows;
  d->n = num_vars;
  d->card.assign(cardinalities, cardinalities + num_vars);
  d->plane_base.resize(num_vars);
  uint64_t planes = 0;
  for (uint32_t v = 0; v < num_vars; ++v) {
    if (d->card[v] == 0 || d->card[v] > 256)
      return fail(OBNSL_ERR_INVALID_ARGUMENT, "cardinality must be in [1, 256]");
    d->plane_base[v] = (uint32_t)planes;
    planes += d->card[v];
  }
  d->total_planes = (uint32_t)planes;
  d->ones_plane = d->total_planes;
  const uint64_t words = (num_rows + 31) / 32;
  d->wp = (uint32_t)((words + 63) / 64 * 64);
  d->stride = (uint64_t)d->wp * 32;
  if ((planes + 1) * (uint64_t)d->wp >= (1ull << 32))
    return fail(OBNSL_ERR_UNSUPPORTED, "row bitmaps exceed 2^32 words");

  HIP_TRY(d->codes.ensure((size_t)d->stride * num_vars));
  HIP_TRY(hipMemset(d->codes.p, 0, (size_t)d->stride * num_vars));
  HIP_TRY(hipMemcpy2D(d->codes.p, d->stride, codes, col_stride, num_rows, num_vars,
                      codes_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  HIP_TRY(d->planes.ensure((size_t)(planes + 1) * d->wp));
  HIP_TRY(d->d_card.upload(d->card));
  HIP_TRY(d->d_plane_base.upload(d->plane_base));
  std::vector<uint64_t> binom((size_t)(num_vars + 1) * (KMAX + 1));
  for (uint32_t v = 0; v <= num_vars; ++v)
    for (uint32_t j = 0; j <= (uint32_t)KMAX; ++j) binom[(size_t)v * (KMAX + 1) + j] = binom_host(v, j);
  HIP_TRY(d->d_binom.upload(binom));
  // the first pack also checks every code against its variable's cardinality:
  // a larger code would index past the tables of the per-row counting kernels
  DevBuf<uint32_t> bad;
  HIP_TRY(bad.ensure(1));
  HIP_TRY(hipMemset(bad.p, 0, sizeof(uint32_t)));
  obnsl_status st = run_pack(d.get(), bad.p);
  if (st != OBNSL_OK) return st;
  HIP_TRY(hipStreamSynchronize(d->stream));
  uint32_t bad_var = 0;
  HIP_TRY(hipMemcpy(&bad_var, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (bad_var)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "variable " + std::to_string(bad_var - 1) +
                                                " has a code that is not below its cardinality");
  *out = d.release();
  return OBNSL_OK;
}

obnsl_status obnsl_data_destroy(obnsl_data* d) {
  if (!d) return OBNSL_OK;
  (void)hipSetDevice(d->device);
  (void)hipDeviceSynchronize();
  delete d;
  return OBNSL_OK;
}

obnsl_status obnsl_data_set_stream(obnsl_data* d, void* hip_stream) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "data is null");
  d->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return OBNSL_OK;
}

obnsl_status obnsl_data_pack(obnsl_data* d) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "data is null");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  return run_pack(d);
}

obnsl_status obnsl_data_shape(const obnsl_data* d, uint64_t* num_rows, uint32_t* num_vars) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "data is null");
  if (num_rows) *num_rows = d->N;
  if (num_vars) *num_vars = d->n;
  return OBNSL_OK;
}

obnsl_status obnsl_num_subsets(const obnsl_data* d, uint32_t k, uint64_t* num) {
  if (!d || !num) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  *num = binom_host(d->n, k);
  return OBNSL_OK;
}

obnsl_status obnsl_count(obnsl_data* d, const uint32_t* var_ids, uint32_t k,
                         uint64_t* counts_out, uint64_t cap) {
  if (!d || !counts_out || (k && !var_ids)) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k > KMAX) return fail(OBNSL_ERR_UNSUPPORTED, "subset larger than OBNSL_MAX_SUBSET");
  for (uint32_t i = 0; i < k; ++i) {
    if (var_ids[i] >= d->n) return fail(OBNSL_ERR_OUT_OF_RANGE, "variable id out of range");
    if (i && var_ids[i] < var_ids[i - 1])
      return fail(OBNSL_ERR_INVALID_ARGUMENT, "var_ids must be sorted");
  }
  uint64_t cells = 1;
  std::vector<uint32_t> pbase(KMAX, 0), cards(KMAX, 1);
  for (uint32_t i = 0; i < k; ++i) {
    pbase[i] = d->plane_base[var_ids[i]];
    cards[i] = d->card[var_ids[i]];
    cells *= cards[i];
    if (cells > (1ull << 26)) return fail(OBNSL_ERR_UNSUPPORTED, "table has more than 2^26 cells");
  }
  if (cells > cap) return fail(OBNSL_ERR_INVALID_ARGUMENT, "counts_out is too small");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  OneTable ot;
  std::memset(&ot, 0, sizeof(ot));
  for (uint32_t i = 0; i < k; ++i) {
    ot.pbase[i] = pbase[i];
    ot.cards[i] = cards[i];
  }
  DevBuf<uint32_t>& dcnt = d->one_cnt;
  HIP_TRY(dcnt.ensure(cells));
  HIP_TRY(hipMemsetAsync(dcnt.p, 0, cells * sizeof(uint32_t), d->stream));
  const uint32_t gx = std::min<uint32_t>((d->wp + 255) / 256, 64);
  const uint32_t gy = (uint32_t)std::min<uint64_t>(cells, 1024);
  if (k <= (uint32_t)KFAST)
    hipLaunchKernelGGL(count_single_kernel<KFAST>, dim3(gx, gy), dim3(256), 0, d->stream, d->planes.p,
                       d->wp, k, ot, (uint32_t)cells, dcnt.p, d->ones_plane);
  else
    hipLaunchKernelGGL(count_single_kernel<KMAX>, dim3(gx, gy), dim3(256), 0, d->stream, d->planes.p,
                       d->wp, k, ot, (uint32_t)cells, dcnt.p, d->ones_plane);
  HIP_TRY(hipGetLastError());
  std::vector<uint32_t> h(cells);
  HIP_TRY(hipMemcpyAsync(h.data(), dcnt.p, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  for (uint64_t i = 0; i < cells; ++i) counts_out[i] = h[i];
  return OBNSL_OK;
}

obnsl_status obnsl_local_score(int device, const uint64_t* counts, const uint32_t* cards,
                               uint32_t k, uint32_t child_idx, obnsl_score_kind kind,
                               double ess, double* score_out) {
  if (!counts || !score_out || (k && !cards)) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k == 0 || k > KMAX) return fail(OBNSL_ERR_UNSUPPORTED, "subset size must be in [1, OBNSL_MAX_SUBSET]");
  if (child_idx >= k) return fail(OBNSL_ERR_INVALID_ARGUMENT, "child_var not present in contingency table");
  obnsl_status st = check_kind(kind);
  if (st != OBNSL_OK) return st;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(OBNSL_ERR_NO_DEVICE, "no HIP device available: the local-score engine has no CPU path");
  if (device < 0 || device >= ndev) return fail(OBNSL_ERR_INVALID_ARGUMENT, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  TableDesc td;
  std::memset(&td, 0, sizeof(td));
  td.k = k;
  uint64_t cells = 1;
  for (uint32_t i = 0; i < k; ++i) {
    if (cards[i] == 0) return fail(OBNSL_ERR_INVALID_ARGUMENT, "zero cardinality");
    td.cards[i] = cards[i];
    cells *= cards[i];
    if (cells > (1ull << 26)) return fail(OBNSL_ERR_UNSUPPORTED, "table has more than 2^26 cells");
  }
  std::vector<uint32_t> h(cells);
  for (uint64_t i = 0; i < cells; ++i) {
    if (counts[i] >= (1ull << 32)) return fail(OBNSL_ERR_UNSUPPORTED, "count exceeds 2^32-1");
    h[i] = (uint32_t)counts[i];
  }
  // per-device scratch kept for the life of the process (one call at a time per device)
  struct Scratch {
    DevBuf<uint32_t> cnt;
    DevBuf<double> out;
  };
  static std::mutex mu;
  static Scratch* scratch_of[MAX_DEVICES] = {nullptr};
  if (device >= MAX_DEVICES) return fail(OBNSL_ERR_INVALID_ARGUMENT, "device index out of range");
  std::lock_guard<std::mutex> lock(mu);
  if (!scratch_of[device]) scratch_of[device] = new Scratch();
  Scratch& sc = *scratch_of[device];
  HIP_TRY(sc.cnt.ensure(cells));
  HIP_TRY(sc.out.ensure(KMAX));
  HIP_TRY(hipMemcpyAsync(sc.cnt.p, h.data(), cells * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
  hipLaunchKernelGGL(score_one_kernel, dim3(1), dim3(64), 0, nullptr, sc.cnt.p, td, (int)kind, ess, sc.out.p);
  HIP_TRY(hipGetLastError());
  double res[KMAX];
  HIP_TRY(hipMemcpy(res, sc.out.p, k * sizeof(double), hipMemcpyDeviceToHost));
  *score_out = res[child_idx];
  return OBNSL_OK;
}

obnsl_status obnsl_score_all(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t k,
                             uint32_t shard, uint32_t num_shards, double* scores_out,
                             int out_on_device) {
  if (!d || !scores_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k == 0 || k > KMAX || k > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, min(num_vars, OBNSL_MAX_SUBSET)]");
  if (num_shards == 0 || shard >= num_shards)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "shard must be in [0, num_shards)");
  if (!out_on_device && num_shards != 1)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "sharded scoring needs out_on_device");
  obnsl_status st = check_kind(kind);
  if (st != OBNSL_OK) return st;
  st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  const uint64_t nsub = binom_host(d->n, k);
  DevBuf<double> tmp;
  double* d_scores = scores_out;
  if (!out_on_device) {
    HIP_TRY(tmp.ensure(nsub * k));
    d_scores = tmp.p;
  }

  uint64_t max_cells = 1;
  bool use_reduced = false;
  st = choose_reduced(d, k, &use_reduced, &max_cells);
  if (st != OBNSL_OK) return st;
  if (use_reduced) {
    ReducedViews views;
    std::memset(&views, 0, sizeof(views));
    Plan* own = nullptr;     // lists this shard's tables
    const double N_d = (double)d->N;
    const double g0 = kind == OBNSL_SCORE_BDEU ? std::lgamma(N_d + ess) - std::lgamma(ess)
                                               : N_d * std::log(N_d);
    const double* gprev = nullptr;
    for (uint32_t j = 1; j <= k; ++j) {
      Plan* full = nullptr;
      Plan* use = nullptr;
      if (j < k || num_shards == 1) {
        st = build_plan(d, j, 0, 1, true, &full);
        if (st != OBNSL_OK) return st;
        use = full;
      }
      if (j == k) {
        // G of the next smaller size, over ALL its subsets (cached per data epoch).
        // Queued before this size's counts are waited for: those may still be in
        // flight on a side stream, and this launch does not read them.
        if (k > 1) {
          Plan* prev = nullptr;
          st = build_plan(d, k - 1, 0, 1, true, &prev);
          if (st != OBNSL_OK) return st;
          if (!(prev->g_epoch == d->epoch && prev->g_kind == (int)kind && prev->g_ess == ess)) {
            ReducedViews pv = views;  // sizes 1..k-1 are the full plans already; v[k] is still empty
            st = run_functional(d, prev, pv, k - 1, max_cells, kind, ess, nullptr, 0.0, nullptr);
            if (st != OBNSL_OK) return st;
          }
          gprev = prev->G.p;
        }
        st = build_plan(d, k, shard, num_shards, true, &own);
        if (st != OBNSL_OK) return st;
        if (num_shards > 1) {
          // a small top size is cheaper to count whole than to count twice
          st = build_plan(d, k, 0, 1, true, &full);
          if (st != OBNSL_OK) return st;
          use = (full->total_items <= whole_below(full) || full->counted_epoch == d->epoch) ? full : own;
        }
      }
      st = ensure_counts(d, use);
      if (st != OBNSL_OK) return st;
      views.v[j].red = use->red.p;
      views.v[j].rowoff = use->rowoff.p;
      views.v[j].cgs = use->cgs.p;
      views.v[j].kr = use->kr;
      views.v[j].kc = use->kc;
    }
    st = run_functional(d, own, views, k, max_cells, kind, ess, gprev, g0, d_scores);
    if (st != OBNSL_OK) return st;
  } else if (std::getenv("OBNSL_FORCE_BITGEMM") == nullptr || k > (uint32_t)KFAST) {
    // large tables: per-row LDS histograms over this shard's contiguous rank range
    const uint64_t lo = nsub / num_shards * shard + std::min<uint64_t>(shard, nsub % num_shards);
    const uint64_t hi = nsub / num_shards * (shard + 1) + std::min<uint64_t>(shard + 1, nsub % num_shards);
    st = run_hist_range(d, k, lo, hi, kind, ess, d_scores);
    if (st != OBNSL_OK) return st;
  } else {
    Plan* plan = nullptr;
    st = build_plan(d, k, shard, num_shards, false, &plan);
    if (st != OBNSL_OK) return st;
    for (auto& bp : plan->batches) {
      Batch* b = bp.get();
      if (b->ntables == 0) continue;
      HIP_TRY(d->counts.ensure(std::max<uint64_t>(b->ncells, 1)));
      st = launch_count(d, plan, b, d->counts.p);
      if (st != OBNSL_OK) return st;
      const uint32_t sgrid = (uint32_t)((b->ntables + 3) / 4);
      hipLaunchKernelGGL(score_plan_kernel, dim3(sgrid), dim3(256), 0, d->stream, d->counts.p,
                         b->rgs.p, b->nrgs, plan->cgs.p, d->d_card.p, plan->kr, plan->kc,
                         b->ntables, (int)kind, ess, d_scores);
      HIP_TRY(hipGetLastError());
    }
  }
  if (!out_on_device) {
    HIP_TRY(hipMemcpyAsync(scores_out, d_scores, nsub * k * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
  }
  return OBNSL_OK;
}

}  // extern "C"

namespace {

obnsl_status precount_impl(obnsl_data* d, uint32_t kmax, uint32_t shard, uint32_t num_shards,
                           bool shard_lower, uint32_t* sum_size_out, uint64_t* sum_cells_out) {
  if (sum_size_out) *sum_size_out = 0;
  if (sum_cells_out) *sum_cells_out = 0;
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (kmax == 0 || kmax > (uint32_t)KMAX || kmax > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, min(num_vars, OBNSL_MAX_SUBSET)]");
  if (num_shards == 0 || shard >= num_shards)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "shard must be in [0, num_shards)");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  // sizes whose scorer reads reduced count tables (a size served by the
  // histogram path needs none, and neither do the sizes above it)
  uint32_t top_size = 0;
  const bool whole_request = true;
  for (uint32_t j = 1; j <= kmax; ++j) {
    bool red = false;
    st = choose_reduced(d, j, &red, nullptr);
    if (st != OBNSL_OK) return st;
    if (!red) break;
    top_size = j;
  }
  if (top_size == 0) return OBNSL_OK;  // nothing to share
  // if the requested top size itself goes through the histogram path, the sizes
  // below it are all counted whole (none of them is "the sharded top size")
  const bool top_is_requested = whole_request && top_size == kmax;
  kmax = top_size;
  if (!d->side[0]) {
    HIP_TRY(hipEventCreateWithFlags(&d->side_fork, hipEventDisableTiming));
    for (int l = 0; l < 2; ++l) {
      HIP_TRY(hipStreamCreateWithFlags(&d->side[l], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&d->side_join[l], hipEventDisableTiming));
    }
  }
  // The top size is counted on lane 0, the size below it on the main stream and
  // the smaller ones on lane 1, so that the scorers of the lower sizes -- queued
  // on the main stream by the obnsl_score_all calls that follow -- run while the
  // top size is still being counted.  The main stream joins a lane where one of
  // its count tables is first used (ensure_counts) or before the planes change
  // (run_pack).
  st = join_side(d);
  if (st != OBNSL_OK) return st;
  Plan* top = nullptr;
  st = build_plan(d, kmax, 0, 1, true, &top);
  if (st != OBNSL_OK) return st;
  if (top_is_requested && num_shards > 1 && top->total_items > whole_below(top)) {
    st = build_plan(d, kmax, shard, num_shards, true, &top);
    if (st != OBNSL_OK) return st;
  }
  // fork: the lanes start after everything queued so far (the pack)
  HIP_TRY(hipEventRecord(d->side_fork, d->stream));
  struct BesideLarge {   // holds for the launches queued by this call only
    obnsl_data* d;
    ~BesideLarge() { d->beside_large = false; }
  } beside_guard{d};
  d->beside_large = top->total_items > 32 && top->counted_epoch != d->epoch;
  if (top->counted_epoch != d->epoch) {
    HIP_TRY(hipStreamWaitEvent(d->side[0], d->side_fork, 0));
    d->count_lane = 0;
    st = ensure_counts(d, top);
    d->count_lane = -1;
    if (st != OBNSL_OK) return st;
    HIP_TRY(hipEventRecord(d->side_join[0], d->side[0]));
    top->pending_lane = 0;
    d->lane_plans[0].push_back(top);
  }
  // second largest size on the main stream; the small sizes below it are short,
  // latency-bound launches and run beside it on lane 1
  if (kmax >= 2) {
    Plan* full = nullptr;
    st = build_plan(d, kmax - 1, 0, 1, true, &full);
    if (st != OBNSL_OK) return st;
    if (shard_lower && top_is_requested && num_shards > 1 &&
        full->total_items >= num_shards && full->counted_epoch != d->epoch) {
      // this rank's slice only, written at its place in the whole table; the
      // rest stays zero and the caller sums the tables of all ranks
      Plan* part = nullptr;
      st = build_plan(d, kmax - 1, shard, num_shards, true, &part);
      if (st != OBNSL_OK) return st;
      const Batch* wb = full->batches[0].get();
      Batch* pb = part->batches[0].get();
      if (part->slice_off + pb->ncells > wb->ncells)
        return fail(OBNSL_ERR_RUNTIME, "internal: slice outside the count table");
      if (!full->red.p) {
        HIP_TRY(full->red.ensure(std::max<uint64_t>(wb->ncells, 1)));
        full->zeroed_epoch = 0;
      }
      if (full->zeroed_epoch != d->epoch)
        HIP_TRY(hipMemsetAsync(full->red.p, 0, wb->ncells * sizeof(uint32_t), d->stream));
      full->wants_zero = true;
      st = launch_count(d, part, pb, full->red.p + part->slice_off, true);
      if (st != OBNSL_OK) return st;
      full->counted_epoch = d->epoch;
      full->awaits_sum = true;
      if (sum_size_out) *sum_size_out = kmax - 1;
      if (sum_cells_out) *sum_cells_out = wb->ncells;
    } else {
      st = ensure_counts(d, full);
      if (st != OBNSL_OK) return st;
    }
  }
  bool lane1_used = false;
  for (uint32_t j = kmax >= 2 ? kmax - 2 : 0; j >= 1; --j) {
    Plan* full = nullptr;
    st = build_plan(d, j, 0, 1, true, &full);
    if (st != OBNSL_OK) return st;
    if (full->counted_epoch == d->epoch) continue;
    if (!lane1_used) HIP_TRY(hipStreamWaitEvent(d->side[1], d->side_fork, 0));
    lane1_used = true;
    d->count_lane = 1;
    st = ensure_counts(d, full);
    d->count_lane = -1;
    if (st != OBNSL_OK) return st;
    full->pending_lane = 1;
    d->lane_plans[1].push_back(full);
  }
  if (lane1_used) HIP_TRY(hipEventRecord(d->side_join[1], d->side[1]));
  return st;
}

// The whole reduced count table of subset size k, for the export / import pair.
obnsl_status whole_table(obnsl_data* d, uint32_t k, Plan** out) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k == 0 || k > (uint32_t)KMAX || k > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, min(num_vars, OBNSL_MAX_SUBSET)]");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  auto it = d->plans.find(std::make_tuple(k, 0u, 1u, true));
  if (it == d->plans.end() || it->second->counted_epoch != d->epoch)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "no count table of this subset size for the current row bitmaps");
  *out = it->second.get();
  return OBNSL_OK;
}

}  // namespace

extern "C" {

obnsl_status obnsl_precount(obnsl_data* d, uint32_t kmax, uint32_t shard, uint32_t num_shards) {
  return precount_impl(d, kmax, shard, num_shards, false, nullptr, nullptr);
}

obnsl_status obnsl_precount_sharded(obnsl_data* d, uint32_t kmax, uint32_t shard,
                                    uint32_t num_shards, uint32_t* sum_size_out,
                                    uint64_t* sum_cells_out) {
  if (!sum_size_out || !sum_cells_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  return precount_impl(d, kmax, shard, num_shards, true, sum_size_out, sum_cells_out);
}

obnsl_status obnsl_counts_cells(obnsl_data* d, uint32_t k, uint64_t* cells_out) {
  Plan* p = nullptr;
  obnsl_status st = whole_table(d, k, &p);
  if (st != OBNSL_OK) return st;
  if (!cells_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  *cells_out = p->batches[0]->ncells;
  return OBNSL_OK;
}

obnsl_status obnsl_counts_pointer(obnsl_data* d, uint32_t k, uint32_t** d_table_out,
                                  uint64_t* cells_out) {
  Plan* p = nullptr;
  obnsl_status st = whole_table(d, k, &p);
  if (st != OBNSL_OK) return st;
  if (!d_table_out || !cells_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (p->pending_lane >= 0) {
    st = join_lane(d, p->pending_lane);
    if (st != OBNSL_OK) return st;
  }
  *d_table_out = p->red.p;
  *cells_out = p->batches[0]->ncells;
  return OBNSL_OK;
}

obnsl_status obnsl_counts_mark_summed(obnsl_data* d, uint32_t k) {
  Plan* p = nullptr;
  obnsl_status st = whole_table(d, k, &p);
  if (st != OBNSL_OK) return st;
  p->awaits_sum = false;
  return OBNSL_OK;
}

obnsl_status obnsl_reduced_table(obnsl_data* d, const uint32_t* var_ids, uint32_t k,
                                 uint32_t* table_out, uint64_t capacity) {
  if (!d || !var_ids || !table_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k == 0 || k > (uint32_t)KMAX || k > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, min(num_vars, OBNSL_MAX_SUBSET)]");
  for (uint32_t i = 0; i < k; ++i) {
    if (var_ids[i] >= d->n) return fail(OBNSL_ERR_OUT_OF_RANGE, "variable id out of range");
    if (i && var_ids[i] <= var_ids[i - 1])
      return fail(OBNSL_ERR_INVALID_ARGUMENT, "var_ids must be sorted");
  }
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  bool red = false;
  st = choose_reduced(d, k, &red, nullptr);
  if (st != OBNSL_OK) return st;
  if (!red)
    return fail(OBNSL_ERR_UNSUPPORTED, "tables of this subset size are counted per row, not by the bit-plane contraction");
  Plan* p = nullptr;
  st = build_plan(d, k, 0, 1, true, &p);
  if (st != OBNSL_OK) return st;
  st = ensure_counts(d, p);
  if (st != OBNSL_OK) return st;
  // the kc smallest variables are the column side, the others the row side
  uint64_t colrank = 0, rowrank = 0, M = 1, cells = 1;
  for (uint32_t j = 0; j < p->kc; ++j) {
    colrank += binom_host(var_ids[j], j + 1);
    cells *= d->card[var_ids[j]] - 1;
  }
  for (uint32_t j = 0; j < p->kr; ++j) {
    rowrank += binom_host(var_ids[p->kc + j], j + 1);
    M *= d->card[var_ids[p->kc + j]] - 1;
  }
  cells *= M;
  if (cells > capacity) return fail(OBNSL_ERR_INVALID_ARGUMENT, "table_out is too small");
  const uint64_t col_before = p->kc ? p->h_col_before[colrank] : 0;
  const uint64_t base = p->h_rowoff[rowrank] + col_before * M;
  if (cells)
    HIP_TRY(hipMemcpyAsync(table_out, p->red.p + base, cells * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return OBNSL_OK;
}

obnsl_status obnsl_counts_export(obnsl_data* d, uint32_t k, uint32_t* d_dst) {
  Plan* p = nullptr;
  obnsl_status st = whole_table(d, k, &p);
  if (st != OBNSL_OK) return st;
  if (!d_dst) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (p->pending_lane >= 0) {
    st = join_lane(d, p->pending_lane);
    if (st != OBNSL_OK) return st;
  }
  HIP_TRY(hipMemcpyAsync(d_dst, p->red.p, p->batches[0]->ncells * sizeof(uint32_t),
                         hipMemcpyDeviceToDevice, d->stream));
  return OBNSL_OK;
}

obnsl_status obnsl_counts_import(obnsl_data* d, uint32_t k, const uint32_t* d_src) {
  Plan* p = nullptr;
  obnsl_status st = whole_table(d, k, &p);
  if (st != OBNSL_OK) return st;
  if (!d_src) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (p->pending_lane >= 0) {
    st = join_lane(d, p->pending_lane);
    if (st != OBNSL_OK) return st;
  }
  HIP_TRY(hipMemcpyAsync(p->red.p, d_src, p->batches[0]->ncells * sizeof(uint32_t),
                         hipMemcpyDeviceToDevice, d->stream));
  p->awaits_sum = false;
  return OBNSL_OK;
}

obnsl_status obnsl_score_subsets(obnsl_data* d, obnsl_score_kind kind, double ess,
                                 const uint32_t* subsets, uint32_t k, uint64_t m,
                                 double* scores_out) {
  if (!d || !subsets || !scores_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (k == 0 || k > KMAX) return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, OBNSL_MAX_SUBSET]");
  obnsl_status st = check_kind(kind);
  if (st != OBNSL_OK) return st;
  st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  std::vector<TableDesc> tds(m);
  std::vector<uint32_t> pb(m * KMAX, 0);
  uint64_t off = 0;
  for (uint64_t t = 0; t < m; ++t) {
    TableDesc& td = tds[t];
    std::memset(&td, 0, sizeof(td));
    td.k = k;
    td.off = off;
    uint64_t cells = 1;
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t v = subsets[t * k + i];
      if (v >= d->n) return fail(OBNSL_ERR_OUT_OF_RANGE, "variable id out of range");
      if (i && v <= subsets[t * k + i - 1])
        return fail(OBNSL_ERR_INVALID_ARGUMENT, "var_ids must be sorted");
      td.cards[i] = d->card[v];
      pb[t * KMAX + i] = d->plane_base[v];
      cells *= d->card[v];
      if (cells > (1ull << 26)) return fail(OBNSL_ERR_UNSUPPORTED, "table has more than 2^26 cells");
    }
    off += cells;
  }
  DevBuf<uint32_t>&dpb = d->list_pb, &dcnt = d->list_cnt;
  DevBuf<TableDesc>& dtd = d->list_td;
  DevBuf<double>& dout = d->list_out;
  uint32_t max_cells = 1;
  for (uint64_t t = 0; t < m; ++t)
    max_cells = std::max<uint32_t>(max_cells, (uint32_t)((t + 1 < m ? tds[t + 1].off : off) - tds[t].off));
  HIP_TRY(dpb.ensure(std::max<size_t>(pb.size(), 1)));
  HIP_TRY(dtd.ensure(std::max<size_t>(tds.size(), 1)));
  if (m) {
    HIP_TRY(hipMemcpyAsync(dpb.p, pb.data(), pb.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(dtd.p, tds.data(), tds.size() * sizeof(TableDesc), hipMemcpyHostToDevice, d->stream));
  }
  HIP_TRY(dcnt.ensure(std::max<uint64_t>(off, 1)));
  HIP_TRY(dout.ensure(std::max<uint64_t>(m * k, 1)));
  HIP_TRY(hipMemsetAsync(dcnt.p, 0, off * sizeof(uint32_t), d->stream));
  // row chunks: about 8192 words each for long lists, more of them when the list
  // is too short to fill the chip otherwise
  const uint32_t wblocks = (d->wp + 255) / 256;
  uint32_t gz = std::max<uint32_t>(1, std::min<uint32_t>(d->wp / 8192, 64));
  if (m < 64) gz = std::max<uint32_t>(gz, (uint32_t)std::min<uint64_t>(64 / std::max<uint64_t>(m, 1), 64));
  gz = std::min(gz, wblocks);
  const uint32_t gy = std::min<uint32_t>(max_cells, m > 64 ? 64u : 1024u);
  for (uint64_t t0 = 0; t0 < m; t0 += (1u << 30)) {
    const uint32_t gx = (uint32_t)std::min<uint64_t>(1u << 30, m - t0);
    if (k <= (uint32_t)KFAST)
      hipLaunchKernelGGL(count_list_kernel<KFAST>, dim3(gx, gy, gz), dim3(256), 0, d->stream, d->planes.p,
                         d->wp, k, dpb.p + t0 * KMAX, dtd.p + t0, dcnt.p, d->ones_plane);
    else
      hipLaunchKernelGGL(count_list_kernel<KMAX>, dim3(gx, gy, gz), dim3(256), 0, d->stream, d->planes.p,
                         d->wp, k, dpb.p + t0 * KMAX, dtd.p + t0, dcnt.p, d->ones_plane);
  }
  HIP_TRY(hipGetLastError());
  if (m) {
    hipLaunchKernelGGL(score_tables_kernel, dim3((uint32_t)((m + 3) / 4)), dim3(256), 0, d->stream,
                       dcnt.p, dtd.p, m, (int)kind, ess, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(scores_out, dout.p, m * k * sizeof(double), hipMemcpyDeviceToHost, d->stream));
  }
  HIP_TRY(hipStreamSynchronize(d->stream));
  return OBNSL_OK;
}

// `keys`: 2 n words of device memory, keys [0, n), ids [n, 2n); null = the data
// set's own buffer (allocated on first use).
static obnsl_status queue_best_parents(obnsl_data* d, const double* const* d_scores,
                                       uint32_t max_parents, unsigned long long* keys = nullptr) {
  if (max_parents + 1 > (uint32_t)KMAX || max_parents + 1 > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  if (!keys) {
    HIP_TRY(d->best_key.ensure(2 * (size_t)d->n));
    keys = d->best_key.p;
  }
  unsigned long long* key = keys;
  unsigned long long* id = keys + d->n;
  hipLaunchKernelGGL(best_init_kernel, dim3((d->n + 255) / 256), dim3(256), 0, d->stream, key, id, d->n);
  BestArgs args;
  std::memset(&args, 0, sizeof(args));
  args.kmax = max_parents + 1;
  for (uint32_t k = 1; k <= max_parents + 1; ++k) {
    const uint64_t nsub = binom_host(d->n, k);
    args.scores[k - 1] = d_scores[k - 1];
    args.nsubsets[k - 1] = nsub;
    args.block_start[k] = args.block_start[k - 1] + (uint32_t)std::min<uint64_t>((nsub + 255) / 256, 512);
  }
  for (int pass = 0; pass < 2; ++pass) {
    if (max_parents + 1 <= (uint32_t)KFAST)
      hipLaunchKernelGGL(best_parents_kernel<KFAST>, dim3(args.block_start[max_parents + 1]), dim3(256), 0,
                         d->stream, args, d->n, d->d_binom.p, key, id, pass);
    else
      hipLaunchKernelGGL(best_parents_kernel<KMAX>, dim3(args.block_start[max_parents + 1]), dim3(256), 0,
                         d->stream, args, d->n, d->d_binom.p, key, id, pass);
  }
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

obnsl_status obnsl_best_parents(obnsl_data* d, const double* const* d_scores,
                                uint32_t max_parents, double* best_score,
                                uint32_t* best_size, uint64_t* best_rank) {
  if (!d || !d_scores || !best_score || !best_size || !best_rank)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = queue_best_parents(d, d_scores, max_parents);
  if (st != OBNSL_OK) return st;
  std::vector<unsigned long long> hboth(2 * (size_t)d->n);
  HIP_TRY(hipMemcpyAsync(hboth.data(), d->best_key.p, hboth.size() * 8, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  const unsigned long long* hk = hboth.data();
  const unsigned long long* hi = hboth.data() + d->n;
  for (uint32_t v = 0; v < d->n; ++v) {
    if (hk[v] == 0) {  // nothing scored for this node in this shard
      best_score[v] = -INFINITY;
      best_size[v] = 0;
      best_rank[v] = 0;
      continue;
    }
    const uint64_t kbits = hk[v];
    const uint64_t bits = (kbits >> 63) ? (kbits & 0x7fffffffffffffffull) : ~kbits;
    double sc;
    std::memcpy(&sc, &bits, 8);
    best_score[v] = sc;
    best_size[v] = (uint32_t)(hi[v] >> 56);
    best_rank[v] = hi[v] & ((1ull << 56) - 1);
  }
  return OBNSL_OK;
}

obnsl_status obnsl_best_parents_device(obnsl_data* d, const double* const* d_scores,
                                       uint32_t max_parents, double* d_out) {
  if (!d || !d_scores || !d_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = queue_best_parents(d, d_scores, max_parents);
  if (st != OBNSL_OK) return st;
  hipLaunchKernelGGL(best_decode_kernel, dim3((d->n + 255) / 256), dim3(256), 0, d->stream,
                     d->best_key.p, d->best_key.p + d->n, d->n, d_out);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

obnsl_status obnsl_best_parents_device_noalloc(obnsl_data* d, const double* const* d_scores,
                                               uint32_t max_parents, double* d_out) {
  if (!d || !d_scores || !d_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (d->n > OBNSL_NOALLOC_MAX_VARS)
    return fail(OBNSL_ERR_UNSUPPORTED,
                "obnsl_best_parents_device_noalloc serves at most " +
                    std::to_string(OBNSL_NOALLOC_MAX_VARS) + " variables (this data set has " +
                    std::to_string(d->n) + "): use obnsl_best_parents_device");
  unsigned long long* ptr = d->device < MAX_DEVICES ? g_scratch_ptr[d->device] : nullptr;
  if (!ptr) return fail(OBNSL_ERR_RUNTIME, "no scratch address for this device");
  // the arg-max kernels start from best_init_kernel's fill of [0, 2 n): what an
  // earlier (possibly larger) call left in the array is never read
  obnsl_status st = queue_best_parents(d, d_scores, max_parents, ptr);
  if (st != OBNSL_OK) return st;
  hipLaunchKernelGGL(best_decode_kernel, dim3((d->n + 255) / 256), dim3(256), 0, d->stream, ptr,
                     ptr + d->n, d->n, d_out);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

obnsl_status obnsl_combine_best_device(obnsl_data* d, const double* d_parts, uint32_t world,
                                       double* d_out) {
  if (!d || !d_parts || !d_out || world == 0) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  hipLaunchKernelGGL(best_combine_kernel, dim3((d->n + 255) / 256), dim3(256), 0, d->stream, d_parts,
                     world, d->n, d_out);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

// Argument and capacity checks shared by the two search entry points.
static obnsl_status search_checks(obnsl_data* d, uint32_t max_parents) {
  const uint32_t n = d->n;
  // reference: exhaustive_search.cpp:18-23
  if (max_parents > n - 1)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  if (n > 64) return fail(OBNSL_ERR_INVALID_ARGUMENT, "num_of_vars must be in [0, 64]");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  // the tables grow as n * 2^n (the reference's :30-40 need n * 2^n * 16 bytes
  // and overflow `1 << n` from n = 31): refuse what the device cannot hold
  if (n > 31)
    return fail(OBNSL_ERR_INVALID_ARGUMENT,
                "num_of_vars must be in [0, 31] for the exact subset dynamic program "
                "(its tables grow as n * 2^n)");
  const uint64_t combs = 1ull << n, half = combs >> 1;
  const uint64_t need = combs * 9 + (uint64_t)n * half * 12 + (64ull << 20);
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (need > free_b)
    return fail(OBNSL_ERR_INVALID_ARGUMENT,
                "num_of_vars too large for the exact subset dynamic program on this device "
                "(its tables grow as n * 2^n)");
  return OBNSL_OK;
}

obnsl_status obnsl_exhaustive_search_tables(obnsl_data* d, const double* const* d_tables,
                                            uint32_t max_parents, uint64_t* parents_out,
                                            double* total_score_out) {
  if (!d || !d_tables || !parents_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = search_checks(d, max_parents);
  if (st != OBNSL_OK) return st;
  const uint32_t n = d->n;
  const uint64_t combs = 1ull << n, half = combs >> 1;
  DpScores sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.max_scored = max_parents + 1;
  for (uint32_t k = 1; k <= max_parents + 1; ++k) {
    if (!d_tables[k - 1]) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null score table");
    sc.tbl[k - 1] = d_tables[k - 1];
  }
  DevBuf<double> best_gs, bls, total;
  DevBuf<signed char> best_ch;
  DevBuf<uint32_t> bps;
  DevBuf<uint64_t> parents;
  HIP_TRY(best_gs.ensure(combs));
  HIP_TRY(best_ch.ensure(combs));
  HIP_TRY(bls.ensure(std::max<uint64_t>((uint64_t)n * half, 1)));
  HIP_TRY(bps.ensure(std::max<uint64_t>((uint64_t)n * half, 1)));
  HIP_TRY(parents.ensure(n));
  HIP_TRY(total.ensure(1));
  HIP_TRY(hipMemsetAsync(best_gs.p, 0, sizeof(double), d->stream));  // best_gs[empty set] = 0 (:42-44)
  const uint32_t grid = (uint32_t)std::min<uint64_t>((combs + 255) / 256, 1u << 20);
  for (uint32_t level = 1; level <= n; ++level)
    hipLaunchKernelGGL(dp_level_kernel, dim3(grid), dim3(256), 0, d->stream, n, level, sc,
                       d->d_binom.p, best_gs.p, best_ch.p, bls.p, bps.p);
  hipLaunchKernelGGL(dp_backtrack_kernel, dim3(1), dim3(64), 0, d->stream, n, best_gs.p, best_ch.p,
                     bps.p, parents.p, total.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(parents_out, parents.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, d->stream));
  double tot = 0.0;
  HIP_TRY(hipMemcpyAsync(&tot, total.p, sizeof(double), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  if (total_score_out) *total_score_out = tot;
  return OBNSL_OK;
}

obnsl_status obnsl_exhaustive_search(obnsl_data* d, obnsl_score_kind kind, double ess,
                                     uint32_t max_parents, uint64_t* parents_out,
                                     double* total_score_out) {
  if (!d || !parents_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = search_checks(d, max_parents);
  if (st != OBNSL_OK) return st;
  st = check_kind(kind);
  if (st != OBNSL_OK) return st;
  const uint32_t n = d->n;
  // scores of every subset size the search asks for, on the device
  std::vector<DevBuf<double>> tables(max_parents + 1);
  std::vector<const double*> ptrs(max_parents + 1);
  st = obnsl_precount(d, max_parents + 1, 0, 1);
  if (st != OBNSL_OK) return st;
  for (uint32_t k = 1; k <= max_parents + 1; ++k) {
    HIP_TRY(tables[k - 1].ensure(std::max<uint64_t>(binom_host(n, k) * k, 1)));
    st = obnsl_score_all(d, kind, ess, k, 0, 1, tables[k - 1].p, 1);
    if (st != OBNSL_OK) return st;
    ptrs[k - 1] = tables[k - 1].p;
  }
  return obnsl_exhaustive_search_tables(d, ptrs.data(), max_parents, parents_out, total_score_out);
}

obnsl_status obnsl_count_path(obnsl_data* d, uint32_t k, int* path_out) {
  if (!d || !path_out || k == 0 || k > (uint32_t)KMAX || k > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "bad obnsl_count_path arguments");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  bool red = false;
  st = choose_reduced(d, k, &red, nullptr);
  if (st != OBNSL_OK) return st;
  *path_out = !red ? OBNSL_PATH_HISTOGRAM : use_mfma_gemm() ? OBNSL_PATH_MFMA : OBNSL_PATH_VALU;
  return OBNSL_OK;
}

obnsl_status obnsl_stream_read(const void* d_src, uint64_t bytes, void* hip_stream, uint32_t* d_sink) {
  if (!d_src || !d_sink || bytes < 16 || (reinterpret_cast<uintptr_t>(d_src) & 15))
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "obnsl_stream_read needs a 16-byte aligned buffer of at least 16 bytes");
  hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 16), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(d_src),
                     bytes / 16, d_sink);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

obnsl_status obnsl_kernel_time(obnsl_data* d, uint32_t k, double* total_ms, uint32_t* launches,
                               int reset) {
  if (!d || !total_ms || !launches || k == 0 || k > (uint32_t)KMAX)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "bad obnsl_kernel_time arguments");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  st = resolve_events(d);
  if (st != OBNSL_OK) return st;
  *total_ms = d->acc_ms[k];
  *launches = d->acc_launches[k];
  if (reset)
    for (int i = 0; i <= KMAX; ++i) {
      d->acc_ms[i] = 0;
      d->acc_launches[i] = 0;
    }
  return OBNSL_OK;
}

}  // extern "C"
