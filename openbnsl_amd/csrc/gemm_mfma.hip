// gemm_mfma.hip -- the bit-plane contraction on the matrix cores.
//
// What the reference does per candidate (child, parent set): walk all N rows and
// build a sparse joint histogram of the subset
// (include/base/contingency_table.h:52-55).  Here a cell count of a joint table
// is popcount(AND of the cell's row bitmaps), and all tables of one subset size
// are one GEMM-shaped contraction over the N rows of the data: rows of A are the
// joint states of every kr-subset, rows of B those of every kc-subset, and
// C[row][col] = sum over samples of A[row][sample] AND B[col][sample].
//
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
// ANDed sample words of a 32-word slab of the tile's A and B rows; a lane reads
// 16 bytes (128 samples of its row) per operand and round.
#include <hip/hip_runtime.h>

#include "engine_state.h"

namespace obnsl_internal {
namespace {

// Timing probes exist only in builds with -DOBNSL_PROBES (tools/): they skip parts
// of the kernel and so produce wrong counts.  In the shipped library every
// OBNSL_PROBE(...) is the constant false and the code behind it is dropped.
#ifdef OBNSL_PROBES
#define OBNSL_PROBE(bit) ((probe & (bit)) != 0)
#else
#define OBNSL_PROBE(bit) false
#endif
enum ProbeBits : int {
  PROBE_NOLOAD = 1,     // stage only the first slab of a unit
  PROBE_SAMELOAD = 2,   // stage the unit's first slab again at every step (cache hits only)
  PROBE_NOWRITE = 4,    // issue the staging loads, skip the LDS stores
  PROBE_NOISSUE = 8,    // skip the staging loads, keep the LDS stores
  PROBE_NOSTORE = 16,   // the epilogue writes nothing
  PROBE_NOATOMIC = 32,  // plain stores where the epilogue would add
};

__device__ __forceinline__ uint4 and4(uint4 a, uint4 b) {
  return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w);
}

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
                                          AfterRound after_round, int rp_lo = 0, int rp_hi = 2) {
  // the words of round rd + 1 are read from LDS while round rd's instructions run
  const uint32_t* pa = &sA[arow * LDS_STRIDE + hoff];
  const uint32_t* pb = &sB[brow * LDS_STRIDE + hoff];
  uint4 wa[NA], wb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + 16 * rp_lo);
#pragma unroll
  for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + 16 * rp_lo);
  static_assert(KCW / 8 == 4, "two iterations of two rounds");
  // two rounds per iteration: the words of a round live in (wa, wb) and (na, nb) in turn
#pragma unroll 1
  for (int rp = rp_lo; rp < rp_hi; ++rp) {
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
// (rp_lo, rp_hi): the pairs of rounds of the slab this wave contracts -- (0, 2) the whole
// slab; a wave that shares its blocks with another takes one pair each.
template <int NA, int NB, class AfterRound>
__device__ __forceinline__ void tile_mfma_pipelined(const uint32_t* __restrict__ sA,
                                                    const uint32_t* __restrict__ sB, uint32_t arow,
                                                    uint32_t brow, uint32_t hoff, obnsl_v16f (&acc)[NA][NB],
                                                    AfterRound after_round, int rp_lo = 0, int rp_hi = 2) {
  const uint32_t* pa = &sA[arow * LDS_STRIDE + hoff];
  const uint32_t* pb = &sB[brow * LDS_STRIDE + hoff];
  uint4 wa[NA], wb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * LDS_STRIDE + 16 * rp_lo);
#pragma unroll
  for (int j = 0; j < NB; ++j) wb[j] = *reinterpret_cast<const uint4*>(pb + 32 * j * LDS_STRIDE + 16 * rp_lo);
  obnsl_v8i oa[NA], ob[NB];
  build_blocks<0, NA, NB>(wa, wb, oa, ob);
  static_assert(KCW / 8 == 4, "two iterations of two rounds");
  constexpr int VPM = (8 * (NA + NB) + NA * NB - 1) / (NA * NB);   // vector instructions per MFMA slot
  // two rounds per iteration: the words of a round live in (wa, wb) and (na, nb) in
  // turn, so no round ends with copying one set into the other
#pragma unroll 1
  for (int rp = rp_lo; rp < rp_hi; ++rp) {
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


// Everything after the tile is known, for one value of JM (a template
// parameter of the whole loop nest, so that the accumulators live in the
// accumulator registers from the first step to the epilogue).
// KS = 2 (tiles of one or two column blocks in a launch whose units add their counts): two
// waves share the 2 x NB blocks of a 64-row half of the tile and contract one half of every
// slab (a pair of rounds) each, so that edge tiles amortise their operand builds over
// more instructions (5 / 7.5 vector instructions per matrix instruction for two / one
// column blocks, against 7.5 / 10 with one row block per wave).  The waves' partial
// counts meet in the atomics of the epilogue.
// KS = 3 (a full-width tile on the staircase whose first 64 rows reach none of its columns):
// all four waves work on the other 64 rows -- two side by side over the four column
// blocks, two over the halves of every slab -- instead of two waves idling through the
// tile: about half the time.
// NARROW (the persistent kernels, whose launches are short: the host sends a data set there
// only if its bitmaps stay within 4 GB): the staging loads are buffer loads -- a scalar
// resource for the bitmaps, a 32-bit byte offset per thread and plane that does not change
// with the step, and the step's offset as a scalar -- instead of loads from 64-bit
// addresses: 16 registers and one vector instruction per load and step less.
template <int KR, int KC, int NA, int NB, int TB, int KS = 1, bool NARROW = false>
__device__ __forceinline__ void mfma_tile_body(
    uint32_t* __restrict__ smem, const uint32_t* __restrict__ planes, uint32_t wp,
    const RowDesc* __restrict__ rows, uint32_t nrows, const ColDesc* __restrict__ cols,
    uint32_t ncols_total, uint32_t row0, uint32_t col0, uint32_t* __restrict__ counts,
    uint32_t step_lo, uint32_t step_hi, bool atomics, uint32_t* __restrict__ image, int probe) {
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
    for (int j = 0; j < KR; ++j) sAo[tid * KR + j] = rows[r].plane[j] * wp * (NARROW ? 4u : 1u);
#pragma unroll
    for (int j = 0; j < KC; ++j) sBo[tid * KC + j] = cols[c].plane[j] * wp * (NARROW ? 4u : 1u);
  }
  const uint32_t st_off = rl * LDS_STRIDE + q * 4;
  // (dword 3 of the resource: 32-bit data, no swizzle, no range check short of 4 GB)
  const __amdgpu_buffer_rsrc_t plane_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(planes), 0, 0xffffffff, 0x00020000);
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
  static_assert(KS != 1 || NA == 1 || 2 * 32 * NB == TB, "two waves side by side cover the tile's columns");
  static_assert(KS == 1 || (TB == 128 && KS == 2 && NA == 2 && NB <= 2) || (TB == 128 && KS == 3 && NA == 2 && NB == 2),
                "the shared forms: 2 x NB blocks per wave, a half of every slab");
  const uint32_t lane = tid & 63, wv = tid >> 6;
  const uint32_t lr = lane & 31, h4 = (lane >> 5) * 4;
  // first row / column of the wave's blocks
  const uint32_t wrow = KS == 3 ? 64 : KS == 2 ? (wv & 1) * 64 : NA == 1 ? wv * 32 : (wv >> 1) * 64;
  const uint32_t wcol = KS == 3 ? (wv & 1) * (32 * NB) : KS == 2 ? 0 : NA == 1 ? 0 : (wv & 1) * (32 * NB);
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
      for (int j = 0; j < KR; ++j) aoff[p][j] = (rows[r].plane[j] * wp + q * 4) * (NARROW ? 4u : 1u);
#pragma unroll
      for (int j = 0; j < KC; ++j) boff[p][j] = (cols[c].plane[j] * wp + q * 4) * (NARROW ? 4u : 1u);
    }

    auto issue_loads = [&](uint32_t step) {
      const uint32_t w0 = (OBNSL_PROBE(PROBE_SAMELOAD) ? step_lo : step) * KCW;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if constexpr (NARROW) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int j = 0; j < KR; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)aoff[p][j], (int)(w0 * 4u), 0);
            ga[p][j] = make_uint4(v.x, v.y, v.z, v.w);
          }
#pragma unroll
          for (int j = 0; j < KC; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)boff[p][j], (int)(w0 * 4u), 0);
            gb[p][j] = make_uint4(v.x, v.y, v.z, v.w);
          }
        } else {
#pragma unroll
          for (int j = 0; j < KR; ++j) ga[p][j] = *reinterpret_cast<const uint4*>(planes + aoff[p][j] + w0);
#pragma unroll
          for (int j = 0; j < KC; ++j) gb[p][j] = *reinterpret_cast<const uint4*>(planes + boff[p][j] + w0);
        }
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
      const bool more = step + 1 < step_hi && !OBNSL_PROBE(PROBE_NOLOAD);
      if (more && !OBNSL_PROBE(PROBE_NOISSUE)) issue_loads(step + 1);
      const uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
      const uint32_t* sB = sA + TB * LDS_STRIDE;
      if constexpr (KS >= 2 && NB == 2) tile_mfma_pipelined<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {}, (int)(wv >> 1), (int)(wv >> 1) + 1);
      else if constexpr (KS >= 2) tile_mfma<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {}, (int)(wv >> 1), (int)(wv >> 1) + 1);
      else if constexpr (NA == 2 && NB == 2) tile_mfma_pipelined<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {});
      else tile_mfma<NA, NB>(sA, sB, wrow + lr, wcol + lr, h4, acc, [](int) {});
      if (more && !OBNSL_PROBE(PROBE_NOWRITE)) write_slab(buf ^ 1);
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
      const uint32_t r = rl + PR * p;
      if constexpr (NARROW) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t s0 = (OBNSL_PROBE(PROBE_SAMELOAD) ? step_lo : step) * (KCW * 4u);
#pragma unroll
        for (int j = 0; j < KR; ++j) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)(sAo[r * KR + j] + q * 16u), (int)s0, 0);
          ga[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, (int)(sBo[r * KC + j] + q * 16u), (int)s0, 0);
          gb[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
      } else {
        const uint32_t w0 = (OBNSL_PROBE(PROBE_SAMELOAD) ? step_lo : step) * KCW + q * 4;
#pragma unroll
        for (int j = 0; j < KR; ++j) ga[j] = *reinterpret_cast<const uint4*>(planes + sAo[r * KR + j] + w0);
#pragma unroll
        for (int j = 0; j < KC; ++j) gb[j] = *reinterpret_cast<const uint4*>(planes + sBo[r * KC + j] + w0);
      }
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
      const bool more = step + 1 < step_hi && !OBNSL_PROBE(PROBE_NOLOAD);
      if (more && !OBNSL_PROBE(PROBE_NOISSUE)) issue_pass(step + 1, 0);
      const uint32_t* sA = smem + buf * 2 * TB * LDS_STRIDE;
      const uint32_t* sB = sA + TB * LDS_STRIDE;
      auto stage_round = [&](int rd) {
        if (more) {
          // (`wp != 0` always holds.  Measured with hipcc 7.2: with the stores behind a
          // condition the compiler cannot resolve, the 2 x 4 form takes 245 registers;
          // with unconditional stores 256 and 44 spilled.)
          if (wp != 0 && !OBNSL_PROBE(PROBE_NOWRITE)) write_pass(buf ^ 1, rd);
          if (rd + 1 < NP && !OBNSL_PROBE(PROBE_NOISSUE)) issue_pass(step + 1, rd + 1);
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
          if (v && !OBNSL_PROBE(PROBE_NOSTORE)) atomicAdd(image + rt * TB + wcol + 32 * j + lr, v);
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
        if (OBNSL_PROBE(PROBE_NOSTORE)) continue;
        uint32_t* dst = counts + base + (uint64_t)c * M;
        if (atomics && !OBNSL_PROBE(PROBE_NOATOMIC)) {
          if (v) atomicAdd(dst, v);
        } else {
          *dst = v;
        }
      }
    }
}

// TB = 128: 4 waves, two workgroups per CU.  TB = 256: 8 waves, one workgroup per CU --
// half the plane words fetched and three quarters of the operand builds and LDS reads
// per instruction (a wave's full-tile share is 2 x 4 blocks), for launches of many tiles.
//
// RUNS = false: workgroup w works on unit w alone.
// RUNS = true: persistent workgroups, as many as the chip holds at once.  Run r is the
// units [unit_start[r], unit_start[r + 1]), worked on one after the other.  Workgroup w
// starts with run w (long runs of about equal cost, most of the launch) and then takes
// the remaining, shorter runs in order from a queue -- one atomic counter, `queue`, zero
// at launch -- until none is left, so that a workgroup that was slowed down (other
// kernels run beside this one) takes fewer of them and all finish together; no
// workgroup is dispatched after the first wave of them, which saves the gap between
// one workgroup's end and its successor's start once per unit.  Every workgroup leaves
// the loop on the first queue index past the last run.  (The loop costs registers
// that the 256 x 256 form does not have: RUNS goes with 128 x 128 tiles only.)
template <int KR, int KC, int TB, bool RUNS, bool NARROW>
__global__ __launch_bounds__(2 * TB, (TB == 128 && KR + KC <= 5) ? 2 : 1) void bitgemm_mfma_kernel(
    const uint32_t* __restrict__ planes, uint32_t wp, const RowDesc* __restrict__ rows,
    uint32_t nrows, const ColDesc* __restrict__ cols, uint32_t ncols_total,
    const MfmaUnit* __restrict__ units, const uint32_t* __restrict__ unit_start, uint32_t nruns,
    uint32_t* __restrict__ queue, uint32_t* __restrict__ counts, int flags, GemmParts extra
#ifdef OBNSL_PROBES
    , uint64_t* __restrict__ trace
#endif
) {
  // two slabs of A and of B (one array: a second __shared__ object beside it could make
  // the compiler drain loads before every LDS read), and behind them the plane
  // offsets of the tile's rows and columns
  constexpr uint32_t SLAB_WORDS = 2 * 2 * TB * LDS_STRIDE + (TB == 128 ? 0 : TB * (KR + KC));
  __shared__ __attribute__((aligned(16))) uint32_t smem[SLAB_WORDS + (RUNS ? 4 : 0)];
  const int probe = flags >> 16;
  (void)probe;
#ifdef OBNSL_PROBES
  const uint64_t t_begin = wall_clock64();
  uint32_t runs_done = 0;
#endif
  constexpr uint32_t JFULL = TB / 32;
#define OBNSL_MFMA_BODY(NA, NB)                                                                          \
  mfma_tile_body<KR, KC, NA, NB, TB, 1, NARROW>(smem, planes, wp, p_rows, p_nrows, p_cols, p_ncols, row0, col0, \
                                     p_counts, u.step_lo, u.step_hi, atomics, image, probe)
#define OBNSL_MFMA_BODY_SHARED(NA, NB, KS)                                                               \
  mfma_tile_body<KR, KC, NA, NB, TB, KS, NARROW>(smem, planes, wp, p_rows, p_nrows, p_cols, p_ncols, row0, col0, \
                                         p_counts, u.step_lo, u.step_hi, atomics, image, probe)
#define OBNSL_MFMA_UNIT(UI)                                                                       \
  {                                                                                               \
    const MfmaUnit u = units[UI];                                                                 \
    /* the part of the launch the unit belongs to: part 0 is described by the kernel's own    */ \
    /* arguments, the others (smaller subset sizes counted by the same launch) by `extra`     */ \
    const RowDesc* __restrict__ p_rows = rows;                                                    \
    const ColDesc* __restrict__ p_cols = cols;                                                    \
    uint32_t* __restrict__ p_counts = counts;                                                     \
    uint32_t p_nrows = nrows, p_ncols = ncols_total;                                              \
    bool p_image = (flags & GEMM_IMAGE) != 0;                                                     \
    if (RUNS && u.part != 0) {                                                                    \
      const GemmPart& e = extra.p[u.part - 1];                                                    \
      p_rows = e.rows;                                                                            \
      p_cols = e.cols;                                                                            \
      p_counts = e.counts;                                                                        \
      p_nrows = e.nrows;                                                                          \
      p_ncols = e.ncols_total;                                                                    \
      p_image = e.image != 0;                                                                     \
    }                                                                                             \
    const uint32_t row0 = u.row_tile * TB, col0 = u.col_tile * TB;                                \
    const uint32_t jm = (flags & GEMM_ALL_BLOCKS) ? JFULL : u.blocks;                             \
    const bool atomics = (flags & GEMM_ATOMIC) != 0;                                              \
    /* image mode: the part's `counts` is its array of tile images */                             \
    uint32_t* image = p_image ? p_counts + (size_t)u.tile * (TB * TB) : nullptr;                  \
    switch (jm) { /* uniform over the workgroup */                                                \
      case 1:                                                                                     \
        if constexpr (RUNS) OBNSL_MFMA_BODY_SHARED(2, 1, 2); /* (runs always add: partial counts */ \
        else OBNSL_MFMA_BODY(1, 1);                          /*  of several waves can meet)      */ \
        break;                                                                                    \
      case 2:                                                                                     \
        if constexpr (RUNS) OBNSL_MFMA_BODY_SHARED(2, 2, 2);                                      \
        else OBNSL_MFMA_BODY(1, 2);                                                               \
        break;                                                                                    \
      case 3: OBNSL_MFMA_BODY(1, 3); break;                                                       \
      default:                                                                                    \
        if constexpr (TB == 128) {                                                                \
          if (flags & GEMM_NO_2X2) OBNSL_MFMA_BODY(1, 4);                                         \
          else if (RUNS && u.top_blocks == 0) OBNSL_MFMA_BODY_SHARED(2, 2, 3);                    \
          else OBNSL_MFMA_BODY(2, 2);                                                             \
        } else {                                                                                  \
          if (jm == 4) OBNSL_MFMA_BODY(1, 4);                                                     \
          else OBNSL_MFMA_BODY(2, 4);                                                             \
        }                                                                                         \
        break;                                                                                    \
    }                                                                                             \
  }
  if constexpr (RUNS) {
    uint32_t run = blockIdx.x;
    while (run < nruns) {   // uniform over the workgroup
      // the index of the run after this one is asked for now and needed only at the end
      uint32_t grabbed = 0;
      if (threadIdx.x == 0) grabbed = atomicAdd(queue, 1u);
      const uint32_t u0 = unit_start[run], u1 = unit_start[run + 1];
      for (uint32_t ui = u0; ui < u1; ++ui) {
        OBNSL_MFMA_UNIT(ui)
        __syncthreads();   // the next unit refills the slabs
      }
      if (threadIdx.x == 0) smem[SLAB_WORDS] = grabbed;
      __syncthreads();
      // (every unit passes a barrier before anything is stored here again)
      run = gridDim.x + __builtin_amdgcn_readfirstlane(smem[SLAB_WORDS]);
#ifdef OBNSL_PROBES
      ++runs_done;
#endif
    }
  } else {
    OBNSL_MFMA_UNIT(blockIdx.x)
  }
#undef OBNSL_MFMA_UNIT
#undef OBNSL_MFMA_BODY
#undef OBNSL_MFMA_BODY_SHARED
#ifdef OBNSL_PROBES
  if (trace != nullptr && threadIdx.x == 0) {
    // per workgroup: begin, end (100 MHz counter), hardware id, XCC id | runs done << 8
    trace[4 * (size_t)blockIdx.x + 0] = t_begin;
    trace[4 * (size_t)blockIdx.x + 1] = wall_clock64();
    trace[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    trace[4 * (size_t)blockIdx.x + 3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xff) | ((uint64_t)runs_done << 8);
  }
#endif
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

// The same for the image parts of a launch over several subset sizes, in one launch:
// workgroups [first[p], first[p + 1]) serve part p (128 x 128 tiles).
__global__ __launch_bounds__(256) void image_scatter_parts_kernel(ScatterParts sp) {
  constexpr int TB = 128;
  uint32_t p = 0;
#pragma unroll
  for (int j = 1; j < GEMM_MAX_PARTS; ++j)
    if ((uint32_t)j < sp.nparts && blockIdx.x >= sp.first[j]) p = j;
  const uint32_t* __restrict__ images = sp.p[0].images;
  const uint2* __restrict__ items = sp.p[0].items;
  const RowDesc* __restrict__ rows = sp.p[0].rows;
  uint32_t* __restrict__ counts = sp.p[0].table;
  uint32_t nrows = sp.p[0].nrows;
#pragma unroll
  for (int j = 1; j < GEMM_MAX_PARTS; ++j)
    if (p == (uint32_t)j) {
      images = sp.p[j].images;
      items = sp.p[j].items;
      rows = sp.p[j].rows;
      counts = sp.p[j].table;
      nrows = sp.p[j].nrows;
    }
  const uint32_t block = blockIdx.x - sp.first[p];
  const uint32_t tile = block / (TB / 8), r8 = block % (TB / 8);
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

template <int KR, int KC, int TB, bool RUNS, bool NARROW>
void launch_form(hipStream_t s, const GemmOperands& op, const MfmaLaunch& l, uint32_t* counts, int flags) {
#ifdef OBNSL_PROBES
  hipLaunchKernelGGL((bitgemm_mfma_kernel<KR, KC, TB, RUNS, NARROW>), dim3(l.nworkgroups), dim3(2 * TB), 0, s, op.planes,
                     op.wp, op.rows, op.nrows, op.cols, op.ncols_total, l.units, l.unit_start, l.nruns, l.queue,
                     counts, flags, l.extra, l.trace);
#else
  hipLaunchKernelGGL((bitgemm_mfma_kernel<KR, KC, TB, RUNS, NARROW>), dim3(l.nworkgroups), dim3(2 * TB), 0, s, op.planes,
                     op.wp, op.rows, op.nrows, op.cols, op.ncols_total, l.units, l.unit_start, l.nruns, l.queue,
                     counts, flags, l.extra);
#endif
}
template <int KR, int KC, int TB>
void launch_one(hipStream_t s, const GemmOperands& op, const MfmaLaunch& l, uint32_t* counts, int flags) {
  if constexpr (TB == 128) {
    // (the persistent form exists for narrow data sets only: the host checks)
    if (l.unit_start != nullptr) {
      launch_form<KR, KC, TB, true, true>(s, op, l, counts, flags);
      return;
    }
    if (flags & GEMM_NARROW) {
      launch_form<KR, KC, TB, false, true>(s, op, l, counts, flags);
      return;
    }
  }
  // (256 x 256 tiles keep 64-bit addresses: with buffer loads, whose offsets they read from
  // LDS pass by pass, the 128-variable launch took 2762 ms against 2631)
  launch_form<KR, KC, TB, false, false>(s, op, l, counts, flags);
}

}  // namespace

bool gemm_mfma_has(uint32_t kr, uint32_t kc, uint32_t tile) {
  if (tile == 256) return kr == 2 || (kr == 3 && kc == 2);   // kc == 0 is launched as kc = 1
  return tile == 128;
}

void launch_gemm_mfma(uint32_t kr, uint32_t kc, uint32_t tile, hipStream_t s, const GemmOperands& op,
                      const MfmaLaunch& l, uint32_t* counts, int flags, int probe) {
#ifdef OBNSL_PROBES
  flags |= probe << 16;
#else
  (void)probe;
#endif
#define OBNSL_MFMA_CASE(KR, KC)                                                                        \
  case KR * 10 + KC:                                                                                   \
    if (tile == 256) {                                                                                 \
      if constexpr (KR == 2 || (KR == 3 && KC == 2))                                                   \
        launch_one<KR, KC, 256>(s, op, l, counts, flags);                                              \
    } else {                                                                                           \
      launch_one<KR, KC, 128>(s, op, l, counts, flags);                                                \
    }                                                                                                  \
    break
  switch (kr * 10 + kc) {
    OBNSL_MFMA_CASE(1, 1);
    OBNSL_MFMA_CASE(2, 1);
    OBNSL_MFMA_CASE(2, 2);
    OBNSL_MFMA_CASE(3, 2);
    OBNSL_MFMA_CASE(3, 3);
    OBNSL_MFMA_CASE(4, 3);
    OBNSL_MFMA_CASE(4, 4);
    default: break;
  }
#undef OBNSL_MFMA_CASE
}

void launch_image_scatter(uint32_t tile, hipStream_t s, const uint32_t* images, const uint2* items, uint32_t nitems,
                          const RowDesc* rows, uint32_t nrows, uint32_t* table) {
  if (tile == 256)
    hipLaunchKernelGGL(image_scatter_kernel<256>, dim3(nitems * (256 / 8)), dim3(256), 0, s, images, items, rows,
                       nrows, table);
  else
    hipLaunchKernelGGL(image_scatter_kernel<128>, dim3(nitems * (128 / 8)), dim3(256), 0, s, images, items, rows,
                       nrows, table);
}

void launch_image_scatter_parts(hipStream_t s, const ScatterParts& sp) {
  hipLaunchKernelGGL(image_scatter_parts_kernel, dim3(sp.first[sp.nparts]), dim3(256), 0, s, sp);
}

}  // namespace obnsl_internal
