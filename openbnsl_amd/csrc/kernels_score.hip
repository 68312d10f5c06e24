// kernels_score.hip -- fp64 local scores from count tables.
//
// The reference evaluates BDeu from one sparse table
// (include/score/local_score.h:38-61).  score_table() is that loop over a dense
// table by one wavefront; score_reduced_kernel expands the contraction's reduced
// count tables to full tables in LDS (inclusion-exclusion) and evaluates one
// functional per table, from which the k scores of a subset are differences.
#include <hip/hip_runtime.h>

#include <cmath>

#include "engine_state.h"

namespace obnsl_internal {
namespace {

// ------------------------------------------------------------------ scoring
// 1 / x for a normal x > 0: the hardware estimate and two Newton steps (far below one
// unit in the last place of what it feeds: a series term and a quotient that is
// corrected once more).
__device__ __forceinline__ double rcp_pos(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ln(x) for a normal x > 0 (counts plus a prior, products of a few of them): x = 2^e m with
// m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s) with s = (m - 1) / (m + 1), |s| <= 0.1716, as
// the odd series through s^19 (the next term is below 2.3e-17 of the sum).  About a third of
// the instructions of the library routine, which also serves zero, negative, infinite and
// subnormal arguments; the scorer's time is mostly this function.
__device__ __forceinline__ double log_pos(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double num = m - 1.0, den = m + 1.0;
  const double r = rcp_pos(den);
  double s = num * r;
  s = fma(fma(-s, den, num), r, s);            // the quotient, corrected by its own residual
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p *= z;
  const double two_s = s + s;
  const double lm = fma(two_s, p, two_s);
  const double ef = (double)e;
  // e ln 2 with ln 2 split so that the first product is exact
  return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, lm));
}

// lgamma(x) for x >= 16 given lx = ln(x): the Stirling series through the 1/x^11 term (the
// next term is < 2e-18 at x = 16).
__device__ __forceinline__ double lgamma_stirling(double x, double lx) {
  const double inv = rcp_pos(x), inv2 = inv * inv;
  const double ser =
      inv * (1.0 / 12.0 -
             inv2 * (1.0 / 360.0 -
                     inv2 * (1.0 / 1260.0 -
                             inv2 * (1.0 / 1680.0 - inv2 * (1.0 / 1188.0 - inv2 * (691.0 / 360360.0))))));
  return (x - 0.5) * lx - x + 0.91893853320467274178 + ser;
}

// lgamma(x) for x > 0: shift the argument up to >= 16 with the recurrence
// lgamma(x) = lgamma(x + m) - ln(x (x+1) ... (x+m-1)), then the Stirling series.  Agrees
// with glibc's lgamma to ~2e-14 absolute over the arguments the scores use (counts plus a
// positive prior); far smaller and lighter on registers than the general-purpose library
// routine, which matters for the scorer's occupancy.
__device__ __forceinline__ double lgamma_pos(double x) {
  double p = 1.0;
  const bool shifted = x < 16.0;
  while (x < 16.0) {
    p *= x;
    x += 1.0;
  }
  double r = lgamma_stirling(x, log_pos(x));
  if (shifted) r -= log_pos(p);
  return r;
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
      t1 += (double)nijk * log_pos((double)nijk);
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
      else t2 += (double)nij * log_pos((double)nij);
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
// Shared memory of a scorer workgroup besides the tables E (the caller declares it once,
// whatever subset size the workgroup turns out to serve).
template <int KCAP>
struct ScorerShared {
  SizeView sv[KFAST + 1];
  uint32_t vars[4][KFAST], cards[4][KFAST];
  uint64_t pat_base[4][1 << KCAP];   // first counter of the reduced table of a support pattern
};

template <int KT, int PARTS, int KCAP, bool READ_RECORDS = false>
__device__ __forceinline__ void reduced_body(
    uint32_t block, uint32_t* __restrict__ E_all, ScorerShared<KCAP>& sh, const ReducedViews& views,
    const RowGroupDesc* __restrict__ rgs, uint64_t ntables, uint32_t tables_per_wg,
    const uint32_t* __restrict__ card, const uint64_t* __restrict__ binom, uint32_t num_rows, int kind,
    double ess, double* __restrict__ G, const double* __restrict__ Gprev, double g0,
    double* __restrict__ scores, const double* __restrict__ lg_alpha, uint32_t e_stride,
    const uint32_t* __restrict__ wg_rg, uint32_t* __restrict__ recs = nullptr, int rec_mode = 0) {
  // The wavefront works on PARTS tables at a time (2 or 4: 32 or 16 lanes each):
  // a table is a chain of dependent loads (descriptors -> table bases -> counts
  // -> G of the subsets below), and several chains in flight per wave hide more
  // of that latency than one; four pay when the tables are small (few cells per
  // lane either way).  All parts pass the same barriers; a part without a table
  // idles through them.
  static_assert(KT <= KCAP && PARTS <= 4, "shared arrays are sized for KCAP variables and four parts");
  constexpr uint32_t SC_PARTS = PARTS, SC_LANES = 64 / PARTS;
  SizeView* sv = sh.sv;
  uint32_t (*s_vars)[KFAST] = sh.vars;
  uint32_t (*s_cards)[KFAST] = sh.cards;
  constexpr uint32_t k = KT;
  const uint32_t half = threadIdx.x / SC_LANES, lane = threadIdx.x % SC_LANES;
  uint32_t* E = E_all + (size_t)half * e_stride;
  uint64_t* pat_base = sh.pat_base[half];
#pragma unroll
  for (int j = 0; j <= KFAST; ++j)
    if (threadIdx.x == (uint32_t)j) sv[j] = views.v[j];
  const uint64_t t0 = (uint64_t)block * tables_per_wg;
  const uint64_t t1 = min(t0 + (uint64_t)tables_per_wg, ntables);
  // What a table needs before its counts can be gathered -- its rank, its variables'
  // cardinalities and where the reduced table of each support pattern starts -- takes a
  // chain of five dependent loads to work out from the descriptors, and is the same in
  // every pass.  rec_mode 1 leaves it behind as a record per table (REC_WORDS words:
  // rank, cardinalities - 1 as bytes, one start per pattern); rec_mode 2 reads that
  // record instead of the descriptors: one load in place of the chain.
  constexpr uint32_t REC_WORDS = table_record_words(KT);
  constexpr bool from_records = READ_RECORDS;
  // row group of the workgroup's first table (from the host: one load instead of a
  // binary search of dependent loads); every part advances from there through the groups
  uint32_t g = 0;
  RowGroupDesc rg;
  rg.rank_part = 0;
  rg.tbl_start = 0;
  rg.ncolgroups = 0;
  if (!from_records) {
    g = wg_rg[block];
    rg = rgs[g];
  }
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
    uint64_t rank = 0;
    if (active && from_records) {
      const uint32_t* __restrict__ rec = recs + t * REC_WORDS;
      rank = (uint64_t)rec[0] | ((uint64_t)rec[1] << 32);
      const uint32_t c_lo = rec[2], c_hi = rec[3];
#pragma unroll
      for (int i = 0; i < KFAST; ++i) {
        if (i < (int)k) {
          cards[i] = (((i < 4 ? c_lo : c_hi) >> (8 * (i & 3))) & 0xffu) + 1u;
          inv_card[i] = 1.0f / (float)cards[i];
          cells *= cards[i];
        }
      }
      for (uint32_t pat = lane + 1; pat < (1u << k); pat += SC_LANES) pat_base[pat] = rec[3 + pat];
    }
    if (active && !from_records) {
      while (t >= rg.tbl_start + rg.ncolgroups) rg = rgs[++g];
      gamma = (uint32_t)(t - rg.tbl_start);
      rank = rg.rank_part + gamma;
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
        const uint64_t start = s.rowoff[rowrank] + (uint64_t)col_before * mred;
        pat_base[pat] = start;
        if (rec_mode == 1) recs[t * REC_WORDS + 3 + pat] = (uint32_t)start;
      }
      if (rec_mode == 1 && lane == 0) {
        uint32_t c_lo = 0, c_hi = 0;
#pragma unroll
        for (int i = 0; i < KFAST; ++i) {
          if (i >= (int)k) continue;
          if (i < 4) c_lo |= ((cards[i] - 1u) & 0xffu) << (8 * i);
          else c_hi |= ((cards[i] - 1u) & 0xffu) << (8 * (i - 4));
        }
        uint32_t* rec = recs + t * REC_WORDS;
        rec[0] = (uint32_t)rank;
        rec[1] = (uint32_t)(rank >> 32);
        rec[2] = c_lo;
        rec[3] = c_hi;
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
    double gsum = 0.0;
    if (active) {
      if (kind == OBNSL_SCORE_BDEU) {
        const double a = ess / (double)cells;
        const double lg_a = lg_alpha[cells];  // lgamma(a): the same for every table of this size
        // lgamma(n + a) - lgamma(a) for n < LG_SMALL (0 for an empty cell): one load in
        // place of the shifted evaluation; larger counts need no shift
        const double* __restrict__ small = lg_alpha + (REDUCED_MAX_CELLS + 1) + (size_t)cells * LG_SMALL;
        for (uint32_t c = lane; c < cells; c += SC_LANES) {
          const uint32_t nc = E[c];
          if (nc < (uint32_t)LG_SMALL) {
            gsum += small[nc];
          } else {
            const double x = (double)nc + a;
            gsum += lgamma_stirling(x, log_pos(x)) - lg_a;
          }
        }
      } else {
        for (uint32_t c = lane; c < cells; c += SC_LANES) {
          const uint32_t nc = E[c];
          if (nc) gsum += (double)nc * log_pos((double)nc);
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

// One subset size per launch: G of its tables and, when `scores` is given, the k scores
// of each (Gprev = the functional of the next smaller size, computed by an earlier launch).
template <int KT, int PARTS>
__global__ __launch_bounds__(64) void score_reduced_kernel(
    ReducedViews views, const RowGroupDesc* __restrict__ rgs, uint32_t nrgs,
    uint64_t ntables, uint32_t tables_per_wg, const uint32_t* __restrict__ card,
    const uint64_t* __restrict__ binom, uint32_t num_rows, int kind, double ess,
    double* __restrict__ G, const double* __restrict__ Gprev, double g0,
    double* __restrict__ scores, const double* __restrict__ lg_alpha, uint32_t e_stride,
    const uint32_t* __restrict__ wg_rg) {
  extern __shared__ __attribute__((aligned(16))) uint32_t E_all[];
  __shared__ ScorerShared<KT> sh;
  reduced_body<KT, PARTS, KT>(blockIdx.x, E_all, sh, views, rgs, ntables, tables_per_wg, card, binom, num_rows, kind,
                              ess, G, Gprev, g0, scores, lg_alpha, e_stride, wg_rg);
}

// Every subset size 1..kmax (<= FUSED_KMAX) in one launch: the functional G of every
// table of every size.  Workgroups [block_start[k], block_start[k + 1]) serve size k; the
// scores follow from G in scores_best_kernel.  Workgroup 0 also resets the per-node
// arg-max words (keys 0, ids all ones) that scores_best_kernel and best_parents_kernel
// reduce into.
template <bool READ_RECORDS>
__global__ __launch_bounds__(64) void functional_all_kernel(FunctionalAllArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t E_all[];
  __shared__ ScorerShared<FUSED_KMAX> sh;
  if (blockIdx.x == 0 && a.best_key != nullptr) {
    for (uint32_t v = threadIdx.x; v < a.n; v += 64) {
      a.best_key[v] = 0ull;
      a.best_key[a.n + v] = ~0ull;
    }
  }
  uint32_t k = 1;
#pragma unroll
  for (int j = 2; j <= FUSED_KMAX; ++j)
    if ((uint32_t)j <= a.kmax && blockIdx.x >= a.size[j].block_start) k = j;
#define OBNSL_FUNCTIONAL_CASE(KT)                                                                            \
  case KT: {                                                                                                  \
    const FunctionalSizeArgs& z = a.size[KT];                                                                 \
    const uint32_t block = blockIdx.x - z.block_start;                                                        \
    if (z.parts == 4)                                                                                         \
      reduced_body<KT, 4, FUSED_KMAX, READ_RECORDS>(block, E_all, sh, a.views, z.rgs, z.ntables, z.tables_per_wg, a.card, \
                                      a.binom, a.num_rows, a.kind, a.ess, z.G, nullptr, 0.0, nullptr,         \
                                      a.lg_alpha, z.e_stride, z.wg_rg, z.recs, z.rec_mode);                   \
    else                                                                                                      \
      reduced_body<KT, 2, FUSED_KMAX, READ_RECORDS>(block, E_all, sh, a.views, z.rgs, z.ntables, z.tables_per_wg, a.card, \
                                      a.binom, a.num_rows, a.kind, a.ess, z.G, nullptr, 0.0, nullptr,         \
                                      a.lg_alpha, z.e_stride, z.wg_rg, z.recs, z.rec_mode);                   \
  } break
  switch (k) {   // uniform over the workgroup
    OBNSL_FUNCTIONAL_CASE(1);
    OBNSL_FUNCTIONAL_CASE(2);
    OBNSL_FUNCTIONAL_CASE(3);
    OBNSL_FUNCTIONAL_CASE(4);
    OBNSL_FUNCTIONAL_CASE(5);
    default: break;
  }
#undef OBNSL_FUNCTIONAL_CASE
}

// Largest v with C(v, j) <= rank; binom is [(n+1)][KMAX+1].
__device__ __forceinline__ uint32_t unrank_step(const uint64_t* __restrict__ binom, uint32_t n, uint32_t j,
                                                uint64_t rank) {
  uint32_t lo = j - 1, hi = n - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (binom[mid * (KMAX + 1) + j] <= rank) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ uint64_t ordered_key(double x) {
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Scores of every size from the functionals, and the first pass of the per-node arg-max:
// one thread per table t (colex rank) of size k,
//   score(v_i | S \ v_i) = G_k[t] - G_{k-1}[rank(S \ v_i)]   (BIC: minus its penalty)
// written to scores[k][t * k + i], and the largest ordered score per node kept in
// best_key (reduced per workgroup in LDS first).  With a mask, only the tables it marks
// are scored (the others belong to other shards: their entries stay untouched).
__global__ __launch_bounds__(256) void scores_best_kernel(ScoresAllArgs a) {
  __shared__ unsigned long long s_best[BEST_LDS_VARS];
  uint32_t k = 1;
#pragma unroll
  for (int j = 2; j <= FUSED_KMAX; ++j)
    if ((uint32_t)j <= a.kmax && blockIdx.x >= a.block_start[j]) k = j;
  const double* __restrict__ Gk = nullptr;
  const double* __restrict__ Gp = nullptr;
  const uint8_t* __restrict__ mask = nullptr;
  const uint16_t* __restrict__ vtab = nullptr;
  double* __restrict__ scores = nullptr;
  uint64_t nsub = 0;
  uint32_t first_block = 0, nblocks = 1;
#pragma unroll
  for (int j = 1; j <= FUSED_KMAX; ++j)
    if ((uint32_t)j == k) {
      Gk = a.G[j];
      Gp = a.G[j - 1];
      scores = a.scores[j];
      mask = a.mask[j];
      vtab = a.vars[j];
      nsub = a.nsubsets[j];
      first_block = a.block_start[j];
      nblocks = a.block_start[j + 1] - a.block_start[j];
    }
  const bool use_lds = a.best_key != nullptr && a.n <= BEST_LDS_VARS;
  if (use_lds) {
    for (uint32_t v = threadIdx.x; v < a.n; v += 256) s_best[v] = 0ull;
    __syncthreads();
  }
  const double half_log_n = 0.5 * log((double)a.num_rows);
  for (uint64_t t = (uint64_t)(blockIdx.x - first_block) * 256 + threadIdx.x; t < nsub; t += (uint64_t)nblocks * 256) {
    if (mask != nullptr && !mask[t]) continue;   // another shard's table
    const double g = Gk[t];
    uint32_t vars[FUSED_KMAX];
    uint64_t rem = t;
    uint32_t cells = 1;
#pragma unroll
    for (int j = FUSED_KMAX; j >= 1; --j) {
      vars[j - 1] = 0;
      if ((uint32_t)j > k) continue;
      uint32_t v;
      if (vtab != nullptr) {
        v = vtab[t * k + (uint32_t)(j - 1)];
      } else {
        v = unrank_step(a.binom, a.n, (uint32_t)j, rem);
        rem -= a.binom[v * (KMAX + 1) + j];
      }
      vars[j - 1] = v;
      if (a.kind == OBNSL_SCORE_BIC) cells *= a.card[v];
    }
#pragma unroll
    for (int i = 0; i < FUSED_KMAX; ++i) {
      if ((uint32_t)i >= k) continue;
      double gp = a.g0;
      if (k > 1) {
        uint64_t sub = 0;
        uint32_t pos = 1;
#pragma unroll
        for (int j = 0; j < FUSED_KMAX; ++j) {
          if ((uint32_t)j >= k || j == i) continue;
          sub += a.binom[vars[j] * (KMAX + 1) + pos];
          ++pos;
        }
        gp = Gp[sub];
      }
      double sc = g - gp;
      if (a.kind == OBNSL_SCORE_BIC) {
        const uint32_t r = a.card[vars[i]];
        sc -= half_log_n * (double)(cells / r) * (double)(r - 1);
      }
      scores[t * k + i] = sc;
      if (a.best_key != nullptr && sc > -INFINITY) {
        const unsigned long long key = ordered_key(sc);
        if (use_lds) atomicMax(&s_best[vars[i]], key);
        else atomicMax(&a.best_key[vars[i]], key);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    // (reading the word first and skipping the atomic where it cannot raise it was measured
    // slower: 12.2 us against 9.6 for the 37-variable pass)
    for (uint32_t v = threadIdx.x; v < a.n; v += 256) {
      const unsigned long long x = s_best[v];
      if (x) atomicMax(&a.best_key[v], x);
    }
  }
}

template <int KT>
void launch_reduced_k(hipStream_t s, uint32_t parts, uint32_t grid, const ReducedScorerArgs& a) {
  if (parts == 4)
    hipLaunchKernelGGL((score_reduced_kernel<KT, 4>), dim3(grid), dim3(64), (size_t)4 * a.e_stride * sizeof(uint32_t),
                       s, a.views, a.rgs, a.nrgs, a.ntables, a.tables_per_wg, a.card, a.binom, a.num_rows, a.kind,
                       a.ess, a.G, a.Gprev, a.g0, a.scores, a.lg_alpha, a.e_stride, a.wg_rg);
  else
    hipLaunchKernelGGL((score_reduced_kernel<KT, 2>), dim3(grid), dim3(64), (size_t)2 * a.e_stride * sizeof(uint32_t),
                       s, a.views, a.rgs, a.nrgs, a.ntables, a.tables_per_wg, a.card, a.binom, a.num_rows, a.kind,
                       a.ess, a.G, a.Gprev, a.g0, a.scores, a.lg_alpha, a.e_stride, a.wg_rg);
}

}  // namespace

void launch_score_plan(hipStream_t s, uint32_t grid, const uint32_t* counts, const RowGroupDesc* rgs, uint32_t nrgs,
                       const ColGroupDesc* cgs, const uint32_t* card, uint32_t kr, uint32_t kc, uint64_t ntables,
                       int kind, double ess, double* scores) {
  hipLaunchKernelGGL(score_plan_kernel, dim3(grid), dim3(256), 0, s, counts, rgs, nrgs, cgs, card, kr, kc, ntables,
                     kind, ess, scores);
}

void launch_score_tables(hipStream_t s, const uint32_t* counts, const TableDesc* tables, uint64_t ntables, int kind,
                         double ess, double* scores) {
  hipLaunchKernelGGL(score_tables_kernel, dim3((uint32_t)((ntables + 3) / 4)), dim3(256), 0, s, counts, tables,
                     ntables, kind, ess, scores);
}

void launch_score_one(hipStream_t s, const uint32_t* counts, const TableDesc& td, int kind, double ess,
                      double* scores) {
  hipLaunchKernelGGL(score_one_kernel, dim3(1), dim3(64), 0, s, counts, td, kind, ess, scores);
}

void launch_functional_all(hipStream_t s, uint32_t grid, size_t lds_bytes, const FunctionalAllArgs& a) {
  // every size's tables from their records, or every size's from the descriptors
  bool all_read = true;
  for (uint32_t k = 1; k <= a.kmax; ++k) all_read = all_read && (a.size[k].rec_mode == 2 || a.size[k].ntables == 0);
  if (all_read)
    hipLaunchKernelGGL(functional_all_kernel<true>, dim3(grid), dim3(64), lds_bytes, s, a);
  else
    hipLaunchKernelGGL(functional_all_kernel<false>, dim3(grid), dim3(64), lds_bytes, s, a);
}

void launch_scores_best(hipStream_t s, const ScoresAllArgs& a) {
  hipLaunchKernelGGL(scores_best_kernel, dim3(a.block_start[a.kmax + 1]), dim3(256), 0, s, a);
}

void launch_score_reduced(hipStream_t s, uint32_t k, uint32_t parts, uint32_t grid, const ReducedScorerArgs& a) {
  switch (k) {
    case 1: launch_reduced_k<1>(s, parts, grid, a); break;
    case 2: launch_reduced_k<2>(s, parts, grid, a); break;
    case 3: launch_reduced_k<3>(s, parts, grid, a); break;
    case 4: launch_reduced_k<4>(s, parts, grid, a); break;
    case 5: launch_reduced_k<5>(s, parts, grid, a); break;
    case 6: launch_reduced_k<6>(s, parts, grid, a); break;
    case 7: launch_reduced_k<7>(s, parts, grid, a); break;
    default: launch_reduced_k<8>(s, parts, grid, a); break;
  }
}

}  // namespace obnsl_internal
