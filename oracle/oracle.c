/*
 * oracle.c -- CPU restatement of the OpenBNSL local-score hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package links, loads or
 * calls this file.  It exists so that tests, the smoke check and the
 * cpu_baseline leg of the benchmark can compare the HIP engine with an
 * independent, plain-C statement of what the reference computes.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Parity status: the counting and BDeu functions are
 * pinned against the reference's own C++ (compiled from its sources into
 * oracle/_ref by oracle/Makefile) through tests/golden fixtures.  The
 * exhaustive-search DP is pinned only by the reference test's structural
 * criterion, because src/structure_learning/exhaustive_search.cpp cannot be
 * compiled here (it includes a Boost header that this image lacks); its optimum
 * is additionally checked against every DAG on 4-5 nodes in the tests.  BIC has
 * no reference implementation at all: PARITY UNPINNED for BIC.  Simulated
 * annealing is not restated here: its checker is the reference's own
 * simulated_annealing.cpp, compiled into oracle/_ref, and the graphs it returned
 * (tests/golden).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_KMAX 16

enum { ORACLE_BDEU = 0, ORACLE_K2 = 1, ORACLE_BIC = 2, ORACLE_AIC = 3 };

/* ---- include/utils/next_combmask.h:36-47 (Gosper's hack) ---------------- */
int oracle_next_combmask(uint64_t* comb, int n) {
  uint64_t c = *comb;
  if (c == 0) return 0;
  uint64_t x = c & (~c + 1);
  uint64_t y = c + x;
  c = (((c & ~y) / x) >> 1) | y;
  *comb = c;
  if (n >= 64) return c != 0 && y != 0; /* 1<<64 is not representable */
  return c < ((uint64_t)1 << n);
}

/* ---- include/utils/combmask2vec.h:11-26 --------------------------------- */
int oracle_combmask2vec(uint64_t mask, uint32_t* out) {
  int k = 0;
  while (mask) {
    out[k++] = (uint32_t)__builtin_ctzll(mask);
    mask &= mask - 1;
  }
  return k;
}

/* Number of k-subsets visited by the do/while idiom of
 * gtest/utils/next_combination.cpp:9-16. */
uint64_t oracle_comb_count(int n, int m) {
  uint64_t comb = m >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << m) - 1);
  uint64_t cnt = 0;
  do {
    cnt++;
  } while (oracle_next_combmask(&comb, n));
  return cnt;
}

/* ---- include/base/contingency_table.h:34-56, 77-85 ----------------------
 * Joint histogram of the (ascending) variable subset `vars`.  The reference
 * keeps a sparse std::map key->count; this restatement keeps the same keys in
 * a dense array of prod(card) cells, so "cells present in the map" == cells
 * with a non-zero count and "ascending key order" == ascending index.
 * data is column-major: code of variable v in row i is data[v*stride + i].
 * Returns 0, or -1 if vars is not sorted (contingency_table.h:37-38), -2 if
 * the dense table would not fit ncells_capacity. */
int oracle_count(const uint8_t* data, uint64_t N, uint64_t stride,
                 const uint32_t* card, const uint32_t* vars, int k,
                 uint64_t* counts, uint64_t ncells_capacity) {
  uint64_t w[ORACLE_KMAX];
  if (k > ORACLE_KMAX) return -2;
  for (int i = 1; i < k; i++)
    if (vars[i] < vars[i - 1]) return -1;
  uint64_t mult = 1;
  for (int i = k - 1; i >= 0; i--) { /* last variable has weight 1 (:43-49) */
    w[i] = mult;
    mult *= card[vars[i]];
  }
  if (mult > ncells_capacity) return -2;
  memset(counts, 0, mult * sizeof(uint64_t));
  for (uint64_t r = 0; r < N; r++) { /* :52-55 */
    uint64_t key = 0;
    for (int i = 0; i < k; i++) key += (uint64_t)data[vars[i] * stride + r] * w[i];
    counts[key]++;
  }
  return 0;
}

/* ---- include/score/local_score.h:22-62 ----------------------------------
 * BDeu exactly as written there: a_ijk = ess / prod(card) (:33-41), a_ij =
 * a_ijk * r_child (:45); first sum over non-zero cells in ascending key order
 * (:51-56), then the parent-configuration marginal, obtained by zeroing the
 * child digit of the key (contingency_table.h:68-70), in ascending stripped
 * key order (:57-60).  Each bracket is evaluated, then accumulated.
 * BIC (no reference code; PARITY UNPINNED): sum N_ijk ln(N_ijk/N_ij)
 * - 0.5 ln(N) q (r-1) with q = full product of the parent cardinalities.
 * K2/AIC: returns NaN (the reference throws "Unsupported score type.",
 * local_score.h:42-44; callers map NaN to that error).
 * `marg` is caller scratch of `cells` entries. */
double oracle_local_score(const uint64_t* counts, const uint32_t* cards, int k,
                          int child_idx, int kind, double ess, uint64_t* marg) {
  uint64_t cells = 1, w_child = 1;
  for (int i = 0; i < k; i++) cells *= cards[i];
  for (int i = k - 1; i > child_idx; i--) w_child *= cards[i];
  const uint64_t r = cards[child_idx];
  memset(marg, 0, cells * sizeof(uint64_t));
  if (kind == ORACLE_BDEU) {
    const double a_ijk = ess / (double)cells;
    const double a_ij = a_ijk * (double)r;
    const double lg_a_ijk = lgamma(a_ijk);
    const double lg_a_ij = lgamma(a_ij);
    double ls = 0;
    for (uint64_t key = 0; key < cells; key++) {
      const uint64_t n_ijk = counts[key];
      if (n_ijk == 0) continue;
      ls += lgamma((double)n_ijk + a_ijk) - lg_a_ijk;
      const uint64_t state = (key / w_child) % r;
      marg[key - state * w_child] += n_ijk;
    }
    for (uint64_t key = 0; key < cells; key++) {
      const uint64_t n_ij = marg[key];
      if (n_ij == 0) continue;
      ls -= lgamma((double)n_ij + a_ij) - lg_a_ij;
    }
    return ls;
  }
  if (kind == ORACLE_BIC) {
    uint64_t total = 0;
    double ll = 0;
    for (uint64_t key = 0; key < cells; key++) {
      const uint64_t n_ijk = counts[key];
      if (n_ijk == 0) continue;
      total += n_ijk;
      ll += (double)n_ijk * log((double)n_ijk);
      const uint64_t state = (key / w_child) % r;
      marg[key - state * w_child] += n_ijk;
    }
    for (uint64_t key = 0; key < cells; key++) {
      const uint64_t n_ij = marg[key];
      if (n_ij == 0) continue;
      ll -= (double)n_ij * log((double)n_ij);
    }
    const double q = (double)(cells / r);
    return ll - 0.5 * log((double)total) * q * (double)(r - 1);
  }
  return NAN;
}

/* Colex successor on an index tuple: the same visiting order as Gosper's hack
 * on bitmasks (increasing mask value), but valid for n > 64 as well. */
static int next_colex(uint32_t* c, int k, uint32_t n) {
  for (int i = 0; i < k; i++) {
    const uint32_t lim = (i + 1 < k) ? c[i + 1] : n;
    if (c[i] + 1 < lim) {
      c[i]++;
      for (int j = 0; j < i; j++) c[j] = (uint32_t)j;
      return 1;
    }
  }
  return 0;
}

uint64_t oracle_binom(uint32_t n, uint32_t k) {
  if (k > n) return 0;
  if (k > n - k) k = n - k;
  uint64_t r = 1;
  for (uint32_t i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}

/* Subsets of size k in the visiting order of
 * src/structure_learning/exhaustive_search.cpp:51-52,129 (Gosper order).
 * Writes C(n,k)*k variable ids. */
void oracle_enumerate(uint32_t n, int k, uint32_t* out) {
  uint32_t c[ORACLE_KMAX];
  for (int i = 0; i < k; i++) c[i] = (uint32_t)i;
  uint64_t t = 0;
  if (k == 0 || (uint32_t)k > n) return;
  do {
    memcpy(out + t * k, c, k * sizeof(uint32_t));
    t++;
  } while (next_colex(c, k, n));
}

/* Scoring phase of exhaustive_search.cpp:47-73 for one subset size: for every
 * size-k subset S in visiting order, one table, then one score per member of S
 * taken as the child (members ascending).  out[t*k + i].  nthreads <= 1 runs
 * the serial loop the reference runs; nthreads > 1 spreads subsets over OpenMP
 * threads (the reference has no such pragma; used as the "all cores" figure). */
int oracle_score_all(const uint8_t* data, uint64_t N, uint64_t stride, uint32_t n,
                     const uint32_t* card, int k, int kind, double ess,
                     double* out, int nthreads) {
  if (k <= 0 || (uint32_t)k > n || k > ORACLE_KMAX) return -1;
  const uint64_t T = oracle_binom(n, (uint32_t)k);
  uint32_t* subsets = (uint32_t*)malloc(T * k * sizeof(uint32_t));
  if (!subsets) return -3;
  oracle_enumerate(n, k, subsets);
  int rc = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint64_t cap = 0;
    uint64_t *counts = NULL, *marg = NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int64_t t = 0; t < (int64_t)T; t++) {
      const uint32_t* vars = subsets + (uint64_t)t * k;
      uint32_t cards[ORACLE_KMAX];
      uint64_t cells = 1;
      for (int i = 0; i < k; i++) {
        cards[i] = card[vars[i]];
        cells *= cards[i];
      }
      if (cells > cap) {
        free(counts);
        free(marg);
        cap = cells;
        counts = (uint64_t*)malloc(cap * sizeof(uint64_t));
        marg = (uint64_t*)malloc(cap * sizeof(uint64_t));
      }
      if (!counts || !marg || oracle_count(data, N, stride, card, vars, k, counts, cap)) {
        rc = -3;
        continue;
      }
      for (int i = 0; i < k; i++)
        out[(uint64_t)t * k + i] = oracle_local_score(counts, cards, k, i, kind, ess, marg);
    }
    free(counts);
    free(marg);
  }
  free(subsets);
  return rc;
}

/* ---- src/structure_learning/exhaustive_search.cpp:14-165 ----------------
 * The whole search, structured as the reference structures it: one loop over
 * subset sizes and Gosper-ordered masks, scoring and DP bookkeeping interleaved.
 * parents_out[v] = bitmask of the parents of v in the returned graph.
 * Returns 0; -1 for max_parents > n-1 (:18-19); -2 for n > 64 (:22-23) and for
 * n too large for the n*2^n tables to be addressable here (n > 30). */
/* `tables` != NULL: the local scores are read from tables[k-1][idx * k + ci]
 * (idx = position of the subset among the size-k subsets in the visiting order
 * of :51-52,129, i.e. its colex rank) instead of being computed from the data;
 * everything else -- visiting order, tie rules, back-track -- is the same loop. */
static int search_impl(const uint8_t* data, uint64_t N, uint64_t stride,
                       uint32_t n, const uint32_t* card, int kind, double ess,
                       uint32_t max_parents, const double* const* tables,
                       uint64_t* parents_out, double* total_score_out) {
  if (n == 0 || max_parents > n - 1) return -1;
  if (n > 30) return -2;
  const uint64_t combs = (uint64_t)1 << n;
  const uint32_t max_varset = max_parents + 1;
  double* best_gs = (double*)calloc(combs, sizeof(double));
  uint8_t* best_ch = (uint8_t*)calloc(combs, 1);
  double* bls_score = (double*)malloc(sizeof(double) * n * combs);
  uint64_t* bls_ps = (uint64_t*)malloc(sizeof(uint64_t) * n * combs);
  uint64_t maxcells = 1;
  if (!tables) { /* upper bound on table size: product of the max_varset largest cards */
    uint32_t tmp[64];
    memcpy(tmp, card, n * sizeof(uint32_t));
    for (uint32_t i = 0; i < max_varset && i < n; i++) {
      uint32_t bi = i;
      for (uint32_t j = i + 1; j < n; j++)
        if (tmp[j] > tmp[bi]) bi = j;
      uint32_t t = tmp[i];
      tmp[i] = tmp[bi];
      tmp[bi] = t;
      maxcells *= tmp[i];
    }
  }
  uint64_t* counts = (uint64_t*)malloc(maxcells * sizeof(uint64_t));
  uint64_t* marg = (uint64_t*)malloc(maxcells * sizeof(uint64_t));
  if (!best_gs || !best_ch || !bls_score || !bls_ps || !counts || !marg) return -3;
  for (uint64_t i = 0; i < (uint64_t)n * combs; i++) {
    bls_score[i] = 0.0; /* value-initialised pairs, :37-40 */
    bls_ps[i] = 0;
  }
  best_gs[0] = 0.0; /* :42-44 */

  for (uint32_t size = 0; size <= n; size++) { /* :47 */
    const int calc_ls = size <= max_varset; /* :48 */
    uint64_t varset = size >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << size) - 1);
    uint64_t idx = 0; /* position in the visiting order of this size */
    do {
      uint32_t vec[64], cards[64];
      const int k = oracle_combmask2vec(varset, vec);
      if (calc_ls && !tables) { /* :55-57; the empty set gives the one-cell table {0:N} */
        oracle_count(data, N, stride, card, vec, k, counts, maxcells);
        for (int i = 0; i < k; i++) cards[i] = card[vec[i]];
      }
      double bgs = 0;
      int bch = -1;
      for (int ci = 0; ci < k; ci++) { /* :61 */
        const uint32_t child = vec[ci];
        const uint64_t pmask = varset - ((uint64_t)1 << child);
        double ls = -INFINITY; /* :65 */
        if (calc_ls && tables) ls = tables[k - 1][idx * (uint64_t)k + (uint64_t)ci];
        else if (calc_ls) ls = oracle_local_score(counts, cards, k, ci, kind, ess, marg);
        double best_ls = ls; /* :75-87 */
        uint64_t best_ps = pmask;
        for (int pi = 0; pi < k; pi++) {
          if (pi == ci) continue;
          const uint64_t sub = pmask - ((uint64_t)1 << vec[pi]);
          const double l = bls_score[child * combs + sub];
          if (l >= best_ls) {
            best_ls = l;
            best_ps = bls_ps[child * combs + sub];
          }
        }
        bls_score[child * combs + pmask] = best_ls;
        bls_ps[child * combs + pmask] = best_ps;
        const double gs = best_gs[pmask] + best_ls; /* :89-94 */
        if (bgs >= 0 || gs > bgs) {
          bgs = gs;
          bch = (int)child;
        }
      }
      best_gs[varset] = bgs; /* :115-116 */
      best_ch[varset] = (uint8_t)bch;
      idx++;
    } while (oracle_next_combmask(&varset, (int)n));
  }

  for (uint32_t v = 0; v < n; v++) parents_out[v] = 0;
  uint64_t varset = combs - 1; /* :132-148 */
  if (total_score_out) *total_score_out = best_gs[varset];
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t ch = best_ch[varset];
    const uint64_t pmask = varset - ((uint64_t)1 << ch);
    parents_out[ch] |= bls_ps[ch * combs + pmask];
    varset = pmask;
  }
  free(best_gs);
  free(best_ch);
  free(bls_score);
  free(bls_ps);
  free(counts);
  free(marg);
  return 0;
}

int oracle_exhaustive_search(const uint8_t* data, uint64_t N, uint64_t stride,
                             uint32_t n, const uint32_t* card, int kind, double ess,
                             uint32_t max_parents, uint64_t* parents_out,
                             double* total_score_out) {
  return search_impl(data, N, stride, n, card, kind, ess, max_parents, NULL, parents_out,
                     total_score_out);
}

/* The dynamic program and back-track of exhaustive_search.cpp:75-148 on given
 * local scores (test input: tie-heavy tables pin the tie rules of :82 and :90-94). */
int oracle_exhaustive_search_tables(uint32_t n, uint32_t max_parents, const double* const* tables,
                                    uint64_t* parents_out, double* total_score_out) {
  return search_impl(NULL, 0, 0, n, NULL, 0, 0.0, max_parents, tables, parents_out,
                     total_score_out);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
