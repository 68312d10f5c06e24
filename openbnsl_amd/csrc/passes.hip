// passes.hip -- the passes of a data set over its streams: which counting path serves a
// subset size (the bit-plane contraction or the per-row histogram), the pack, the scoring
// of one size (obnsl_score_all) or of every size at once (obnsl_score_pass), the count of
// every size ahead of the scorers (obnsl_precount, sharded over the subsets or over the
// rows) and the per-node arg-max.  What it replaces is the scoring loop of
// exhaustive_search (src/structure_learning/exhaustive_search.cpp:47-73): one table and
// k scores per subset, subset after subset.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine_state.h"

namespace obnsl_internal {

// Launch shape of the reduced-mode scorer over the tables of `plan` (subset size k).
struct ScorerShape {
  uint32_t tpw = 0, parts = 2, grid = 0, e_stride = 0;
};
static obnsl_status scorer_shape(obnsl_data* d, Plan* plan, uint32_t k, uint64_t max_cells, ScorerShape* out) {
  Batch* b = plan->batches[0].get();
  // a few consecutive tables per wavefront amortise the descriptor loads and
  // the dispatch; keep at least ~16 workgroups per CU for latency hiding
  const uint32_t tpw_env = env().tables_per_wg;
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
  out->tpw = tpw;
  out->parts = parts;
  out->grid = grid;
  out->e_stride = (uint32_t)((max_cells + 3) & ~3ull);
  return OBNSL_OK;
}

// Per ess, on the device (from the host's libm): lgamma(ess / c) for c = 1..REDUCED_MAX_CELLS
// at index c, and behind those, at (REDUCED_MAX_CELLS + 1) + c * LG_SMALL + n, the difference
// lgamma(n + ess / c) - lgamma(ess / c) = sum_{i < n} ln(ess / c + i) for n < LG_SMALL.
static obnsl_status lg_alpha_table(obnsl_data* d, obnsl_score_kind kind, double ess, const double** out) {
  *out = nullptr;
  if (kind != OBNSL_SCORE_BDEU) return OBNSL_OK;
  auto& slot = d->lg_alpha[ess];
  if (!slot) {
    std::vector<double> h((REDUCED_MAX_CELLS + 1) * (1 + LG_SMALL), 0.0);
    for (uint64_t c = 1; c <= REDUCED_MAX_CELLS; ++c) {
      const double a = ess / (double)c;
      h[c] = std::lgamma(a);
      double* small = h.data() + (REDUCED_MAX_CELLS + 1) + c * LG_SMALL;
      double acc = 0.0;
      for (int n = 1; n < LG_SMALL; ++n) {
        acc += std::log(a + (double)(n - 1));
        small[n] = acc;
      }
    }
    slot = std::make_unique<DevBuf<double>>();
    HIP_TRY(slot->upload(h));
  }
  *out = slot->p;
  return OBNSL_OK;
}

// Reduced mode: expand + evaluate the functional G for every table of `plan`
// (and the scores, when `scores` is given) -- one score_reduced_kernel launch.
obnsl_status run_functional(obnsl_data* d, Plan* plan, const ReducedViews& views, uint32_t k,
                            uint64_t max_cells, obnsl_score_kind kind, double ess,
                            const double* gprev, double g0, double* scores) {
  HIP_TRY(plan->G.ensure(std::max<uint64_t>(binom_host(d->n, k), 1)));
  const double* lg_alpha = nullptr;
  OB_TRY(lg_alpha_table(d, kind, ess, &lg_alpha));
  Batch* b = plan->batches[0].get();
  if (b->ntables) {
    ScorerShape sh;
    OB_TRY(scorer_shape(d, plan, k, max_cells, &sh));
    ReducedScorerArgs sa;
    sa.views = views;
    sa.rgs = b->rgs.p;
    sa.nrgs = b->nrgs;
    sa.ntables = b->ntables;
    sa.tables_per_wg = sh.tpw;
    sa.card = d->d_card.p;
    sa.binom = d->d_binom.p;
    sa.num_rows = (uint32_t)d->total_rows;
    sa.kind = (int)kind;
    sa.ess = ess;
    sa.G = plan->G.p;
    sa.Gprev = gprev;
    sa.g0 = g0;
    sa.scores = scores;
    sa.lg_alpha = lg_alpha;
    sa.e_stride = sh.e_stride;
    sa.wg_rg = b->wg_rg.p;
    sa.best_key = nullptr;
    launch_score_reduced(d->stream, k, sh.parts, sh.grid, sa);
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
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (d->timed_now()) {
      OB_TRY(next_event_pair(d, k, &ev));
      HIP_TRY(hipEventRecord(ev->first, d->stream));
    }
    launch_count_hist(d->stream, nt * (uint32_t)nchunks, lds, d->codes.p, d->stride, d->N, d->n, d->d_card.p,
                      d->d_binom.p, k, hb->rank0, (uint32_t)nchunks, rpc, hb->off.p, d->counts.p);
    HIP_TRY(hipGetLastError());
    if (ev) HIP_TRY(hipEventRecord(ev->second, d->stream));
    launch_score_tables(d->stream, d->counts.p, hb->tds.p, (uint64_t)nt, (int)kind, ess, d_scores + hb->rank0 * k);
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
  const bool allow_reduced = !env().no_reduced;
  const bool force_hist = env().force_hist;
  const bool force_bits = env().force_bitgemm;
  bool use = allow_reduced && max_cells <= REDUCED_MAX_CELLS && !force_hist && k <= (uint32_t)KFAST;
  if (use && !force_bits) {
    // reduced cells of all size-k tables = e_k(r_1 - 1, ..., r_n - 1), the elementary
    // symmetric polynomial of the reduced radices (no plan is built for an estimate)
    std::vector<double> e(k + 1, 0.0);
    e[0] = 1.0;
    for (uint32_t v = 0; v < d->n; ++v)
      for (uint32_t j = std::min(k, v + 1); j >= 1; --j) e[j] += e[j - 1] * (double)(d->card[v] - 1);
    // (from the rows of the whole data set, so that ranks holding slices of different
    // lengths take the same path)
    const double words = d->rows_sharded() ? (double)((d->total_rows + 31) / 32) : (double)d->wp;
    const double est_bit = e[k] * words * (use_mfma_gemm() ? 2.0e-5 : 6.7e-5);
    const double est_hist = (double)binom_host(d->n, k) * (double)d->total_rows * 4.5e-4;
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

obnsl_status run_pack(obnsl_data* d, uint32_t* bad_code) {
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
      zl.n[zl.count] = (uint64_t)b0->nitems * p->tile * p->tile + IMG_TAIL;
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
  launch_pack(d->stream, d->codes.p, d->N, d->stride, d->n, d->d_card.p, d->d_plane_base.p, d->planes.p, d->wp,
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

// obnsl_score_all: the scoring phase of exhaustive_search
// (src/structure_learning/exhaustive_search.cpp:47-73) for one subset size.
obnsl_status score_all_impl(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t k, uint32_t shard,
                            uint32_t num_shards, double* scores_out, int out_on_device) {
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
    const double N_d = (double)d->total_rows;
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
  } else if (d->rows_sharded()) {
    return fail(OBNSL_ERR_UNSUPPORTED,
                "a data set that holds a slice of the rows serves subset sizes of the bit-plane contraction only");
  } else if (!env().force_bitgemm || k > (uint32_t)KFAST) {
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
      launch_score_plan(d->stream, sgrid, d->counts.p, b->rgs.p, b->nrgs, plan->cgs.p, d->d_card.p, plan->kr,
                        plan->kc, b->ntables, (int)kind, ess, d_scores);
      HIP_TRY(hipGetLastError());
    }
  }
  if (!out_on_device) {
    HIP_TRY(hipMemcpyAsync(scores_out, d_scores, nsub * k * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
  }
  return OBNSL_OK;
}

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
  // (run_pack).  A short pass -- where those launches would compete for the chip's
  // workgroup slots -- counts every size in ONE launch on the main stream instead.
  st = join_side(d);
  if (st != OBNSL_OK) return st;
  Plan* top = nullptr;
  st = build_plan(d, kmax, 0, 1, true, &top);
  if (st != OBNSL_OK) return st;
  if (top_is_requested && num_shards > 1 && top->total_items > whole_below(top)) {
    st = build_plan(d, kmax, shard, num_shards, true, &top);
    if (st != OBNSL_OK) return st;
  }
  {
    // every size that still has to be counted for these row bitmaps, largest first
    std::vector<CountPart> parts;
    Plan* sliced_whole = nullptr;
    bool ok = top->counted_epoch != d->epoch && !top->batches.empty();
    auto own_table = [&](Plan* p) -> obnsl_status {
      if (!p->red.p) {
        HIP_TRY(p->red.ensure(std::max<uint64_t>(p->batches[0]->ncells, 1)));
        p->zeroed_epoch = 0;  // fresh memory: not cleared by any pack
      }
      return OBNSL_OK;
    };
    if (ok) {
      OB_TRY(own_table(top));
      parts.push_back({top, top->batches[0].get(), top->red.p, top->zeroed_epoch == d->epoch});
    }
    for (uint32_t j = kmax - 1; ok && j >= 1; --j) {
      Plan* full = nullptr;
      OB_TRY(build_plan(d, j, 0, 1, true, &full));
      if (full->counted_epoch == d->epoch) continue;
      if (j == kmax - 1 && shard_lower && top_is_requested && num_shards > 1 && full->total_items >= num_shards) {
        // this rank's slice only, written at its place in the whole table; the
        // rest stays zero and the caller sums the tables of all ranks
        Plan* part = nullptr;
        OB_TRY(build_plan(d, j, shard, num_shards, true, &part));
        const Batch* wb = full->batches[0].get();
        Batch* pb = part->batches[0].get();
        if (part->slice_off + pb->ncells > wb->ncells)
          return fail(OBNSL_ERR_RUNTIME, "internal: slice outside the count table");
        if (pb->nitems == 0) {
          ok = false;   // (an empty slice: the separate launches handle it)
          break;
        }
        OB_TRY(own_table(full));
        sliced_whole = full;
        parts.push_back({part, pb, full->red.p + part->slice_off, false});
      } else {
        OB_TRY(own_table(full));
        parts.push_back({full, full->batches[0].get(), full->red.p, full->zeroed_epoch == d->epoch});
      }
    }
    if (ok && can_count_together(d, parts)) {
      if (sliced_whole) {
        // the whole table must be clear around the slice
        Batch* wb = sliced_whole->batches[0].get();
        if (sliced_whole->zeroed_epoch != d->epoch)
          HIP_TRY(hipMemsetAsync(sliced_whole->red.p, 0, wb->ncells * sizeof(uint32_t), d->stream));
        sliced_whole->wants_zero = true;
        for (CountPart& p : parts)
          if (p.table != p.plan->red.p) p.prezeroed = true;
      }
      OB_TRY(launch_count_together(d, parts));
      for (CountPart& p : parts) p.plan->counted_epoch = d->epoch;
      if (sliced_whole) {
        sliced_whole->counted_epoch = d->epoch;
        sliced_whole->awaits_sum = true;
        if (sum_size_out) *sum_size_out = kmax - 1;
        if (sum_cells_out) *sum_cells_out = sliced_whole->batches[0]->ncells;
      }
      return OBNSL_OK;
    }
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

// Ranks that split the ROWS: every subset size 1..kmax counted whole over this handle's
// slice, the tables in one contiguous range of device memory (so that one all-reduce sums
// them over the ranks), each marked as a partial sum until the caller acknowledges the sum.
obnsl_status precount_rows_impl(obnsl_data* d, uint32_t kmax, uint32_t** d_counts_out, uint64_t* words_out) {
  if (!d || !d_counts_out || !words_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  *d_counts_out = nullptr;
  *words_out = 0;
  if (kmax == 0 || kmax > (uint32_t)KMAX || kmax > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "subset size must be in [1, min(num_vars, OBNSL_MAX_SUBSET)]");
  OB_TRY(set_device(d->device));
  for (uint32_t j = 1; j <= kmax; ++j) {
    bool red = false;
    OB_TRY(choose_reduced(d, j, &red, nullptr));
    if (!red)
      return fail(OBNSL_ERR_UNSUPPORTED, "subset size " + std::to_string(j) +
                                             " is counted per row on this data set: its tables cannot be "
                                             "summed over ranks that split the rows");
  }
  OB_TRY(join_side(d));
  std::vector<Plan*> whole(kmax + 1, nullptr);
  for (uint32_t j = 1; j <= kmax; ++j) OB_TRY(build_plan(d, j, 0, 1, true, &whole[j]));
  if (d->arena_kmax != kmax) {
    std::vector<uint64_t> off(kmax + 1, 0);
    uint64_t total = 0;
    for (uint32_t j = 1; j <= kmax; ++j) {
      off[j] = total;
      total += (std::max<uint64_t>(whole[j]->batches[0]->ncells, 1) + 63) / 64 * 64;
    }
    HIP_TRY(hipStreamSynchronize(d->stream));   // nothing in flight reads the tables being replaced
    for (auto& kv : d->plans) {
      Plan* p = kv.second.get();
      if (p->red.borrowed) {
        p->red.reset();
        p->counted_epoch = 0;
      }
    }
    d->arena.reset();
    HIP_TRY(d->arena.ensure(total));
    // (on the data set's stream: a fill on the null stream is not ordered before the count
    // launches of a non-blocking stream)
    HIP_TRY(hipMemsetAsync(d->arena.p, 0, total * sizeof(uint32_t), d->stream));
    for (uint32_t j = 1; j <= kmax; ++j) {
      whole[j]->red.borrow(d->arena.p + off[j], std::max<uint64_t>(whole[j]->batches[0]->ncells, 1));
      whole[j]->counted_epoch = 0;
      whole[j]->zeroed_epoch = d->epoch;   // cleared just now
    }
    d->arena_kmax = kmax;
    d->arena_words = total;
  }
  std::vector<uint64_t> before(kmax + 1, 0);
  for (uint32_t j = 1; j <= kmax; ++j) before[j] = whole[j]->counted_epoch;
  uint32_t fresh = 0, owing = 0;
  d->counting_rows = true;
  obnsl_status st = precount_impl(d, kmax, 0, 1, false, nullptr, nullptr);
  d->counting_rows = false;
  if (st != OBNSL_OK) return st;
  for (uint32_t j = 1; j <= kmax; ++j) {
    if (whole[j]->counted_epoch != d->epoch) {   // (precount leaves nothing out; keep the table honest if it did)
      d->counting_rows = true;
      st = ensure_counts(d, whole[j]);
      d->counting_rows = false;
      if (st != OBNSL_OK) return st;
    }
    if (whole[j]->counted_epoch != before[j]) {
      ++fresh;
      if (d->rows_sharded()) whole[j]->awaits_sum = true;
    }
    if (whole[j]->awaits_sum) ++owing;
  }
  if (fresh != 0 && fresh != kmax)
    return fail(OBNSL_ERR_RUNTIME, "internal: some count tables of a row slice are older than others");
  OB_TRY(join_side(d));   // what the caller queues on the data set's stream next sees every table
  *d_counts_out = d->arena.p;
  // nothing to sum: the tables of these row bitmaps were counted and summed before
  *words_out = (fresh == 0 && owing == 0 && d->rows_sharded()) ? 0 : d->arena_words;
  return OBNSL_OK;
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

// Per-node best parent sets from device score tables: reset of the arg-max words and the
// pass that finds each node's largest score (both skipped when the fused scorer has done
// them), then the pass that picks the smallest (size, rank) among the entries that reach it.
// `keys`: 2 n words of device memory, keys [0, n), ids [n, 2 n); null = the data set's own
// buffer (allocated on first use).
// decode_out (device, 3 n doubles; only with the data set's own buffer): the last workgroup
// of the second pass writes the records, which saves the launch of best_decode_kernel.
static obnsl_status own_best_words(obnsl_data* d) {
  if (d->best_key.n >= 2 * (size_t)d->n + 1) return OBNSL_OK;
  HIP_TRY(d->best_key.ensure(2 * (size_t)d->n + 1));
  // (the word behind the keys and ids counts the workgroups that are done: zero between launches)
  HIP_TRY(hipMemsetAsync(d->best_key.p + 2 * (size_t)d->n, 0, sizeof(unsigned long long), d->stream));
  return OBNSL_OK;
}

obnsl_status subset_vars_table(obnsl_data* d, uint32_t k, bool may_build, const uint16_t** out) {
  *out = nullptr;
  auto it = d->subset_vars.find(k);
  if (it != d->subset_vars.end()) {
    *out = it->second->p;
    return OBNSL_OK;
  }
  const uint64_t nsub = binom_host(d->n, k);
  if (!may_build || nsub == 0 || nsub * k * sizeof(uint16_t) > SUBSET_VARS_MAX_BYTES) return OBNSL_OK;
  auto buf = std::make_unique<DevBuf<uint16_t>>();
  HIP_TRY(buf->ensure(nsub * k));
  launch_subset_vars(d->stream, d->d_binom.p, d->n, k, nsub, buf->p);
  HIP_TRY(hipGetLastError());
  *out = buf->p;
  d->subset_vars[k] = std::move(buf);
  return OBNSL_OK;
}

obnsl_status queue_best_parents(obnsl_data* d, const double* const* d_scores, uint32_t max_parents,
                                unsigned long long* keys, bool skip_first_pass, double* decode_out) {
  if (max_parents + 1 > (uint32_t)KMAX || max_parents + 1 > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  OB_TRY(set_device(d->device));
  // (a caller that brings its own arg-max words is one that must not allocate: the tables
  // of the subsets' variables are used where they exist, never built)
  const bool may_build = keys == nullptr || keys == d->best_key.p;
  if (!keys) {
    OB_TRY(own_best_words(d));
    keys = d->best_key.p;
  }
  if (decode_out && keys != d->best_key.p)
    return fail(OBNSL_ERR_RUNTIME, "internal: fused record write without the data set's own arg-max words");
  unsigned long long* key = keys;
  unsigned long long* id = keys + d->n;
  if (!skip_first_pass) launch_best_init(d->stream, key, id, d->n);
  BestArgs args;
  std::memset(&args, 0, sizeof(args));
  args.kmax = max_parents + 1;
  for (uint32_t k = 1; k <= max_parents + 1; ++k) {
    const uint64_t nsub = binom_host(d->n, k);
    args.scores[k - 1] = d_scores[k - 1];
    OB_TRY(subset_vars_table(d, k, may_build, &args.vars[k - 1]));
    args.nsubsets[k - 1] = nsub;
    args.block_start[k] = args.block_start[k - 1] + (uint32_t)std::min<uint64_t>((nsub + 255) / 256, 512);
  }
  for (int pass = skip_first_pass ? 1 : 0; pass < 2; ++pass)
    launch_best_parents(d->stream, args, d->n, d->d_binom.p, key, id, pass, pass == 1 ? decode_out : nullptr,
                        decode_out ? reinterpret_cast<unsigned int*>(d->best_key.p + 2 * (size_t)d->n) : nullptr);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

// 1 per colex rank of a table the sharded plan `own` lists (built once per plan).
static obnsl_status shard_mask(obnsl_data* d, Plan* own, uint32_t k, const uint8_t** out) {
  if (!own->own_mask_built) {
    std::vector<uint8_t> h(std::max<uint64_t>(binom_host(d->n, k), 1), 0);
    for (auto& bp : own->batches) {
      const Batch* b = bp.get();
      for (size_t g = 0; g < b->h_rank_part.size(); ++g)
        for (uint32_t c = 0; c < b->h_ncolgroups[g]; ++c) {
          const uint64_t r = b->h_rank_part[g] + c;
          if (r >= h.size()) return fail(OBNSL_ERR_RUNTIME, "internal: table rank outside its subset size");
          h[r] = 1;
        }
    }
    HIP_TRY(own->own_mask.upload(h));
    own->own_mask_built = true;
  }
  *out = own->own_mask.p;
  return OBNSL_OK;
}

// The scoring phase of exhaustive_search (src/structure_learning/exhaustive_search.cpp:47-73)
// for every subset size 1..max_parents+1 at once, and the per-node best parent sets: one
// launch evaluates the functional of every table of every size, one turns them into
// scores (this shard's) and finds each node's best score, one breaks ties, one writes the
// records.  Falls back to one obnsl_score_all per size where a size is counted per row
// or is larger than the fused launches serve.  d_best_out may be null (scores only).
obnsl_status score_pass_impl(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t max_parents, uint32_t shard,
                             uint32_t num_shards, double* const* d_tables, double* d_best_out,
                             unsigned long long* keys) {
  if (!d || !d_tables) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  const uint32_t kmax = max_parents + 1;
  if (kmax > (uint32_t)KMAX || kmax > d->n)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  if (num_shards == 0 || shard >= num_shards)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "shard must be in [0, num_shards)");
  for (uint32_t k = 1; k <= kmax; ++k)
    if (!d_tables[k - 1]) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null score table");
  OB_TRY(check_kind(kind));
  OB_TRY(set_device(d->device));
  bool fused = kmax <= (uint32_t)FUSED_KMAX && !env().no_fused;
  uint64_t max_cells = 1;
  for (uint32_t k = 1; k <= kmax && fused; ++k) {
    bool red = false;
    OB_TRY(choose_reduced(d, k, &red, k == kmax ? &max_cells : nullptr));
    if (!red) fused = false;
  }
  if (!fused) {
    for (uint32_t k = 1; k <= kmax; ++k)
      OB_TRY(score_all_impl(d, kind, ess, k, shard, num_shards, d_tables[k - 1], 1));
    if (d_best_out) {
      std::vector<const double*> ptrs(d_tables, d_tables + kmax);
      OB_TRY(queue_best_parents(d, ptrs.data(), max_parents, keys, false));
      if (!keys) keys = d->best_key.p;
      launch_best_decode(d->stream, keys, keys + d->n, d->n, d_best_out);
      HIP_TRY(hipGetLastError());
    }
    return OBNSL_OK;
  }
  if (d_best_out && !keys) {
    OB_TRY(own_best_words(d));
    keys = d->best_key.p;
  }
  FunctionalAllArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  ScoresAllArgs sa;
  std::memset(&sa, 0, sizeof(sa));
  Plan* listed[FUSED_KMAX + 1] = {nullptr};   // whose tables size j's workgroups walk (and whose G they fill)
  uint32_t grid = 0;
  size_t lds = 0;
  for (uint32_t j = 1; j <= kmax; ++j) {
    Plan* full = nullptr;
    OB_TRY(build_plan(d, j, 0, 1, true, &full));
    Plan* own = full;
    if (num_shards > 1) OB_TRY(build_plan(d, j, shard, num_shards, true, &own));
    // the counts the scorer reads: the whole table of every size below the top one; the top
    // size's whole table when it was counted whole (one shard, or a small size), else this
    // shard's own
    Plan* use = full;
    if (j == kmax && num_shards > 1 &&
        !(full->total_items <= whole_below(full) || full->counted_epoch == d->epoch))
      use = own;
    OB_TRY(ensure_counts(d, use));
    fa.views.v[j].red = use->red.p;
    fa.views.v[j].rowoff = use->rowoff.p;
    fa.views.v[j].cgs = use->cgs.p;
    fa.views.v[j].kr = use->kr;
    fa.views.v[j].kc = use->kc;
    // the functional of every table of the sizes below the top one (the next size's
    // scores subtract them), of this shard's tables at the top
    Plan* lp = j == kmax ? own : full;
    listed[j] = lp;
    HIP_TRY(lp->G.ensure(std::max<uint64_t>(binom_host(d->n, j), 1)));
    uint64_t own_cells = 1;
    bool red_j = false;
    OB_TRY(choose_reduced(d, j, &red_j, &own_cells));
    ScorerShape sh;
    Batch* b = lp->batches[0].get();
    if (b->ntables) OB_TRY(scorer_shape(d, lp, j, own_cells, &sh));
    FunctionalSizeArgs& z = fa.size[j];
    // the tables' records: read when they exist for these count-table layouts, else written
    // by this launch (not where an offset would not fit its 32 bits, or the records 2 GiB)
    z.recs = nullptr;
    z.rec_mode = 0;
    {
      const void* key[2 * (KFAST + 1)] = {nullptr};
      bool fits = !env().no_table_records;
      for (uint32_t i = 1; i <= j; ++i) {
        key[2 * i] = fa.views.v[i].rowoff;
        key[2 * i + 1] = fa.views.v[i].cgs;
      }
      for (uint32_t i = 1; i <= j && fits; ++i) {
        Plan* pi = nullptr;   // the plan whose table size i's view reads
        for (auto& kv : d->plans)
          if (kv.second->red.p == fa.views.v[i].red && kv.second->k == i) pi = kv.second.get();
        if (!pi || pi->batches.empty() || pi->batches[0]->ncells >= (1ull << 32)) fits = false;
      }
      const uint64_t words = (uint64_t)b->ntables * table_record_words((int)j);
      if (fits && b->ntables && words * sizeof(uint32_t) <= (2ull << 30)) {
        const bool same = lp->recs_ready && std::equal(key, key + 2 * (KFAST + 1), lp->recs_key);
        if (!same) {
          HIP_TRY(lp->recs.ensure(words));
          std::copy(key, key + 2 * (KFAST + 1), lp->recs_key);
          lp->recs_ready = true;   // (by the launch queued below; later launches follow it on the stream)
        }
        z.recs = lp->recs.p;
        z.rec_mode = same ? 2 : 1;
      }
    }
    z.rgs = b->rgs.p;
    z.wg_rg = b->wg_rg.p;
    z.G = lp->G.p;
    z.ntables = b->ntables;
    z.tables_per_wg = std::max(sh.tpw, 1u);
    z.parts = sh.parts;
    z.e_stride = sh.e_stride;
    z.block_start = grid;
    grid += sh.grid;
    lds = std::max(lds, (size_t)sh.parts * sh.e_stride * sizeof(uint32_t));
    sa.G[j] = lp->G.p;
    sa.scores[j] = d_tables[j - 1];
    sa.nsubsets[j] = binom_host(d->n, j);
    if (num_shards > 1) OB_TRY(shard_mask(d, own, j, &sa.mask[j]));
    OB_TRY(subset_vars_table(d, j, true, &sa.vars[j]));
  }
  for (uint32_t j = kmax + 1; j <= (uint32_t)FUSED_KMAX; ++j) fa.size[j].block_start = grid;
  const double* lg_alpha = nullptr;
  OB_TRY(lg_alpha_table(d, kind, ess, &lg_alpha));
  fa.kmax = kmax;
  fa.card = d->d_card.p;
  fa.binom = d->d_binom.p;
  fa.num_rows = (uint32_t)d->total_rows;
  fa.kind = (int)kind;
  fa.ess = ess;
  fa.lg_alpha = lg_alpha;
  fa.best_key = d_best_out ? keys : nullptr;
  fa.n = d->n;
  if (grid == 0) return fail(OBNSL_ERR_RUNTIME, "internal: fused scorer without tables");
  // (a launch reads records only when every size has them: else those that have them
  // are worked out from the descriptors once more, without being rewritten)
  {
    bool all_read = true;
    for (uint32_t j = 1; j <= kmax; ++j) all_read = all_read && (fa.size[j].rec_mode == 2 || fa.size[j].ntables == 0);
    if (!all_read)
      for (uint32_t j = 1; j <= kmax; ++j)
        if (fa.size[j].rec_mode == 2) fa.size[j].rec_mode = 0;
  }
  launch_functional_all(d->stream, grid, lds, fa);
  HIP_TRY(hipGetLastError());
  const double N_d = (double)d->total_rows;
  sa.g0 = kind == OBNSL_SCORE_BDEU ? std::lgamma(N_d + ess) - std::lgamma(ess) : N_d * std::log(N_d);
  sa.kmax = kmax;
  sa.card = d->d_card.p;
  sa.binom = d->d_binom.p;
  sa.n = d->n;
  sa.num_rows = (uint32_t)d->total_rows;
  sa.kind = (int)kind;
  sa.best_key = d_best_out ? keys : nullptr;
  sa.block_start[1] = 0;
  for (uint32_t j = 1; j <= kmax; ++j)
    sa.block_start[j + 1] = sa.block_start[j] + (uint32_t)std::min<uint64_t>((sa.nsubsets[j] + 255) / 256, 2048);
  launch_scores_best(d->stream, sa);
  HIP_TRY(hipGetLastError());
  for (uint32_t j = 1; j <= kmax; ++j) {
    listed[j]->g_epoch = d->epoch;
    listed[j]->g_kind = (int)kind;
    listed[j]->g_ess = ess;
  }
  if (d_best_out) {
    std::vector<const double*> ptrs(d_tables, d_tables + kmax);
    // (the records in a launch of their own: with the last workgroup of the tie-break pass
    // writing them -- queue_best_parents' decode_out -- that pass took 25.8 us instead of
    // 8.7 + 4.0 for the two launches: every workgroup waits on a device-wide fence first)
    OB_TRY(queue_best_parents(d, ptrs.data(), max_parents, keys, true));
    launch_best_decode(d->stream, keys, keys + d->n, d->n, d_best_out);
    HIP_TRY(hipGetLastError());
  }
  return OBNSL_OK;
}

}  // namespace obnsl_internal
