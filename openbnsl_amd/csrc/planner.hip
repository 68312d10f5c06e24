// planner.hip -- host side of the counting and scoring launches: the plan of a
// subset size (rows, columns, tiles and work units of its contraction; built once
// per data set and kept), the launches that fill its count table, the choice
// between the contraction and the per-row histogram path, and the order of the
// launches of one pass over the streams of a data set.  What it replaces is the
// scoring loop of exhaustive_search
// (src/structure_learning/exhaustive_search.cpp:47-73): one table and k scores
// per subset, subset after subset.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine_state.h"

namespace obnsl_internal {

bool use_mfma_gemm() { return !env().gemm_valu; }

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

}  // namespace

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
    const uint64_t edge = use_mfma_gemm() ? (uint64_t)MFMA_TILE : (uint64_t)env().valu_tile;
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
    plan->mfma = use_mfma_gemm();
    plan->tile = plan->mfma ? MFMA_TILE : (uint32_t)env().valu_tile;
    make_tiles(plan->tile, tiles);
    if (plan->mfma && (kr == 2 || (kr == 3 && kc == 2))) {
      // launches of many tiles take 256 x 256 tiles
      const int big = env().mfma_tile;
      const size_t big_from = (size_t)env().big_tile_from;
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
    // the column blocks the first 64 rows of each tile reach (the rows are sorted by their
    // number of valid columns: the 64th row has the most of them)
    b->h_item_top.assign(tiles.size(), 0);
    for (size_t i = 0; i < tiles.size(); ++i) {
      const uint64_t first = (uint64_t)items[i].x * plan->tile, c0 = (uint64_t)items[i].y * plan->tile;
      const uint64_t last = std::min<uint64_t>(first + 64, rows.size()) - 1;
      const uint64_t lim = rows[last].ncols;
      b->h_item_top[i] = lim <= c0 ? 0u : (uint32_t)std::min<uint64_t>((lim - c0 + 31) / 32, b->h_item_blocks[i]);
    }
    b->nrows = (uint32_t)rows.size();
    b->nitems = (uint32_t)items.size();
    b->nrgs = (uint32_t)rgs.size();
    b->ntables = tables;
    b->ncells = off;
    HIP_TRY(b->rows.upload(rows));
    HIP_TRY(b->items.upload(items));
    HIP_TRY(b->rgs.upload(rgs));
    b->h_tbl_start.resize(rgs.size());
    b->h_rank_part.resize(rgs.size());
    b->h_ncolgroups.resize(rgs.size());
    for (size_t i = 0; i < rgs.size(); ++i) {
      b->h_tbl_start[i] = rgs[i].tbl_start;
      b->h_rank_part[i] = rgs[i].rank_part;
      b->h_ncolgroups[i] = rgs[i].ncolgroups;
    }
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

// What a contraction step of a tile costs, by its number of 32-column blocks that hold
// a valid column: a tile with fewer blocks amortises its A operand over fewer
// instructions (measured 31 / 24 / 20 ns per instruction for 1 / 2 / 4 blocks).
static uint32_t unit_weight(uint32_t blk) { return 2 * blk + 1; }
// (a full-width 128 x 128 tile whose first 64 rows reach none of its columns is contracted
// by all four waves on its other 64 rows, in about half the time: runs of units only)
static uint32_t tile_weight(uint32_t blk, uint32_t top_blocks, uint32_t tile_edge) {
  if (tile_edge != 128 || top_blocks != 0) return unit_weight(blk);
  return blk == 4 ? 5 : unit_weight(blk);
}

// One tile of a launch's work: which part (subset size) it belongs to and where it is.
struct TileRef {
  uint32_t part, tile, row_tile, col_tile, blocks, top_blocks;
};
struct PartShape {   // what check_units needs to know of a part
  const Batch* b;
  uint32_t tile_edge, ncols_total;
};

static void tiles_of(const Batch* b, uint32_t part, std::vector<TileRef>& out) {
  for (size_t i = 0; i < b->h_items.size(); ++i)
    out.push_back({part, (uint32_t)i, b->h_items[i].x, b->h_items[i].y, b->h_item_blocks[i], b->h_item_top[i]});
}

// Host-side check of a unit list before it is uploaded: every unit inside its part's
// tile list, rows, columns and steps; every tile's steps covered exactly once; the
// workgroup runs a partition of the list.  A list that fails is a bug in this file;
// it is refused here rather than launched.
static obnsl_status check_units(const std::vector<PartShape>& parts, const std::vector<MfmaUnit>& units,
                                const std::vector<uint32_t>& unit_start, uint32_t steps_total) {
  std::vector<std::vector<std::vector<std::pair<uint32_t, uint32_t>>>> per(parts.size());
  for (size_t p = 0; p < parts.size(); ++p) per[p].resize(parts[p].b->h_items.size());
  for (const MfmaUnit& u : units) {
    if (u.part >= parts.size()) return fail(OBNSL_ERR_RUNTIME, "internal: work unit outside the launch's parts");
    const Batch* b = parts[u.part].b;
    const uint32_t tb = parts[u.part].tile_edge;
    if (u.tile >= b->h_items.size()) return fail(OBNSL_ERR_RUNTIME, "internal: work unit outside the tile list");
    const uint2 it = b->h_items[u.tile];
    if (u.row_tile != it.x || u.col_tile != it.y)
      return fail(OBNSL_ERR_RUNTIME, "internal: work unit does not match its tile");
    if ((uint64_t)u.row_tile * tb >= b->nrows || (uint64_t)u.col_tile * tb >= parts[u.part].ncols_total)
      return fail(OBNSL_ERR_RUNTIME, "internal: work unit outside the rows or columns of the launch");
    if (u.blocks < 1 || u.blocks > tb / 32 || u.blocks != b->h_item_blocks[u.tile])
      return fail(OBNSL_ERR_RUNTIME, "internal: work unit with a bad column-block count");
    if (u.top_blocks > u.blocks || u.top_blocks != b->h_item_top[u.tile])
      return fail(OBNSL_ERR_RUNTIME, "internal: work unit with a bad column-block count of its first rows");
    if (u.step_lo >= u.step_hi || u.step_hi > steps_total || u.step_hi - u.step_lo > MFMA_MAX_STEPS)
      return fail(OBNSL_ERR_RUNTIME, "internal: work unit with a bad step range");
    per[u.part][u.tile].push_back({u.step_lo, u.step_hi});
  }
  for (auto& part : per)
    for (auto& v : part) {
      std::sort(v.begin(), v.end());
      uint32_t at = 0;
      for (auto& r : v) {
        if (r.first != at)
          return fail(OBNSL_ERR_RUNTIME, "internal: work units do not cover a tile's steps exactly once");
        at = r.second;
      }
      if (at != steps_total)
        return fail(OBNSL_ERR_RUNTIME, "internal: work units do not cover a tile's steps exactly once");
    }
  if (!unit_start.empty()) {
    if (unit_start.front() != 0 || unit_start.back() != units.size())
      return fail(OBNSL_ERR_RUNTIME, "internal: workgroup runs do not cover the unit list");
    for (size_t w = 0; w + 1 < unit_start.size(); ++w)
      if (unit_start[w] >= unit_start[w + 1]) return fail(OBNSL_ERR_RUNTIME, "internal: empty or reversed workgroup run");
  }
  return OBNSL_OK;
}

static void dump_units(size_t ntiles, const std::vector<MfmaUnit>& units, uint32_t steps_total, uint32_t target,
                       uint32_t nworkgroups) {
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> hist;
  for (auto& u : units) hist[{u.part, u.blocks, u.step_hi - u.step_lo}]++;
  std::fprintf(stderr, "units: %zu tiles, %zu units, %u workgroups, steps %u, target %u\n", ntiles, units.size(),
               nworkgroups, steps_total, target);
  for (auto& h : hist)
    std::fprintf(stderr, "  part %u blocks %u steps %u: %u units\n", std::get<0>(h.first), std::get<1>(h.first),
                 std::get<2>(h.first), h.second);
}

// Work units of the matrix-core launch of one batch: every tile is cut along the
// contraction into runs whose cost (steps x column blocks) is close to
// total cost / target, so a launch is `target` workgroups of equal length
// (built once per batch and row count).
static obnsl_status build_units(Batch* b, const Plan* plan, uint32_t steps_total, uint32_t target) {
  if (b->units_steps == steps_total && b->units_target == target && b->nunits) return OBNSL_OK;
  uint64_t total = 0;
  for (uint32_t blk : b->h_item_blocks) total += (uint64_t)unit_weight(blk) * steps_total;
  const double per_unit = std::max(1.0, (double)total / (double)std::max(1u, target));
  std::vector<MfmaUnit> units;
  std::vector<std::pair<uint32_t, MfmaUnit>> tmp;  // (cost, unit)
  bool split = false;
  for (size_t i = 0; i < b->h_items.size(); ++i) {
    const uint32_t blk = b->h_item_blocks[i];
    uint32_t ns = (uint32_t)std::max(1.0, std::floor((double)unit_weight(blk) * steps_total / per_unit + 0.5));
    ns = std::min(ns, std::max(1u, steps_total / 4));            // at least 4 steps per unit
    ns = std::max(ns, (steps_total + MFMA_MAX_STEPS - 1) / MFMA_MAX_STEPS);  // f32 counts stay exact
    ns = std::min(ns, steps_total);
    if (ns > 1) split = true;
    for (uint32_t sidx = 0; sidx < ns; ++sidx) {
      MfmaUnit u;
      std::memset(&u, 0, sizeof(u));
      u.row_tile = b->h_items[i].x;
      u.col_tile = b->h_items[i].y;
      u.step_lo = (uint32_t)((uint64_t)steps_total * sidx / ns);
      u.step_hi = (uint32_t)((uint64_t)steps_total * (sidx + 1) / ns);
      u.blocks = blk;
      u.tile = (uint32_t)i;
      u.top_blocks = b->h_item_top[i];
      tmp.push_back({unit_weight(blk) * (u.step_hi - u.step_lo), u});
    }
  }
  std::stable_sort(tmp.begin(), tmp.end(),
                   [](const std::pair<uint32_t, MfmaUnit>& x, const std::pair<uint32_t, MfmaUnit>& y) {
                     return x.first > y.first;
                   });
  units.reserve(tmp.size());
  for (auto& t : tmp) units.push_back(t.second);
  OB_TRY(check_units({{b, plan->tile, plan->ncols_total}}, units, {}, steps_total));
  if (env().dump_units) dump_units(b->h_items.size(), units, steps_total, target, (uint32_t)units.size());
  HIP_TRY(b->units.upload(units));
  b->nunits = (uint32_t)units.size();
  b->nworkgroups = b->nunits;
  b->runs = false;
  b->units_steps = steps_total;
  b->units_target = target;
  b->units_split = split;
  return OBNSL_OK;
}

// The launch as runs of units.  The contraction of the whole launch is laid out as one
// line -- tile after tile in the order given, each step weighted by its cost -- and cut in
// phases: the first phase into `slots` runs of equal cost, one per workgroup the chip
// holds at once; phase p after it into pieces[p] x slots shorter runs that even out the
// end of the launch.  A run that crosses a tile boundary becomes one unit per tile.
static void cut_runs(const std::vector<TileRef>& tiles, uint32_t steps_total, uint32_t slots,
                     const std::vector<double>& phases, const std::vector<uint32_t>& pieces,
                     std::vector<MfmaUnit>& units, std::vector<uint32_t>& unit_start) {
  units.clear();
  unit_start.clear();
  const size_t nt = tiles.size();
  std::vector<uint64_t> start(nt + 1, 0);   // position of each tile on the line
  for (size_t i = 0; i < nt; ++i)
    start[i + 1] = start[i] + (uint64_t)tile_weight(tiles[i].blocks, tiles[i].top_blocks, 128) * steps_total;
  const double total = (double)start[nt];
  // a point of the line as (tile, step), rounded to whole steps
  auto locate = [&](double x, size_t& tile, uint32_t& step) {
    if (x >= total) {
      tile = nt;
      step = 0;
      return;
    }
    size_t i = std::upper_bound(start.begin(), start.end(), (uint64_t)x) - start.begin() - 1;
    i = std::min(i, nt - 1);
    const double w = (double)tile_weight(tiles[i].blocks, tiles[i].top_blocks, 128);
    uint32_t st = (uint32_t)std::min<double>(steps_total, std::floor((x - (double)start[i]) / w + 0.5));
    if (st >= steps_total) {
      ++i;
      st = 0;
    }
    tile = i;
    step = st;
  };
  size_t cur_tile = 0;
  uint32_t cur_step = 0;
  double phase_lo = 0.0;
  for (size_t p = 0; p < phases.size(); ++p) {
    const double phase_hi = p + 1 == phases.size() ? 1.0 : std::min(1.0, phase_lo + phases[p]);
    const uint32_t nr = slots * (p < pieces.size() ? std::max(1u, pieces[p]) : 1u);
    for (uint32_t r = 1; r <= nr; ++r) {
      size_t to_tile;
      uint32_t to_step;
      if (p + 1 == phases.size() && r == nr) {
        to_tile = nt;
        to_step = 0;
      } else {
        locate(total * (phase_lo + (phase_hi - phase_lo) * (double)r / (double)nr), to_tile, to_step);
      }
      if (to_tile < cur_tile || (to_tile == cur_tile && to_step <= cur_step)) continue;   // nothing left for this run
      unit_start.push_back((uint32_t)units.size());
      while (cur_tile < to_tile || (cur_tile == to_tile && cur_step < to_step)) {
        const uint32_t hi = cur_tile < to_tile ? steps_total : to_step;
        const uint32_t end = std::min(hi, cur_step + MFMA_MAX_STEPS);   // f32 counts stay exact
        MfmaUnit u;
        std::memset(&u, 0, sizeof(u));
        u.row_tile = tiles[cur_tile].row_tile;
        u.col_tile = tiles[cur_tile].col_tile;
        u.step_lo = cur_step;
        u.step_hi = end;
        u.blocks = tiles[cur_tile].blocks;
        u.top_blocks = tiles[cur_tile].top_blocks;
        u.tile = tiles[cur_tile].tile;
        u.part = tiles[cur_tile].part;
        units.push_back(u);
        cur_step = end;
        if (cur_step == steps_total) {
          ++cur_tile;
          cur_step = 0;
        }
      }
    }
    phase_lo = phase_hi;
  }
  unit_start.push_back((uint32_t)units.size());
}

// Runs of units for one batch (built once per batch and row count; `target` marks
// the form in the cache test).  One workgroup per run, dispatched in run order as
// workgroup slots come free.
static obnsl_status build_runs(Batch* b, const Plan* plan, uint32_t steps_total, uint32_t slots, uint32_t target) {
  if (b->units_steps == steps_total && b->units_target == target && b->nunits) return OBNSL_OK;
  std::vector<TileRef> tiles;
  tiles_of(b, 0, tiles);
  std::vector<MfmaUnit> units;
  std::vector<uint32_t> unit_start;
  cut_runs(tiles, steps_total, slots, env().run_phases, env().run_pieces, units, unit_start);
  OB_TRY(check_units({{b, plan->tile, plan->ncols_total}}, units, unit_start, steps_total));
  const uint32_t nruns = (uint32_t)unit_start.size() - 1;
  if (env().dump_units) dump_units(tiles.size(), units, steps_total, target, nruns);
  HIP_TRY(b->units.upload(units));
  HIP_TRY(b->unit_start.upload(unit_start));
  b->nunits = (uint32_t)units.size();
  b->nworkgroups = nruns;
  b->nruns = nruns;
  b->runs = true;
  b->units_steps = steps_total;
  b->units_target = target;
  b->units_split = units.size() > tiles.size();
  return OBNSL_OK;
}

// A top size of at most this many tiles is counted whole on every rank rather
// than in shards (one 128 x 128 matrix-core tile is sixteen 64 x 64 ones).
uint64_t whole_below(const Plan* p) { return p->mfma ? 4 : 64; }

// Units per matrix-core launch by number of tiles: one or two tiles, up to 32, more
// without and with tile images (OBNSL_WANT="a,b,c[,d]" overrides: tuning runs).
static uint32_t want_tier(int tier) {
  static const uint32_t dflt[4] = {64u, 384u, 896u, 4096u};
  return env().want[0] ? env().want[tier] : dflt[tier];
}

// How a launch over one batch is shaped: split launches of 128 x 128 tiles add into tile
// images (up to the image budget) instead of the count table itself -- not the launches of
// one or two tiles: their few units' atomics cost less than the scatter launch
// (OBNSL_NO_IMAGE: comparison runs) -- and how many units it is cut into:
// the matrix-core kernel runs two 128 x 128 workgroups per CU: a few rounds of
// 512 are enough, and every extra split adds the whole table in atomics
// (measured on the 37-variable workload, whole pass: one tile cut into ~64 units -- 16
// units of ~64 steps when it runs beside a launch of many tiles, whose workgroup slots
// it would take --, 17 tiles into ~384, more tiles into ~4096 units, at least two per
// tile and up to 12288 of ~96 steps (70 variables, 10^7 rows: 306 tiles, +5 % from 4096
// to 8192..16384 units; 37 variables, five-variable tables: 759 tiles, +4 % from 4096 to
// 8192) when the units add into tile images, ~896 when they add into the
// table with scattered atomics: short units pay their prologue and epilogue, long ones
// leave CUs idle at the end of the launch).
struct LaunchShape {
  uint64_t img_cells = 0;
  bool image_ok = false;
  uint32_t want = 0, slots = 0, kcn = 1;
  bool runs = false;   // short launch of many 128 x 128 tiles: runs of units per workgroup
};
static LaunchShape launch_shape(const obnsl_data* d, const Plan* plan, const Batch* b, uint32_t steps_total) {
  LaunchShape sh;
  sh.img_cells = (uint64_t)b->nitems * plan->tile * plan->tile;
  sh.image_ok = plan->mfma && !env().no_image && b->nitems > 2 &&
                sh.img_cells * sizeof(uint32_t) <= env().image_max_bytes;
  sh.want = env().target_wgs ? env().target_wgs
            : !plan->mfma ? 8192u
            : b->nitems <= 2 ? (d->beside_large ? std::min(want_tier(0), std::max(16u, steps_total / 64))
                                                : want_tier(0))
            : b->nitems <= 32 ? want_tier(1)
            : sh.image_ok ? std::max({want_tier(3), 2 * b->nitems,
                                      (uint32_t)std::min<uint64_t>((uint64_t)b->nitems * steps_total / 96, 12288)})
                          : want_tier(2);
  sh.kcn = plan->kc == 0 ? 1 : plan->kc;  // kc == 0: the single ONES column
  sh.slots = d->num_cus * ((plan->kr + sh.kcn <= 5) ? 2u : 1u);
  // Runs of units: short launches of 128 x 128 tiles, cut along the contraction anyway, of
  // more than 32 tiles.  "Short": at most 2048 contraction steps per workgroup slot --
  // there the gap between one workgroup's end and its successor's start, ~10 us per
  // unit, is what the form saves; longer launches are as fast with one unit per
  // workgroup.
  const uint64_t tile_steps = (uint64_t)b->nitems * steps_total;
  // (the persistent kernels address the bitmaps with 32-bit byte offsets)
  const bool narrow = ((uint64_t)d->total_planes + 1) * d->wp * sizeof(uint32_t) <= 0xffffffffull && !env().wide_loads;
  sh.runs = env().runs && narrow && plan->mfma && plan->tile == 128 && sh.image_ok && b->nitems > 32 && b->nitems < sh.want &&
            tile_steps >= 8ull * sh.slots &&
            tile_steps <= (uint64_t)env().runs_max_steps * sh.slots;
  return sh;
}

static obnsl_status next_event_pair(obnsl_data* d, uint32_t k, std::pair<hipEvent_t, hipEvent_t>** out) {
  if (d->ev_used >= 4096) OB_TRY(resolve_events(d));   // bound the pool: fold what has been recorded so far
  if (d->ev_used == d->ev.size()) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    d->ev.push_back({e0, e1});
    d->ev_size.push_back(0);
  }
  d->ev_size[d->ev_used] = k;
  *out = &d->ev[d->ev_used++];
  return OBNSL_OK;
}

// Tile images of a batch, cleared for the launch that is about to add into them (the
// IMG_TAIL words behind them, the run queue's counter, with them).
static obnsl_status prepare_images(obnsl_data* d, Plan* plan, Batch* b, uint64_t img_cells, uint32_t* table) {
  HIP_TRY(b->img.ensure(img_cells + IMG_TAIL));
  if (b->img.n < img_cells + IMG_TAIL)
    return fail(OBNSL_ERR_RUNTIME, "internal: tile images smaller than the launch's tiles");
  if (b->img_zeroed_epoch != d->epoch)
    HIP_TRY(hipMemsetAsync(b->img.p, 0, (img_cells + IMG_TAIL) * sizeof(uint32_t), d->count_stream()));
  b->img_zeroed_epoch = 0;   // about to be added into
  b->img_wants_zero = true;
  if (table == plan->red.p) plan->wants_zero = false;  // every cell is stored, none added into
  return OBNSL_OK;
}

// One bit-GEMM launch (timed with an event pair) filling `counts` for one batch.
obnsl_status launch_count(obnsl_data* d, Plan* plan, Batch* b, uint32_t* counts, bool prezeroed) {
  if (b->nitems == 0) return OBNSL_OK;
  if (!counts) return fail(OBNSL_ERR_RUNTIME, "internal: count launch without a table");
  const uint32_t steps_total = d->wp / KCW;
  const LaunchShape sh = launch_shape(d, plan, b, steps_total);
  // split the contraction when there are too few tiles to fill the chip
  // (workgroups of equal length run in rounds of 2 per CU; many short ones keep
  // the last round from idling most of the chip)
  uint32_t ksplit = 1;
  const uint32_t max_split = std::max<uint32_t>(1, steps_total / env().min_sps);
  if (b->nitems < sh.want) ksplit = std::min<uint32_t>((sh.want + b->nitems - 1) / b->nitems, max_split);
  uint32_t sps = (steps_total + ksplit - 1) / ksplit;
  if (plan->mfma) sps = std::min(sps, MFMA_MAX_STEPS);  // f32 accumulators stay exact below 2^24
  ksplit = (steps_total + sps - 1) / sps;
  if (plan->mfma) {
    if (!gemm_mfma_has(plan->kr, sh.kcn, plan->tile)) return fail(OBNSL_ERR_RUNTIME, "internal: no bit-GEMM instance");
    if (sh.runs)
      OB_TRY(build_runs(b, plan, steps_total, sh.slots, 0x80000000u | sh.slots));
    else
      OB_TRY(build_units(b, plan, steps_total, std::max(sh.want, b->nitems)));
  }
  int flags = ((plan->mfma ? b->units_split : ksplit > 1) ? GEMM_ATOMIC : 0) |
              (env().no_jskip ? GEMM_ALL_BLOCKS : 0) | (env().no_2x2 ? GEMM_NO_2X2 : 0);
  if (((uint64_t)d->total_planes + 1) * d->wp * sizeof(uint32_t) <= 0xffffffffull && !env().wide_loads)
    flags |= GEMM_NARROW;
  const bool image = sh.image_ok && (flags & GEMM_ATOMIC);
  if (plan->mfma && b->runs && !image) return fail(OBNSL_ERR_RUNTIME, "internal: runs of units without tile images");
  uint32_t* table = counts;
  if (image) {
    OB_TRY(prepare_images(d, plan, b, sh.img_cells, table));
    flags |= GEMM_IMAGE;
    counts = b->img.p;
  } else {
    if ((flags & GEMM_ATOMIC) && !prezeroed)
      HIP_TRY(hipMemsetAsync(counts, 0, b->ncells * sizeof(uint32_t), d->count_stream()));
    if (counts == plan->red.p) plan->wants_zero = (flags & GEMM_ATOMIC) != 0;
  }
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (d->timed_now()) {
    OB_TRY(next_event_pair(d, plan->k, &ev));
    HIP_TRY(hipEventRecord(ev->first, d->count_stream()));
  }
  GemmOperands op;
  op.planes = d->planes.p;
  op.wp = d->wp;
  op.rows = b->rows.p;
  op.nrows = b->nrows;
  op.cols = plan->cols.p;
  op.ncols_total = plan->ncols_total;
  if (plan->mfma) {
    MfmaLaunch l;
    std::memset(&l, 0, sizeof(l));
    l.units = b->units.p;
    l.unit_start = b->runs ? b->unit_start.p : nullptr;
    l.nruns = b->runs ? b->nruns : 0;
    l.queue = b->runs ? b->img.p + sh.img_cells : nullptr;
    l.nworkgroups = b->nworkgroups;
#ifdef OBNSL_PROBES
    if (d->trace_want && plan->k == d->trace_size) {
      HIP_TRY(d->trace.ensure(4 * (size_t)b->nworkgroups));
      d->trace_workgroups = b->nworkgroups;
      l.trace = d->trace.p;
    }
#endif
    launch_gemm_mfma(plan->kr, sh.kcn, plan->tile, d->count_stream(), op, l, counts, flags, env().probe);
    uint64_t* info = d->launch_info[plan->k];
    info[0] = b->nitems;
    info[1] = b->nunits;
    info[2] = b->nworkgroups;
    info[3] = 0;
    for (size_t i = 0; i < b->h_item_blocks.size(); ++i) {
      const uint32_t blk = b->h_item_blocks[i];
      // (runs of units contract only the lower 64 rows of a full-width tile whose upper rows reach no column)
      const bool lower_only = b->runs && plan->tile == 128 && blk == 4 && b->h_item_top[i] == 0;
      info[3] += (uint64_t)(lower_only ? 64 : plan->tile) * 32 * blk;
    }
    info[4] = b->ncells;
    info[5] = 1;
  } else {
    launch_gemm_valu(plan->kr, sh.kcn, plan->tile, d->count_stream(), op, b->items.p, b->nitems, b->nitems * ksplit,
                     counts, steps_total, sps, flags);
  }
  HIP_TRY(hipGetLastError());
  if (image) {
    launch_image_scatter(plan->tile, d->count_stream(), b->img.p, b->items.p, b->nitems, b->rows.p, b->nrows, table);
    HIP_TRY(hipGetLastError());
  }
  if (ev) HIP_TRY(hipEventRecord(ev->second, d->count_stream()));
  return OBNSL_OK;
}

// Whether the counting of one pass -- the subset sizes of `parts`, largest first -- can be
// ONE launch of persistent workgroups: a short matrix-core launch (see launch_shape)
// whose kernel form also serves the smaller sizes (fewer variables per row and column).
bool can_count_together(const obnsl_data* d, const std::vector<CountPart>& parts) {
  if (!env().together || parts.empty() || parts.size() > (size_t)GEMM_MAX_PARTS) return false;
  const uint32_t steps_total = d->wp / KCW;
  const CountPart& top = parts[0];
  if (!launch_shape(d, top.plan, top.b, steps_total).runs) return false;
  for (const CountPart& p : parts)
    if (!p.plan->mfma || p.plan->tile != 128 || p.plan->kr > top.plan->kr || p.plan->kc > top.plan->kc ||
        p.b->nitems == 0)
      return false;
  return true;
}

// One launch that counts several subset sizes: parts[0] the largest (its kernel form),
// the others smaller.  As many persistent workgroups as the chip holds take the first
// phase of runs and then the later ones from a queue; nothing else needs to run beside
// the launch, so no other launch competes with it for workgroup slots (two of its
// workgroups fill a CU) and nothing waits across streams.  Parts of more than two tiles
// add into tile images, the others into their (cleared) count tables.
obnsl_status launch_count_together(obnsl_data* d, std::vector<CountPart>& parts) {
  if (!can_count_together(d, parts)) return fail(OBNSL_ERR_RUNTIME, "internal: launch over several sizes refused");
  const uint32_t steps_total = d->wp / KCW;
  const LaunchShape top_shape = launch_shape(d, parts[0].plan, parts[0].b, steps_total);
  // the cached work list of this combination of batches
  TogetherLaunch* tl = nullptr;
  for (auto& c : d->together)
    if (c->steps == steps_total && c->batches.size() == parts.size() &&
        std::equal(parts.begin(), parts.end(), c->batches.begin(),
                   [](const CountPart& p, const Batch* b) { return p.b == b; }))
      tl = c.get();
  if (!tl) {
    auto c = std::make_unique<TogetherLaunch>();
    c->steps = steps_total;
    std::vector<TileRef> tiles;
    std::vector<PartShape> shapes;
    for (size_t p = 0; p < parts.size(); ++p) {
      c->batches.push_back(parts[p].b);
      tiles_of(parts[p].b, (uint32_t)p, tiles);
      shapes.push_back({parts[p].b, parts[p].plan->tile, parts[p].plan->ncols_total});
    }
    // the line: tiles of every part, those with the most column blocks first
    std::stable_sort(tiles.begin(), tiles.end(), [](const TileRef& x, const TileRef& y) { return x.blocks > y.blocks; });
    std::vector<MfmaUnit> units;
    std::vector<uint32_t> unit_start;
    // The phases after the first even out its runs' times (+- 13 %) and cost every workgroup
    // one more prologue and flush each: worth it for long runs only.  Measured on the
    // 37-variable pass over 1/1, 1/2, 1/4 and 1/8 of 10^6 rows (147, 74, 37 and 19 steps per
    // slot), launch in ms: phases (65, 15, 10, 10) 0.412 0.242 0.169 0.142, (70, 30) - 0.235 0.133
    // 0.089, one phase - 0.276 0.145 0.087.  The default, (74, 26) with the second phase in
    // runs of half its slot share: 0.396 and 0.220 for the first two (first shares from 73
    // to 75: 0.395-0.398; 76 and more: 0.409-0.428; (72, 14, 14) with the last phase in
    // half-length runs 0.400; three runs per slot in the last phase, or four phases:
    // 0.416-0.429): every run costs its workgroup ~7 us of prologue and flush, every run
    // too long a tail.
    std::vector<double> phases = env().together_phases;
    std::vector<uint32_t> pieces = env().together_pieces;
    if (!env().together_phases_set) {
      uint64_t tile_steps = 0;
      for (const TileRef& t : tiles) tile_steps += steps_total;
      const uint64_t per_slot = tile_steps / std::max(1u, top_shape.slots);
      if (per_slot < 28) {
        phases = {1.0};
        pieces = {1};
      } else if (per_slot < 56) {
        phases = {0.7, 0.3};
        pieces = {1, 1};
      }
    }
    cut_runs(tiles, steps_total, top_shape.slots, phases, pieces, units, unit_start);
    OB_TRY(check_units(shapes, units, unit_start, steps_total));
    c->nruns = (uint32_t)unit_start.size() - 1;
    c->nworkgroups = std::min(c->nruns, top_shape.slots);
    if (env().dump_units) dump_units(tiles.size(), units, steps_total, 0, c->nworkgroups);
    HIP_TRY(c->units.upload(units));
    HIP_TRY(c->unit_start.upload(unit_start));
    tl = c.get();
    d->together.push_back(std::move(c));
  }
  // every part's destination: tile images (cleared) or its count table (cleared)
  GemmOperands op;
  std::memset(&op, 0, sizeof(op));
  MfmaLaunch l;
  std::memset(&l, 0, sizeof(l));
  int flags = GEMM_ATOMIC | (env().no_jskip ? GEMM_ALL_BLOCKS : 0) | (env().no_2x2 ? GEMM_NO_2X2 : 0);
  uint32_t* top_counts = nullptr;
  for (size_t p = 0; p < parts.size(); ++p) {
    CountPart& cp = parts[p];
    const LaunchShape sh = launch_shape(d, cp.plan, cp.b, steps_total);
    // (parts of one or two tiles add into their tables; measured on the 37-variable pass:
    // with the 17 tiles of the second-largest size adding into their table too -- one
    // scatter launch fewer -- the launch takes 0.450 ms instead of 0.422)
    cp.image = sh.image_ok;
    uint32_t* dst = cp.table;
    if (cp.image) {
      OB_TRY(prepare_images(d, cp.plan, cp.b, sh.img_cells, cp.table));
      dst = cp.b->img.p;
    } else {
      if (!cp.prezeroed)
        HIP_TRY(hipMemsetAsync(cp.table, 0, cp.b->ncells * sizeof(uint32_t), d->count_stream()));
      if (cp.table == cp.plan->red.p) cp.plan->wants_zero = true;
    }
    if (p == 0) {
      if (!cp.image) return fail(OBNSL_ERR_RUNTIME, "internal: launch over several sizes without tile images");
      op.planes = d->planes.p;
      op.wp = d->wp;
      op.rows = cp.b->rows.p;
      op.nrows = cp.b->nrows;
      op.cols = cp.plan->cols.p;
      op.ncols_total = cp.plan->ncols_total;
      top_counts = dst;
      flags |= GEMM_IMAGE;
      l.queue = cp.b->img.p + sh.img_cells;
    } else {
      GemmPart& e = l.extra.p[p - 1];
      e.rows = cp.b->rows.p;
      e.cols = cp.plan->cols.p;
      e.counts = dst;
      e.nrows = cp.b->nrows;
      e.ncols_total = cp.plan->ncols_total;
      e.image = cp.image ? 1 : 0;
    }
  }
  l.units = tl->units.p;
  l.unit_start = tl->unit_start.p;
  l.nruns = tl->nruns;
  l.nworkgroups = tl->nworkgroups;
#ifdef OBNSL_PROBES
  if (d->trace_want && parts[0].plan->k == d->trace_size) {
    HIP_TRY(d->trace.ensure(4 * (size_t)l.nworkgroups));
    d->trace_workgroups = l.nworkgroups;
    l.trace = d->trace.p;
  }
#endif
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (d->timed_now()) {
    OB_TRY(next_event_pair(d, parts[0].plan->k, &ev));
    HIP_TRY(hipEventRecord(ev->first, d->count_stream()));
  }
  launch_gemm_mfma(parts[0].plan->kr, top_shape.kcn, 128, d->count_stream(), op, l, top_counts, flags, env().probe);
  HIP_TRY(hipGetLastError());
  {
    uint64_t* info = d->launch_info[parts[0].plan->k];
    info[0] = info[3] = info[4] = 0;
    info[1] = tl->units.n;
    info[2] = tl->nworkgroups;
    info[5] = parts.size();
    for (const CountPart& cp : parts) {
      info[0] += cp.b->nitems;
      for (size_t i = 0; i < cp.b->h_item_blocks.size(); ++i) {
        const uint32_t blk = cp.b->h_item_blocks[i];
        const bool lower_only = cp.plan->tile == 128 && blk == 4 && cp.b->h_item_top[i] == 0;
        info[3] += (uint64_t)(lower_only ? 64 : cp.plan->tile) * 32 * blk;
      }
      info[4] += cp.b->ncells;
    }
  }
  {
    // the finished tile images of every part that used them, into their count tables
    ScatterParts sp;
    std::memset(&sp, 0, sizeof(sp));
    for (CountPart& cp : parts)
      if (cp.image) {
        ScatterPart& x = sp.p[sp.nparts];
        x.images = cp.b->img.p;
        x.items = cp.b->items.p;
        x.rows = cp.b->rows.p;
        x.table = cp.table;
        x.nrows = cp.b->nrows;
        sp.first[sp.nparts + 1] = sp.first[sp.nparts] + cp.b->nitems * (128 / 8);
        ++sp.nparts;
      }
    if (sp.nparts) {
      launch_image_scatter_parts(d->count_stream(), sp);
      HIP_TRY(hipGetLastError());
    }
  }
  if (ev) HIP_TRY(hipEventRecord(ev->second, d->count_stream()));
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
  if (d->rows_sharded() && !d->counting_rows)
    return fail(OBNSL_ERR_INVALID_ARGUMENT,
                "this data set holds a slice of the rows: count with obnsl_precount_rows and sum the "
                "count tables over the ranks before scoring");
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
