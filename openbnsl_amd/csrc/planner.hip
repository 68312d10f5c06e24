// planner.hip -- host side of the counting launches: the plan of a subset size (rows,
// columns, tiles and work units of its contraction; built once per data set and kept)
// and the launches that fill its count table -- one per size, or one of persistent
// workgroups for every size of a short pass.  passes.hip orders them into passes.  What
// they replace is the counting half of the scoring loop of exhaustive_search
// (src/structure_learning/exhaustive_search.cpp:47-73, contingency_table.h:52-55): one
// table per subset, subset after subset.
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

obnsl_status next_event_pair(obnsl_data* d, uint32_t k, std::pair<hipEvent_t, hipEvent_t>** out) {
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

}  // namespace obnsl_internal
