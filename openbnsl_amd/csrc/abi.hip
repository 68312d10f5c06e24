// abi.hip -- the C ABI of libobnsl_engine.so (include/obnsl_engine.h): argument
// checks, the reference's error messages, and the calls into planner.hip, passes.hip and the
// kernel launchers.  Each entry point's comment in the header names the reference
// interface it replaces.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>

#include "engine_state.h"

namespace obnsl_internal {
namespace {
thread_local std::string g_err;
}
obnsl_status fail(obnsl_status st, const std::string& msg) {
  g_err = msg;
  return st;
}
const char* last_error() { return g_err.c_str(); }
}  // namespace obnsl_internal

using namespace obnsl_internal;

extern "C" {

const char* obnsl_last_error(void) { return last_error(); }
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
    // once per device, with that device current: where this GPU holds the scratch
    // array of obnsl_best_parents_device_noalloc
    unsigned long long* unused = nullptr;
    OB_TRY(noalloc_scratch(device, &unused));
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;

  auto d = std::make_unique<obnsl_data>();
  d->device = device;
  d->num_cus = (uint32_t)cus;
  d->N = num_rows;
  d->total_rows = num_rows;
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
  launch_count_single(d->stream, d->planes.p, d->wp, k, ot, (uint32_t)cells, dcnt.p, d->ones_plane);
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
  launch_score_one(nullptr, sc.cnt.p, td, (int)kind, ess, sc.out.p);
  HIP_TRY(hipGetLastError());
  double res[KMAX];
  HIP_TRY(hipMemcpy(res, sc.out.p, k * sizeof(double), hipMemcpyDeviceToHost));
  *score_out = res[child_idx];
  return OBNSL_OK;
}

obnsl_status obnsl_score_all(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t k,
                             uint32_t shard, uint32_t num_shards, double* scores_out,
                             int out_on_device) {
  return score_all_impl(d, kind, ess, k, shard, num_shards, scores_out, out_on_device);
}

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

obnsl_status obnsl_data_set_total_rows(obnsl_data* d, uint64_t total_rows) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "data is null");
  if (total_rows < d->N) return fail(OBNSL_ERR_INVALID_ARGUMENT, "total_rows is smaller than the rows this data set holds");
  if (total_rows >= (1ull << 32)) return fail(OBNSL_ERR_UNSUPPORTED, "more than 2^32-1 rows");
  if (total_rows == d->total_rows) return OBNSL_OK;
  d->total_rows = total_rows;
  for (auto& kv : d->plans) {   // nothing worked out for the old total survives
    kv.second->g_epoch = 0;
    kv.second->counted_epoch = 0;
    kv.second->zeroed_epoch = 0;
    kv.second->awaits_sum = false;
  }
  return OBNSL_OK;
}

obnsl_status obnsl_precount_rows(obnsl_data* d, uint32_t kmax, uint32_t** d_counts_out, uint64_t* words_out) {
  return precount_rows_impl(d, kmax, d_counts_out, words_out);
}

obnsl_status obnsl_counts_mark_summed(obnsl_data* d, uint32_t k) {
  if (d && k == 0) {   // every size at once (after the one all-reduce of obnsl_precount_rows)
    for (auto& kv : d->plans)
      if (kv.second->counted_epoch == d->epoch) kv.second->awaits_sum = false;
    return OBNSL_OK;
  }
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
    launch_count_list(d->stream, dim3(gx, gy, gz), d->planes.p, d->wp, k, dpb.p + t0 * KMAX, dtd.p + t0, dcnt.p,
                      d->ones_plane);
  }
  HIP_TRY(hipGetLastError());
  if (m) {
    launch_score_tables(d->stream, dcnt.p, dtd.p, m, (int)kind, ess, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(scores_out, dout.p, m * k * sizeof(double), hipMemcpyDeviceToHost, d->stream));
  }
  HIP_TRY(hipStreamSynchronize(d->stream));
  return OBNSL_OK;
}

obnsl_status obnsl_best_parents(obnsl_data* d, const double* const* d_scores,
                                uint32_t max_parents, double* best_score,
                                uint32_t* best_size, uint64_t* best_rank) {
  if (!d || !d_scores || !best_score || !best_size || !best_rank)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = queue_best_parents(d, d_scores, max_parents, nullptr, false);
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
  obnsl_status st = queue_best_parents(d, d_scores, max_parents, nullptr, false);
  if (st != OBNSL_OK) return st;
  launch_best_decode(d->stream, d->best_key.p, d->best_key.p + d->n, d->n, d_out);
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
  // (looked up by obnsl_data_create on this device: no HIP call that could allocate here)
  unsigned long long* ptr = nullptr;
  OB_TRY(noalloc_scratch(d->device, &ptr));
  if (!ptr) return fail(OBNSL_ERR_RUNTIME, "no scratch address for this device");
  // one user of the array at a time per GPU: this call's kernels wait for the previous
  // user's (an event per device, created at set-up) and later users wait for these
  OB_TRY(set_device(d->device));
  OB_TRY(noalloc_user_begin(d->device, d->stream));
  // the arg-max kernels start from best_init_kernel's fill of [0, 2 n): what an
  // earlier (possibly larger) call left in the array is never read
  obnsl_status st = queue_best_parents(d, d_scores, max_parents, ptr, false);
  if (st == OBNSL_OK) {
    launch_best_decode(d->stream, ptr, ptr + d->n, d->n, d_out);
    if (hipGetLastError() != hipSuccess) st = fail(OBNSL_ERR_RUNTIME, "obnsl_best_parents_device_noalloc: launch failed");
  }
  const std::string msg = st == OBNSL_OK ? std::string() : std::string(last_error());
  const obnsl_status end = noalloc_user_end(d->device, d->stream);
  if (st != OBNSL_OK) return fail(st, msg);
  return end;
}

obnsl_status obnsl_score_pass(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t max_parents,
                              uint32_t shard, uint32_t num_shards, double* const* d_tables,
                              double* d_best_out) {
  return score_pass_impl(d, kind, ess, max_parents, shard, num_shards, d_tables, d_best_out, nullptr);
}

obnsl_status obnsl_combine_best_device(obnsl_data* d, const double* d_parts, uint32_t world,
                                       double* d_out) {
  if (!d || !d_parts || !d_out || world == 0) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  obnsl_status st = set_device(d->device);
  if (st != OBNSL_OK) return st;
  launch_best_combine(d->stream, d_parts, world, d->n, d_out);
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
    launch_dp_level(d->stream, grid, n, level, sc, d->d_binom.p, best_gs.p, best_ch.p, bls.p, bps.p);
  launch_dp_backtrack(d->stream, n, best_gs.p, best_ch.p, bps.p, parents.p, total.p);
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
  *path_out = !red ? OBNSL_PATH_HISTOGRAM : !env().gemm_valu ? OBNSL_PATH_MFMA : OBNSL_PATH_VALU;
  return OBNSL_OK;
}

obnsl_status obnsl_stream_read(const void* d_src, uint64_t bytes, void* hip_stream, uint32_t* d_sink) {
  if (!d_src || !d_sink || bytes < 16 || (reinterpret_cast<uintptr_t>(d_src) & 15))
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "obnsl_stream_read needs a 16-byte aligned buffer of at least 16 bytes");
  launch_stream_read(reinterpret_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(d_src), bytes / 16, d_sink);
  HIP_TRY(hipGetLastError());
  return OBNSL_OK;
}

obnsl_status obnsl_count_launch_info(obnsl_data* d, uint32_t k, uint64_t* info_out) {
  if (!d || !info_out || k == 0 || k > (uint32_t)KMAX)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "bad obnsl_count_launch_info arguments");
  for (int i = 0; i < 6; ++i) info_out[i] = d->launch_info[k][i];
  return OBNSL_OK;
}

obnsl_status obnsl_kernel_timing(obnsl_data* d, uint32_t every) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "data is null");
  d->timing_every = every;
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

#ifdef OBNSL_PROBES
// Timing probes of the probe build only (tools/; never declared in include/): record, for
// the next matrix-core launches of subset size k, per workgroup its begin and end on the
// 100 MHz counter, its hardware id and its XCC id; read them back after synchronising.
extern "C" obnsl_status obnsl_probe_trace_begin(obnsl_data* d, uint32_t k) {
  if (!d) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  d->trace_want = k != 0;
  d->trace_size = k;
  d->trace_workgroups = 0;
  return OBNSL_OK;
}
extern "C" obnsl_status obnsl_probe_trace_read(obnsl_data* d, uint64_t* out, uint64_t capacity_words,
                                               uint32_t* workgroups_out) {
  if (!d || !out || !workgroups_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  OB_TRY(set_device(d->device));
  HIP_TRY(hipDeviceSynchronize());
  *workgroups_out = d->trace_workgroups;
  const uint64_t words = 4ull * d->trace_workgroups;
  if (words > capacity_words) return fail(OBNSL_ERR_INVALID_ARGUMENT, "trace buffer too small");
  if (words) HIP_TRY(hipMemcpy(out, d->trace.p, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return OBNSL_OK;
}
#endif
