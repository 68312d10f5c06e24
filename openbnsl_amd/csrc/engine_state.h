// Host-side state of a data set (the row bitmaps, the launch plans of every subset
// size, their count tables) and the functions the translation units of
// libobnsl_engine.so share: planner.hip builds the plans and queues the count launches,
// passes.hip orders the passes, the kernel files export one launcher per kernel, abi.hip is
// the C ABI.  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "engine_internal.h"
#include "engine_types.h"

namespace obnsl_internal {

uint64_t binom_host(uint32_t n, uint32_t k);

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool borrowed = false;   // `p` points into another buffer's memory (not freed here)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { reset(); }
  void reset() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    n = 0;
    borrowed = false;
  }
  // Views `count` elements at `ptr` (memory another buffer owns and outlives this one).
  void borrow(T* ptr, size_t count) {
    reset();
    p = ptr;
    n = count;
    borrowed = true;
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
  std::vector<uint32_t> h_item_top;    // ... and that the item's first 64 rows reach (MfmaUnit::top_blocks)
  DevBuf<MfmaUnit> units;              // matrix-core launch: work units for `units_steps` steps
  DevBuf<uint32_t> unit_start;         // `runs`: workgroup w runs units [unit_start[w], unit_start[w + 1])
  uint32_t nunits = 0, units_steps = 0, units_target = 0;
  uint32_t nworkgroups = 0;            // workgroups of the launch (== nunits unless `runs`)
  bool runs = false;                   // persistent workgroups over runs of units (unit_start)
  uint32_t nruns = 0;
  bool units_split = false;            // some tile is cut into several units (atomics needed)
  std::vector<uint64_t> h_tbl_start;   // first table of every row group (host copy)
  std::vector<uint64_t> h_rank_part;   // ... its part of the colex rank, its number of column groups
  std::vector<uint32_t> h_ncolgroups;
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
  DevBuf<uint8_t> own_mask;     // sharded plan: 1 per colex rank of a table this shard scores
  // the scorer's per-table records (rank, cardinalities, where each pattern's counts start) and
  // the layouts of the count tables (sizes 1..k) they were worked out for
  DevBuf<uint32_t> recs;
  bool recs_ready = false;
  const void* recs_key[2 * (KFAST + 1)] = {nullptr};
  bool own_mask_built = false;
  uint64_t g_epoch = 0;
  int g_kind = -1;
  double g_ess = 0.0;
};

// One subset size's share of a count launch over several sizes.
struct CountPart {
  Plan* plan;
  Batch* b;
  uint32_t* table;     // where the counts go (the plan's table, or a slice of the whole plan's)
  bool prezeroed;      // the table is already clear for this launch
  bool image = false;  // (set by the launch) counted through tile images
};
// The cached work list of such a launch (runs of units over the tiles of all its batches).
struct TogetherLaunch {
  std::vector<const Batch*> batches;
  uint32_t steps = 0, nruns = 0, nworkgroups = 0;
  DevBuf<MfmaUnit> units;
  DevBuf<uint32_t> unit_start;
};

}  // namespace obnsl_internal

struct obnsl_data {
  int device = 0;
  uint32_t num_cus = 256;                             // compute units of the device
  hipStream_t stream = nullptr;
  uint64_t N = 0, stride = 0;
  // Rows of the data set the scores are about.  Larger than N when this handle holds one
  // rank's slice of the rows: its count tables are then partial sums (Plan::awaits_sum)
  // until the caller has added them up over the ranks.
  uint64_t total_rows = 0;
  bool rows_sharded() const { return total_rows != N; }
  bool counting_rows = false;   // inside obnsl_precount_rows (the one call that may count a slice)
  // rows-sharded: the count tables of sizes 1..arena_kmax, contiguous for one all-reduce
  obnsl_internal::DevBuf<uint32_t> arena;
  uint32_t arena_kmax = 0;
  uint64_t arena_words = 0;
  uint32_t n = 0, wp = 0, total_planes = 0, ones_plane = 0;
  std::vector<uint32_t> card, plane_base;
  obnsl_internal::DevBuf<uint8_t> codes;
  obnsl_internal::DevBuf<uint32_t> planes, d_card, d_plane_base, counts;
  obnsl_internal::DevBuf<unsigned long long> best_key;   // per-node arg-max scratch of obnsl_best_parents (+ one counter word)
  // scratch of the one-table / explicit-list entry points: grows, never shrinks
  obnsl_internal::DevBuf<uint32_t> one_cnt, list_pb, list_cnt;
  obnsl_internal::DevBuf<obnsl_internal::TableDesc> list_td;
  obnsl_internal::DevBuf<double> list_out;
  obnsl_internal::DevBuf<uint64_t> d_binom;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, bool>, std::unique_ptr<obnsl_internal::Plan>> plans;
  std::map<std::tuple<uint32_t, uint64_t, uint64_t>, std::vector<std::unique_ptr<obnsl_internal::HistBatch>>>
      hist_plans;
  uint64_t epoch = 1;                                 // bumped whenever the planes are rebuilt
  // obnsl_precount counts on two extra streams ("lanes", default priority): lane 0
  // the largest size, lane 1 the small sizes below the second largest
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t side_fork = nullptr, side_join[2] = {nullptr, nullptr};
  int count_lane = -1;                                // launch_count targets side[count_lane] (-1: `stream`)
  bool beside_large = false;                          // the launches of obnsl_precount run beside one of many tiles
  std::vector<obnsl_internal::Plan*> lane_plans[2];   // plans whose counts a lane is still producing
  std::vector<std::unique_ptr<obnsl_internal::TogetherLaunch>> together;  // launches over several subset sizes
  // by subset size k: the variables of every size-k subset, [rank * k + j] (built on first use)
  std::map<uint32_t, std::unique_ptr<obnsl_internal::DevBuf<uint16_t>>> subset_vars;
  // per ess: lgamma(ess / c) for c = 1..REDUCED_MAX_CELLS (index c), from the host's libm
  std::map<double, std::unique_ptr<obnsl_internal::DevBuf<double>>> lg_alpha;
  hipStream_t count_stream() const {
    return count_lane >= 0 && !obnsl_internal::env().serial ? side[count_lane] : stream;
  }
  // obnsl_kernel_timing: 0 = count launches are not timed (no event records on the stream: each
  // costs a few microseconds of a short pass), n = those of every n-th set of row bitmaps are
  uint32_t timing_every = 0;
  bool timed_now() const { return timing_every != 0 && epoch % timing_every == 0; }
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // around each timed count launch
  std::vector<uint32_t> ev_size;                      // subset size of each pending pair
  uint32_t ev_used = 0;                               // pending pairs
  double acc_ms[OBNSL_MAX_SUBSET + 1] = {0};          // per subset size, since the last reset
  // shape of the last matrix-core count launch timed under each subset size: tiles, work units,
  // workgroups, cells computed (tile rows x valid column blocks), cells asked for, subset sizes counted
  uint64_t launch_info[OBNSL_MAX_SUBSET + 1][6] = {{0}};
  uint32_t acc_launches[OBNSL_MAX_SUBSET + 1] = {0};
#ifdef OBNSL_PROBES
  // timing probe: per-workgroup (begin, end, hardware id, XCC id) of the launches of one subset size
  bool trace_want = false;
  uint32_t trace_size = 0, trace_workgroups = 0;
  obnsl_internal::DevBuf<uint64_t> trace;
#endif
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

namespace obnsl_internal {

// ---- planner.hip, passes.hip: plans, launches, passes and the data set's count tables
obnsl_status build_plan(obnsl_data* d, uint32_t k, uint32_t shard, uint32_t nshards, bool reduced, Plan** out);
obnsl_status resolve_events(obnsl_data* d);
bool use_mfma_gemm();
// The next pair of timing events of the data set (created on first use), booked under subset size k.
obnsl_status next_event_pair(obnsl_data* d, uint32_t k, std::pair<hipEvent_t, hipEvent_t>** out);
obnsl_status launch_count(obnsl_data* d, Plan* plan, Batch* b, uint32_t* counts, bool prezeroed = false);
bool can_count_together(const obnsl_data* d, const std::vector<CountPart>& parts);
obnsl_status launch_count_together(obnsl_data* d, std::vector<CountPart>& parts);
obnsl_status join_lane(obnsl_data* d, int lane);
obnsl_status join_side(obnsl_data* d);
obnsl_status ensure_counts(obnsl_data* d, Plan* plan);
obnsl_status run_functional(obnsl_data* d, Plan* plan, const ReducedViews& views, uint32_t k, uint64_t max_cells,
                            obnsl_score_kind kind, double ess, const double* gprev, double g0, double* scores);
obnsl_status run_hist_range(obnsl_data* d, uint32_t k, uint64_t lo, uint64_t hi, obnsl_score_kind kind, double ess,
                            double* d_scores);
obnsl_status choose_reduced(obnsl_data* d, uint32_t k, bool* use_reduced, uint64_t* max_cells_out);
obnsl_status check_kind(obnsl_score_kind kind);
obnsl_status set_device(int device);
obnsl_status run_pack(obnsl_data* d, uint32_t* bad_code = nullptr);
obnsl_status precount_impl(obnsl_data* d, uint32_t kmax, uint32_t shard, uint32_t num_shards, bool shard_lower,
                           uint32_t* sum_size_out, uint64_t* sum_cells_out);
obnsl_status whole_table(obnsl_data* d, uint32_t k, Plan** out);
obnsl_status precount_rows_impl(obnsl_data* d, uint32_t kmax, uint32_t** d_counts_out, uint64_t* words_out);
obnsl_status score_pass_impl(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t max_parents, uint32_t shard,
                             uint32_t num_shards, double* const* d_tables, double* d_best_out,
                             unsigned long long* keys);
obnsl_status queue_best_parents(obnsl_data* d, const double* const* d_scores, uint32_t max_parents,
                                unsigned long long* keys, bool skip_first_pass, double* decode_out = nullptr);
uint64_t whole_below(const Plan* p);
// The table of obnsl_data::subset_vars for size k, or null where it would be too large
// (or is not built yet and may_build is false: callers that must not allocate).
obnsl_status subset_vars_table(obnsl_data* d, uint32_t k, bool may_build, const uint16_t** out);
obnsl_status score_all_impl(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t k, uint32_t shard,
                            uint32_t num_shards, double* scores_out, int out_on_device);

// ---- one launcher per kernel (each queues on `s`; the caller checks hipGetLastError)
// gemm_mfma.hip
void launch_gemm_mfma(uint32_t kr, uint32_t kc, uint32_t tile, hipStream_t s, const GemmOperands& op,
                      const MfmaLaunch& l, uint32_t* counts, int flags, int probe);
void launch_image_scatter(uint32_t tile, hipStream_t s, const uint32_t* images, const uint2* items, uint32_t nitems,
                          const RowDesc* rows, uint32_t nrows, uint32_t* table);
void launch_image_scatter_parts(hipStream_t s, const ScatterParts& sp);
bool gemm_mfma_has(uint32_t kr, uint32_t kc, uint32_t tile);
// gemm_valu.hip
void launch_gemm_valu(uint32_t kr, uint32_t kc, uint32_t tile, hipStream_t s, const GemmOperands& op,
                      const uint2* items, uint32_t nitems, uint32_t grid, uint32_t* counts, uint32_t steps_total,
                      uint32_t steps_per_split, int flags);
// kernels_count.hip
void launch_pack(hipStream_t s, const uint8_t* codes, uint64_t N, uint64_t stride, uint32_t n, const uint32_t* card,
                 const uint32_t* plane_base, uint32_t* planes, uint32_t wp, uint32_t ones_plane, uint32_t* bad_code,
                 const ZeroList& zl);
void launch_stream_read(hipStream_t s, const uint4* src, uint64_t n16, uint32_t* sink);
void launch_count_single(hipStream_t s, const uint32_t* planes, uint32_t wp, uint32_t k, const OneTable& ot,
                         uint32_t cells, uint32_t* counts, uint32_t ones_plane);
void launch_count_list(hipStream_t s, dim3 grid, const uint32_t* planes, uint32_t wp, uint32_t k,
                       const uint32_t* pbase_all, const TableDesc* tds, uint32_t* counts, uint32_t ones_plane);
void launch_count_hist(hipStream_t s, uint32_t grid, size_t lds_bytes, const uint8_t* codes, uint64_t stride,
                       uint64_t num_rows, uint32_t n, const uint32_t* card, const uint64_t* binom, uint32_t k,
                       uint64_t rank0, uint32_t nchunks, uint64_t rows_per_chunk, const uint64_t* tbl_off,
                       uint32_t* counts);
// kernels_score.hip
void launch_score_plan(hipStream_t s, uint32_t grid, const uint32_t* counts, const RowGroupDesc* rgs, uint32_t nrgs,
                       const ColGroupDesc* cgs, const uint32_t* card, uint32_t kr, uint32_t kc, uint64_t ntables,
                       int kind, double ess, double* scores);
void launch_score_tables(hipStream_t s, const uint32_t* counts, const TableDesc* tables, uint64_t ntables, int kind,
                         double ess, double* scores);
void launch_score_one(hipStream_t s, const uint32_t* counts, const TableDesc& td, int kind, double ess,
                      double* scores);
struct ReducedScorerArgs {
  ReducedViews views;
  const RowGroupDesc* rgs;
  uint32_t nrgs;
  uint64_t ntables;
  uint32_t tables_per_wg;
  const uint32_t* card;
  const uint64_t* binom;
  uint32_t num_rows;
  int kind;
  double ess;
  double* G;
  const double* Gprev;
  double g0;
  double* scores;
  const double* lg_alpha;
  uint32_t e_stride;
  const uint32_t* wg_rg;
  unsigned long long* best_key;   // per-node running maximum of the ordered scores (null: not kept)
};
void launch_score_reduced(hipStream_t s, uint32_t k, uint32_t parts, uint32_t grid, const ReducedScorerArgs& a);
void launch_functional_all(hipStream_t s, uint32_t grid, size_t lds_bytes, const FunctionalAllArgs& a);
void launch_scores_best(hipStream_t s, const ScoresAllArgs& a);
// kernels_search.hip
void launch_best_init(hipStream_t s, unsigned long long* keys, unsigned long long* ids, uint32_t n);
void launch_best_parents(hipStream_t s, const BestArgs& args, uint32_t n, const uint64_t* binom,
                         unsigned long long* key, unsigned long long* id, int pass, double* decode_out = nullptr,
                         unsigned int* done_counter = nullptr);
void launch_subset_vars(hipStream_t s, const uint64_t* binom, uint32_t n, uint32_t k, uint64_t nsub, uint16_t* out);
void launch_best_decode(hipStream_t s, const unsigned long long* keys, const unsigned long long* ids, uint32_t n,
                        double* out);
void launch_best_combine(hipStream_t s, const double* parts, uint32_t world, uint32_t n, double* out);
void launch_dp_level(hipStream_t s, uint32_t grid, uint32_t n, uint32_t level, const DpScores& sc,
                     const uint64_t* binom, double* best_gs, signed char* best_ch, double* bls, uint32_t* bps);
void launch_dp_backtrack(hipStream_t s, uint32_t n, const double* best_gs, const signed char* best_ch,
                         const uint32_t* bps, uint64_t* parents, double* total);
// Device address of the per-GPU scratch array of obnsl_best_parents_device_noalloc on
// `device` (looked up once per device, with that device current).
obnsl_status noalloc_scratch(int device, unsigned long long** out);
obnsl_status noalloc_user_begin(int device, hipStream_t stream);
obnsl_status noalloc_user_end(int device, hipStream_t stream);

}  // namespace obnsl_internal
