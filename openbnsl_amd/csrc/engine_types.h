// Plain structs and constants shared by the kernels and the host code that
// fills them (every translation unit of libobnsl_engine.so); not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "obnsl_engine.h"

namespace obnsl_internal {

constexpr int KMAX = OBNSL_MAX_SUBSET;  // largest subset size (per-row counting path)
constexpr int KFAST = 8;                // largest subset size of the bit-plane contraction (2 KSIDE)
constexpr int KSIDE = 4;                // largest kr / kc
constexpr int KCW = 32;                 // 32-bit words of the contraction per LDS slab
constexpr int LDS_STRIDE = KCW + 4;     // padded slab row (words): conflict-free b128 reads
constexpr uint64_t BATCH_CELL_CAP = 1ull << 30;  // counters per batch (4 GiB)
constexpr uint64_t REDUCED_MAX_CELLS = 4096;     // largest full table the reduced-mode scorer expands in LDS
constexpr int LG_SMALL = 16;   // counts below this take lgamma(n + a) - lgamma(a) from a table (per prior a)
constexpr int MAX_DEVICES = 64;

constexpr int MFMA_TILE = 128;        // tile edge of the matrix-core kernel (256 for launches of many tiles)
constexpr uint64_t IMAGE_MAX_BYTES = 256ull << 20;  // tile images of one split launch (4096 tiles)
constexpr uint64_t BIG_TILE_FROM = 16384;  // 128 x 128 tiles from which a launch takes 256 x 256 tiles
constexpr uint64_t IMG_TAIL = 64;     // words behind a launch's tile images (the run queue's counter)
constexpr uint32_t MFMA_MAX_STEPS = (1u << 24) / (KCW * 32);  // samples per workgroup < 2^24

constexpr uint32_t HIST_LDS_CELLS = 16384;   // 64 KiB of LDS counters
constexpr uint32_t HIST_ROWS_PER_ITER = 256 * 16;
constexpr uint32_t BEST_LDS_VARS = 2048;

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
struct OneTable {       // a single explicit subset, passed by value (no upload per call)
  uint32_t pbase[KMAX];
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

// A work unit of the matrix-core launch: one tile and a run of contraction
// steps.  The host cuts every tile into runs of (nearly) equal cost -- steps x
// column blocks -- so that all workgroups of a launch take the same time.
struct MfmaUnit {
  uint32_t row_tile, col_tile, step_lo, step_hi;
  uint32_t blocks;   // 32-column blocks of the tile that hold a valid column (>= 1)
  uint32_t tile;     // index of the tile in its part's tile list (its image)
  uint32_t part;     // part of the launch (0 unless one launch counts several subset sizes)
  uint32_t top_blocks;  // 32-column blocks that the tile's first 64 rows reach (<= blocks; 0: none of them)
};

// Operands of one contraction launch.
struct GemmOperands {
  const uint32_t* planes;
  uint32_t wp;
  const RowDesc* rows;
  uint32_t nrows;
  const ColDesc* cols;
  uint32_t ncols_total;
};

// One launch can count several subset sizes ("parts"): the rows, columns and count
// table (or tile images) of each part after the first, which the kernel's own
// arguments describe.  Smaller sizes have fewer variables per row / column than the
// kernel's KR / KC; their descriptors name the all-ones bitmap in the unused places.
constexpr int GEMM_MAX_PARTS = 5;
struct GemmPart {
  const RowDesc* rows;
  const ColDesc* cols;
  uint32_t* counts;
  uint32_t nrows, ncols_total;
  int image;    // counts = tile images of the part (else its count table, added into)
  int pad_;
};
struct GemmParts {
  GemmPart p[GEMM_MAX_PARTS - 1];
};

// The image parts of a launch over several subset sizes, for their one scatter launch.
struct ScatterPart {
  const uint32_t* images;
  const uint2* items;
  const RowDesc* rows;
  uint32_t* table;
  uint32_t nrows, pad_;
};
struct ScatterParts {
  ScatterPart p[GEMM_MAX_PARTS];
  uint32_t first[GEMM_MAX_PARTS + 1];   // first workgroup of each part; first[nparts] = the grid
  uint32_t nparts;
};

// Work list of one matrix-core launch.  unit_start == nullptr: workgroup w works on
// units[w] (nworkgroups units).  Otherwise run r is units[unit_start[r] ..
// unit_start[r + 1]), r < nruns; nworkgroups persistent workgroups start with runs
// 0 .. nworkgroups - 1 and take the others from the counter *queue (zero at launch).
struct MfmaLaunch {
  const MfmaUnit* units;
  const uint32_t* unit_start;
  uint32_t nruns;
  uint32_t* queue;
  uint32_t nworkgroups;
  GemmParts extra;   // parts 1.. of a launch over several subset sizes (runs only)
  uint64_t* trace;   // probe builds only
};

// Launch flags of the contraction kernels.
enum GemmFlags : int {
  GEMM_ATOMIC = 1,      // several units per tile: partial counts are added, not stored
  GEMM_ALL_BLOCKS = 2,  // compute every column block of an edge tile (OBNSL_NO_JSKIP; same results)
  GEMM_NO_2X2 = 4,      // full tiles in the 1 x 4 wave layout (OBNSL_NO_2X2; same results)
  GEMM_IMAGE = 8,       // `counts` is the launch's array of tile images
  GEMM_NARROW = 16,     // the bitmaps stay within 4 GB: kernels that address them with 32-bit byte offsets
};

struct SizeView {            // where the reduced tables of one subset size live
  const uint32_t* red;       // counters
  const uint64_t* rowoff;    // first counter of a row group, by colex rank of its kr-subset
  const ColGroupDesc* cgs;   // column groups in colex order (col_before in reduced cells)
  uint32_t kr, kc;
};
struct ReducedViews {
  SizeView v[KFAST + 1];
};

// The variables of every size-k subset by colex rank (what a thread otherwise finds with k
// binary searches of dependent loads) are kept as a table while it stays within this.
constexpr uint64_t SUBSET_VARS_MAX_BYTES = 64ull << 20;

// The fused scoring launches (every subset size at once) serve sizes up to this one.
constexpr int FUSED_KMAX = 5;
// Words of a table's record (see the scorer): rank, cardinalities, one start per pattern.
constexpr uint32_t table_record_words(int k) { return (uint32_t)((4 + (1 << k) - 1 + 3) & ~3); }
struct FunctionalSizeArgs {       // one subset size of functional_all_kernel
  const RowGroupDesc* rgs;
  uint32_t* recs;                 // per-table records (null: none)
  int rec_mode;                   // 0: descriptors only, 1: and write the records, 2: read the records
  const uint32_t* wg_rg;          // row group of each workgroup's first table
  double* G;                      // out: functional by colex rank
  uint64_t ntables;
  uint32_t tables_per_wg, parts, e_stride;
  uint32_t block_start;           // first workgroup of this size
};
struct FunctionalAllArgs {
  ReducedViews views;
  FunctionalSizeArgs size[FUSED_KMAX + 1];   // by subset size, 1..kmax
  uint32_t kmax;
  const uint32_t* card;
  const uint64_t* binom;
  uint32_t num_rows;
  int kind;
  double ess;
  const double* lg_alpha;
  unsigned long long* best_key;   // keys [0, n), ids [n, 2 n): reset by workgroup 0 (null: none)
  uint32_t n;
};
struct ScoresAllArgs {            // scores_best_kernel
  const double* G[FUSED_KMAX + 1];   // G[k]: functional of size k by colex rank; G[0]: unused (g0)
  double* scores[FUSED_KMAX + 1];    // scores[k]: table of subset size k
  const uint8_t* mask[FUSED_KMAX + 1];  // mask[k][rank] != 0: this shard scores the table (null: all)
  const uint16_t* vars[FUSED_KMAX + 1]; // vars[k][rank * k + j]: the subset's variables (null: unranked per thread)
  uint64_t nsubsets[FUSED_KMAX + 1];
  uint32_t block_start[FUSED_KMAX + 2];
  uint32_t kmax;
  double g0;                      // functional of the empty table
  const uint32_t* card;
  const uint64_t* binom;
  uint32_t n, num_rows;
  int kind;
  unsigned long long* best_key;   // per-node maximum of the ordered scores (null: not kept)
};

struct BestArgs {                 // every subset size in one launch
  const double* scores[KMAX];     // scores[k-1]: table of subset size k
  const uint16_t* vars[KMAX];     // vars[k-1][rank * k + j]: the subset's variables (null: unranked per thread)
  uint64_t nsubsets[KMAX];
  uint32_t block_start[KMAX + 1]; // workgroups [block_start[k-1], block_start[k]) serve size k
  uint32_t kmax;
};

struct DpScores {
  const double* tbl[KMAX];  // tbl[k-1][rank * k + i]
  uint32_t max_scored;      // max_parents + 1
};

}  // namespace obnsl_internal
