// Shared between the translation units of libobnsl_engine.so; not installed.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "obnsl_engine.h"

namespace obnsl_internal {

// Records `msg` as the calling thread's last error and returns `st`.
obnsl_status fail(obnsl_status st, const std::string& msg);
const char* last_error();

// Every environment switch of the library, read once per process (env.cpp holds
// the table with each variable's name and meaning).  None of them changes a
// result: they choose between kernels and launch shapes that produce the same
// tables and scores.
struct Env {
  bool gemm_valu = false;         // OBNSL_GEMM=valu: vector-ALU contraction instead of the matrix cores
  bool serial = false;            // OBNSL_SERIAL: every launch on the caller's stream
  int valu_tile = 64;             // OBNSL_TILE=128: tile edge of the vector-ALU kernel
  int mfma_tile = 0;              // OBNSL_MFMA_TILE=128|256: force the matrix-core tile edge (0: by launch size)
  uint64_t big_tile_from = 0;     // OBNSL_BIG_TILE_FROM: 128 x 128 tiles from which a launch takes 256 x 256
  bool dump_units = false;        // OBNSL_DUMP_UNITS: print every launch's unit decomposition to stderr
  uint32_t want[4] = {0, 0, 0, 0};  // OBNSL_WANT=a,b,c[,d]: units per launch by number of tiles
  uint32_t target_wgs = 0;        // OBNSL_TARGET_WGS: units per launch, every launch
  bool no_image = false;          // OBNSL_NO_IMAGE: split launches add into the count table itself
  uint64_t image_max_bytes = 0;   // OBNSL_IMAGE_MB: tile images per launch
  uint32_t min_sps = 8;           // OBNSL_MIN_SPS: fewest contraction steps per split (vector-ALU kernel)
  bool no_jskip = false;          // OBNSL_NO_JSKIP: compute the empty column blocks of edge tiles too
  bool no_2x2 = false;            // OBNSL_NO_2X2: full tiles in the 1 x 4 wave layout
  uint32_t tables_per_wg = 0;     // OBNSL_TABLES_PER_WG: tables per scorer workgroup
  bool no_reduced = false;        // OBNSL_NO_REDUCED: count every state (no inclusion-exclusion)
  bool force_hist = false;        // OBNSL_FORCE_HIST: per-row histogram path for every size
  bool force_bitgemm = false;     // OBNSL_FORCE_BITGEMM: bit-plane contraction wherever it is possible
  bool no_fused = false;          // OBNSL_NO_FUSED: one scorer launch per subset size, never the fused launches
  bool no_table_records = false;  // OBNSL_NO_TABLE_RECORDS: the fused scorer works every table out from the descriptors
  bool runs = true;               // OBNSL_NO_RUNS clears it: one work unit per workgroup, never persistent workgroups
  std::vector<double> run_phases; // OBNSL_RUN_PHASES=80,12,8: shares (percent) of a launch per phase of runs
  std::vector<uint32_t> run_pieces;  // OBNSL_RUN_PIECES=1,1,1: runs per workgroup slot in each phase
  bool together = true;           // OBNSL_NO_TOGETHER clears it: one count launch per subset size, never one for all
  std::vector<double> together_phases;    // OBNSL_TOGETHER_PHASES / _PIECES: phases of the launch over all sizes
  std::vector<uint32_t> together_pieces;
  bool wide_loads = false;                // OBNSL_WIDE_LOADS: 64-bit addresses for the bitmaps whatever their size
  bool together_phases_set = false;       // (given in the environment: not adapted to the launch's length)
  uint32_t runs_max_steps = 2048; // OBNSL_RUNS_MAX_STEPS: longest launch (contraction steps per slot) that takes the form
  int probe = 0;                  // OBNSL_PROBE=<bits>: timing probes; read only by builds with -DOBNSL_PROBES
};
const Env& env();

}  // namespace obnsl_internal

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return ::obnsl_internal::fail(OBNSL_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define OB_TRY(expr)                   \
  do {                                 \
    obnsl_status s_ = (expr);          \
    if (s_ != OBNSL_OK) return s_;     \
  } while (0)
