// env.cpp -- the one place where libobnsl_engine.so reads its environment.
//
//   OBNSL_GEMM=valu          count with the vector-ALU and/popcount kernel instead of the matrix cores
//   OBNSL_SERIAL             queue every launch on the caller's stream (kernels do not overlap: profiling)
//   OBNSL_TILE=128           tile edge of the vector-ALU kernel (default 64)
//   OBNSL_MFMA_TILE=128|256  force the tile edge of the matrix-core kernel
//   OBNSL_BIG_TILE_FROM=n    number of 128 x 128 tiles from which a launch takes 256 x 256 tiles
//   OBNSL_DUMP_UNITS         print the unit decomposition of every matrix-core launch to stderr
//   OBNSL_WANT=a,b,c[,d]     units per matrix-core launch: one or two tiles, up to 32, more (without / with images)
//   OBNSL_TARGET_WGS=n       units per launch, whatever its size
//   OBNSL_NO_IMAGE           split launches add into the count table itself (scattered atomics)
//   OBNSL_IMAGE_MB=n         budget of tile images per launch (default 256)
//   OBNSL_MIN_SPS=n          fewest contraction steps per split of the vector-ALU kernel
//   OBNSL_NO_JSKIP           compute the column blocks of an edge tile that hold no valid column
//   OBNSL_NO_2X2             full tiles in the 1 x 4 wave layout
//   OBNSL_TABLES_PER_WG=n    tables per scorer workgroup
//   OBNSL_NO_REDUCED         count every state with the contraction (no inclusion-exclusion in the scorer)
//   OBNSL_FORCE_HIST         per-row histogram path for every subset size
//   OBNSL_FORCE_BITGEMM      bit-plane contraction wherever it can run
//   OBNSL_NO_FUSED           one scorer launch per subset size instead of the launches that serve every size
//   OBNSL_NO_TABLE_RECORDS   the fused scorer works each table out from the descriptors in every pass
//   OBNSL_NO_RUNS            one work unit per workgroup, never persistent workgroups over runs of units
//   OBNSL_RUN_PHASES=a,b,..  shares (percent) of such a launch per phase of runs (default 80,12,8)
//   OBNSL_RUN_PIECES=a,b,..  runs per workgroup slot in each phase (default 1,1,1)
//   OBNSL_NO_TOGETHER        one count launch per subset size, never one persistent launch for all sizes
//   OBNSL_TOGETHER_PHASES=.. / OBNSL_TOGETHER_PIECES=..  phases of that launch (default 72,28 / 1,2;
//                            70,30 / 1,1 and 100 for launches of fewer than 56 / 28 steps per workgroup slot)
//   OBNSL_RUNS_MAX_STEPS=n   longest launch, in contraction steps per workgroup slot, that takes the form
//   OBNSL_WIDE_LOADS         128 x 128 matrix-core kernels address the bitmaps with 64-bit pointers even where they
//                            stay within 4 GB (and then never as persistent workgroups)
//   OBNSL_PROBE=<bits>       timing probes; honoured only by a library built with -DOBNSL_PROBES
//
// All of them select among code paths that produce the same counts and scores.
// (OBNSL_GROUP_EXCHANGE, OBNSL_GROUP_SHARDING, OBNSL_ANNEAL_SCORES and OMP_NUM_THREADS are
// read where they apply: group.hip, anneal.cpp.)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "engine_internal.h"
#include "engine_types.h"

namespace obnsl_internal {

namespace {

bool flag(const char* name) { return std::getenv(name) != nullptr; }
long number(const char* name, long dflt) {
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = std::strtol(e, &end, 10);
  return (end && *end == '\0') ? v : dflt;
}

// "a,b,c" as shares (percent) that sum to 100, or empty
std::vector<double> shares(const char* name) {
  std::vector<double> v;
  const char* e = std::getenv(name);
  if (!e) return v;
  double sum = 0.0;
  const char* p = e;
  while (*p) {
    char* end = nullptr;
    const double f = std::strtod(p, &end);
    if (end == p || !(f > 0.0)) return {};
    v.push_back(f / 100.0);
    sum += f / 100.0;
    if (*end == ',') p = end + 1;
    else if (*end == '\0') break;
    else return {};
  }
  if (v.empty() || sum < 0.99 || sum > 1.01) return {};
  return v;
}
std::vector<uint32_t> counts_list(const char* name, size_t want) {
  std::vector<uint32_t> v;
  const char* e = std::getenv(name);
  if (!e) return v;
  const char* p = e;
  while (*p) {
    char* end = nullptr;
    const long f = std::strtol(p, &end, 10);
    if (end == p || f < 1 || f > 64) return {};
    v.push_back((uint32_t)f);
    if (*end == ',') p = end + 1;
    else if (*end == '\0') break;
    else return {};
  }
  if (v.size() != want) return {};
  return v;
}

Env read_env() {
  Env x;
  const char* g = std::getenv("OBNSL_GEMM");
  x.gemm_valu = g && std::strcmp(g, "valu") == 0;
  x.serial = flag("OBNSL_SERIAL");
  x.valu_tile = number("OBNSL_TILE", 64) == 128 ? 128 : 64;
  const long mt = number("OBNSL_MFMA_TILE", 0);
  x.mfma_tile = (mt == 128 || mt == 256) ? (int)mt : 0;
  x.big_tile_from = (uint64_t)number("OBNSL_BIG_TILE_FROM", (long)BIG_TILE_FROM);
  x.dump_units = flag("OBNSL_DUMP_UNITS");
  if (const char* e = std::getenv("OBNSL_WANT")) {
    unsigned a = 0, b = 0, c = 0, f = 0;
    const int got = std::sscanf(e, "%u,%u,%u,%u", &a, &b, &c, &f);
    if (got >= 3 && a && b && c) {
      x.want[0] = a;
      x.want[1] = b;
      x.want[2] = c;
      x.want[3] = got == 4 && f ? f : c;
    }
  }
  x.target_wgs = (uint32_t)number("OBNSL_TARGET_WGS", 0);
  x.no_image = flag("OBNSL_NO_IMAGE");
  x.image_max_bytes = (uint64_t)number("OBNSL_IMAGE_MB", (long)(IMAGE_MAX_BYTES >> 20)) << 20;
  x.min_sps = (uint32_t)number("OBNSL_MIN_SPS", 8);
  if (x.min_sps == 0) x.min_sps = 1;
  x.no_jskip = flag("OBNSL_NO_JSKIP");
  x.no_2x2 = flag("OBNSL_NO_2X2");
  x.tables_per_wg = (uint32_t)number("OBNSL_TABLES_PER_WG", 0);
  x.no_reduced = flag("OBNSL_NO_REDUCED");
  x.force_hist = flag("OBNSL_FORCE_HIST");
  x.force_bitgemm = flag("OBNSL_FORCE_BITGEMM");
  x.no_fused = flag("OBNSL_NO_FUSED");
  x.no_table_records = flag("OBNSL_NO_TABLE_RECORDS");
  x.runs = !flag("OBNSL_NO_RUNS");
  x.run_phases = {0.80, 0.12, 0.08};
  x.run_pieces = {1, 1, 1};
  x.runs_max_steps = (uint32_t)number("OBNSL_RUNS_MAX_STEPS", 2048);
  {
    const std::vector<double> ph = shares("OBNSL_RUN_PHASES");
    if (!ph.empty()) {
      x.run_phases = ph;
      x.run_pieces.assign(ph.size(), 1);
    }
    const std::vector<uint32_t> pc = counts_list("OBNSL_RUN_PIECES", x.run_phases.size());
    if (!pc.empty()) x.run_pieces = pc;
  }
  x.wide_loads = flag("OBNSL_WIDE_LOADS");
  x.together = !flag("OBNSL_NO_TOGETHER");
  x.together_phases = {0.72, 0.28};
  x.together_pieces = {1, 2};
  {
    const std::vector<double> ph = shares("OBNSL_TOGETHER_PHASES");
    if (!ph.empty()) {
      x.together_phases = ph;
      x.together_pieces.assign(ph.size(), 1);
      x.together_phases_set = true;
    }
    const std::vector<uint32_t> pc = counts_list("OBNSL_TOGETHER_PIECES", x.together_phases.size());
    if (!pc.empty()) x.together_pieces = pc;
  }
#ifdef OBNSL_PROBES
  x.probe = (int)number("OBNSL_PROBE", 0);
#endif
  return x;
}

}  // namespace

const Env& env() {
  static const Env e = read_env();
  return e;
}

}  // namespace obnsl_internal
