// ContingencyTable / calculate_local_score / exhaustive_search of the drop-in
// module: argument checks and result types as in the reference, computation on
// the GPU through the C ABI, and the subset dynamic program on the host.
#include <cmath>
#include <limits>

#include "openbnsl_host.h"

#include <cstdlib>

namespace obnsl {

obnsl_score_kind score_kind(const ScoreType& s) {
  if (std::holds_alternative<BDeu>(s)) return OBNSL_SCORE_BDEU;
  if (std::holds_alternative<K2>(s)) return OBNSL_SCORE_K2;
  if (std::holds_alternative<BIC>(s)) return OBNSL_SCORE_BIC;
  return OBNSL_SCORE_AIC;
}

double score_ess(const ScoreType& s) {
  return std::holds_alternative<BDeu>(s) ? std::get<BDeu>(s).ess : 1.0;
}

// reference: include/base/contingency_table.h:34-56.  The N-row histogram runs
// on the GPU (obnsl_count); the sparse map the reference exposes is rebuilt
// from the dense result (non-zero cells, ascending key).
ContingencyTable::ContingencyTable(const std::vector<size_t>& ids, const DataframeWrapper& df)
    : var_ids(ids) {
  for (size_t i = 1; i < ids.size(); ++i)
    if (ids[i] < ids[i - 1]) throw std::invalid_argument("var_ids must be sorted");
  for (size_t v : ids)
    if (v >= df.num_of_vars) throw std::out_of_range("variable id out of range");
  cardinalities.resize(ids.size());
  radix_weights.resize(ids.size());
  size_t mult = 1;
  for (size_t i = ids.size(); i-- > 0;) {
    cardinalities[i] = df.num_of_values[ids[i]];
    radix_weights[i] = mult;
    mult *= cardinalities[i];
  }
  std::vector<uint32_t> v32(ids.begin(), ids.end());
  std::vector<uint64_t> dense(mult);
  EngineData& eng = df.engine();
  device = eng.device();
  {
    py::gil_scoped_release nogil;
    check(obnsl_count(eng.get(), v32.data(), static_cast<uint32_t>(v32.size()), dense.data(),
                      dense.size()));
  }
  for (size_t key = 0; key < mult; ++key)
    if (dense[key]) counts.emplace_hint(counts.end(), key, dense[key]);
}

// reference: include/score/local_score.h:22-62.  `parent_set` is ignored there
// too (:24): the parents are the table's other variables.
double calculate_local_score(size_t child_var, const std::vector<size_t>& /*parent_set*/,
                             const ContingencyTable& ct, const ScoreType& score_type) {
  size_t child_idx = ct.var_ids.size();
  for (size_t i = 0; i < ct.var_ids.size(); ++i)
    if (ct.var_ids[i] == child_var) child_idx = i;
  if (child_idx == ct.var_ids.size())
    throw std::invalid_argument("child_var not present in contingency table");
  size_t cells = 1;
  for (size_t c : ct.cardinalities) cells *= c;
  std::vector<uint64_t> dense(cells, 0);
  for (const auto& kv : ct.counts) dense[kv.first] = kv.second;
  std::vector<uint32_t> cards(ct.cardinalities.begin(), ct.cardinalities.end());
  double out = 0.0;
  py::gil_scoped_release nogil;
  check(obnsl_local_score(ct.device, dense.data(), cards.data(), static_cast<uint32_t>(cards.size()),
                          static_cast<uint32_t>(child_idx), score_kind(score_type),
                          score_ess(score_type), &out));
  return out;
}

std::vector<std::vector<double>> score_tables(const DataframeWrapper& df,
                                              const ScoreType& score_type, size_t max_parents) {
  EngineData& eng = df.engine();
  std::vector<std::vector<double>> tables(max_parents + 1);
  py::gil_scoped_release nogil;
  for (size_t k = 1; k <= max_parents + 1; ++k) {
    uint64_t nsub = 0;
    check(obnsl_num_subsets(eng.get(), static_cast<uint32_t>(k), &nsub));
    tables[k - 1].resize(nsub * k);
    check(obnsl_score_all(eng.get(), score_kind(score_type), score_ess(score_type),
                          static_cast<uint32_t>(k), 0, 1, tables[k - 1].data(), 0));
  }
  return tables;
}

// Subset dynamic program of the reference
// (src/structure_learning/exhaustive_search.cpp:47-148) on precomputed scores.
// Subsets are visited in increasing mask value instead of size-then-Gosper
// order: every table entry a subset reads belongs to a strict sub-mask, which is
// numerically smaller, so the values (and the tie-breaks, which only depend on
// the child / parent iteration order inside one subset) are the same.
//   best-parent table (:75-87): start from (score of the full candidate set, or
//   -inf beyond max_parents); a sub-table entry replaces it when >= (ties go to
//   the smaller set, later parents override earlier ones);
//   best sink (:89-94): strict >, first child wins ties, 0 = uninitialised.
// The best-parent table is stored per child over the 2^(n-1) masks that do not
// contain the child.
PDAG search_from_tables(size_t n, size_t max_parents,
                        const std::vector<std::vector<double>>& tables, double* total_score) {
  const uint64_t full = n >= 64 ? ~0ull : ((1ull << n) - 1);
  const uint64_t combs = 1ull << n;
  const uint64_t half = combs >> 1;
  std::vector<std::vector<uint64_t>> binom(n + 1, std::vector<uint64_t>(max_parents + 2, 0));
  for (size_t v = 0; v <= n; ++v) {
    binom[v][0] = 1;
    for (size_t j = 1; j <= max_parents + 1; ++j)
      binom[v][j] = v == 0 ? 0 : binom[v - 1][j - 1] + binom[v - 1][j];
  }
  std::vector<double> best_gs(combs, 0.0);
  std::vector<int8_t> best_ch(combs, -1);
  std::vector<double> bls(n * half);
  std::vector<uint32_t> bps(n * half);  // parent masks fit 32 bits: n <= 30 here
  auto squeeze = [](uint64_t mask, unsigned c) {  // drop bit position c
    return (mask & ((1ull << c) - 1)) | ((mask >> (c + 1)) << c);
  };
  const double ninf = -std::numeric_limits<double>::infinity();

  unsigned vars[64];
  for (uint64_t S = 1; S <= full; ++S) {
    unsigned k = 0;
    uint64_t rank = 0;
    for (uint64_t m = S; m; m &= m - 1) {
      vars[k] = static_cast<unsigned>(__builtin_ctzll(m));
      ++k;
      if (k <= max_parents + 1) rank += binom[vars[k - 1]][k];
    }
    const bool scored = k <= max_parents + 1;
    const double* sc = scored ? tables[k - 1].data() + rank * k : nullptr;
    double bgs = 0.0;
    int bch = -1;
    for (unsigned ci = 0; ci < k; ++ci) {
      const unsigned child = vars[ci];
      const uint64_t P = S & ~(1ull << child);
      double best = scored ? sc[ci] : ninf;
      uint64_t best_ps = P;
      for (unsigned pi = 0; pi < k; ++pi) {
        if (pi == ci) continue;
        const uint64_t sub = squeeze(P & ~(1ull << vars[pi]), child);
        const double l = bls[child * half + sub];
        if (l >= best) {
          best = l;
          best_ps = bps[child * half + sub];
        }
      }
      const uint64_t slot = child * half + squeeze(P, child);
      bls[slot] = best;
      bps[slot] = static_cast<uint32_t>(best_ps);
      const double gs = best_gs[P] + best;
      if (bgs >= 0 || gs > bgs) {
        bgs = gs;
        bch = static_cast<int>(child);
      }
    }
    best_gs[S] = bgs;
    best_ch[S] = static_cast<int8_t>(bch);
  }

  PDAG g(n);
  uint64_t S = full;
  if (total_score) *total_score = best_gs[full];
  for (size_t i = 0; i < n; ++i) {
    const unsigned ch = static_cast<unsigned>(best_ch[S]);
    const uint64_t P = S & ~(1ull << ch);
    const uint64_t ps = bps[ch * half + squeeze(P, ch)];
    for (size_t j = 0; j < n; ++j)
      if (ps >> j & 1) g.add_edge(j, ch);
    S = P;
  }
  return g;
}

// Devices named by OPENBNSL_DEVICES ("0,1,2,3"): the group a search is sharded over
// when the caller passes none.
static std::vector<int> devices_from_env() {
  std::vector<int> out;
  const char* e = std::getenv("OPENBNSL_DEVICES");
  if (!e) return out;
  const char* p = e;
  while (*p) {
    char* end = nullptr;
    const long v = std::strtol(p, &end, 10);
    while (end && *end == ' ') ++end;
    if (end == p || v < 0 || v > 1023 || (*end != ',' && *end != '\0'))
      throw std::invalid_argument("OPENBNSL_DEVICES must be a comma-separated list of device indices");
    out.push_back(static_cast<int>(v));
    p = *end == ',' ? end + 1 : end;
  }
  return out;
}

PDAG exhaustive_search(const DataframeWrapper& df, const ScoreType& score_type,
                       size_t max_parents, bool /*is_deterministic*/, std::vector<int> devices) {
  const size_t n = df.num_of_vars;
  // reference: exhaustive_search.cpp:18-23
  if (n == 0 || max_parents > n - 1)
    throw std::invalid_argument("max_parents must be in [0, num_of_vars-1]");
  if (n > 64) throw std::invalid_argument("num_of_vars must be in [0, 64]");
  const obnsl_score_kind kind = score_kind(score_type);
  if (kind != OBNSL_SCORE_BDEU && kind != OBNSL_SCORE_BIC)
    throw std::invalid_argument("Unsupported score type.");
  // Scores, subset dynamic program and back-track all run on the device
  // (obnsl_exhaustive_search).  The reference's n * 2^n tables (:30-40) stop
  // being allocatable long before 64 variables (and its `1 << n` overflows from
  // 31); the engine refuses what the device cannot hold instead of failing in the
  // allocator.  search_from_tables() below is the same program on the host.
  std::vector<uint64_t> masks(n, 0);
  if (devices.empty()) devices = devices_from_env();
  // a list of one device is honoured too: the wrapper's own engine when it lives on that
  // device, else a group of one rank there
  if (devices.size() == 1 && df.engine_device() == devices[0]) devices.clear();
  if (!devices.empty()) {
    // several GPUs: the scoring phase sharded over a group (RCCL exchanges inside
    // the engine), the dynamic program on the group's first device
    std::vector<uint32_t> card(df.num_of_values.begin(), df.num_of_values.end());
    obnsl_group* grp = nullptr;
    py::gil_scoped_release nogil;
    check(obnsl_group_create(df.codes.data(), df.num_of_datapoints, df.num_of_datapoints,
                             static_cast<uint32_t>(n), card.data(), devices.data(),
                             static_cast<uint32_t>(devices.size()), &grp));
    const obnsl_status st = obnsl_group_exhaustive_search(grp, kind, score_ess(score_type),
                                                          static_cast<uint32_t>(max_parents),
                                                          masks.data(), nullptr);
    obnsl_group_destroy(grp);
    check(st);
  } else {
    EngineData& eng = df.engine();
    py::gil_scoped_release nogil;
    check(obnsl_exhaustive_search(eng.get(), kind, score_ess(score_type),
                                  static_cast<uint32_t>(max_parents), masks.data(), nullptr));
  }
  PDAG g(n);
  for (size_t v = 0; v < n; ++v)
    for (size_t j = 0; j < n; ++j)
      if (masks[v] >> j & 1) g.add_edge(j, v);
  return g;
}

// reference: src/structure_learning/simulated_annealing.cpp:100-159.  The
// chains and their score lookups run inside the engine (obnsl_simulated_annealing).
PDAG simulated_annealing(const DataframeWrapper& df, const ScoreType& score_type,
                         size_t max_parents, size_t max_iters, double init_temp,
                         double cooling_rate, bool is_deterministic, uint64_t seed,
                         size_t num_chains) {
  const size_t n = df.num_of_vars;
  if (max_parents >= n) throw std::invalid_argument("max_parents out of range");  // :109-111
  const obnsl_score_kind kind = score_kind(score_type);
  if (kind != OBNSL_SCORE_BDEU && kind != OBNSL_SCORE_BIC)
    throw std::invalid_argument("Unsupported score type.");
  EngineData& eng = df.engine();
  const size_t width = max_parents ? max_parents : 1;
  std::vector<uint32_t> parents(n * width, 0), count(n, 0);
  {
    py::gil_scoped_release nogil;
    check(obnsl_simulated_annealing(eng.get(), kind, score_ess(score_type),
                                    static_cast<uint32_t>(max_parents), max_iters, init_temp,
                                    cooling_rate, is_deterministic ? 1 : 0, seed,
                                    static_cast<uint32_t>(num_chains), parents.data(),
                                    count.data(), nullptr));
  }
  PDAG g(n);
  for (size_t v = 0; v < n; ++v)
    for (size_t j = 0; j < count[v]; ++j) g.add_edge(parents[v * width + j], v);
  return g;
}

PDAG anneal_from_tables(size_t n, size_t max_parents, const std::vector<std::vector<double>>& tables,
                        size_t max_iters, double init_temp, double cooling_rate,
                        bool is_deterministic, uint64_t seed, size_t num_chains) {
  if (max_parents >= n) throw std::invalid_argument("max_parents out of range");
  if (tables.size() != max_parents + 1)
    throw std::invalid_argument("one score table per subset size 1..max_parents+1 is needed");
  std::vector<const double*> ptrs;
  for (size_t k = 1; k <= max_parents + 1; ++k) {
    if (tables[k - 1].size() != obnsl_binomial(static_cast<uint32_t>(n), static_cast<uint32_t>(k)) * k)
      throw std::invalid_argument("score table of the wrong size");
    ptrs.push_back(tables[k - 1].data());
  }
  const size_t width = max_parents ? max_parents : 1;
  std::vector<uint32_t> parents(n * width, 0), count(n, 0);
  {
    py::gil_scoped_release nogil;
    check(obnsl_anneal_tables(static_cast<uint32_t>(n), static_cast<uint32_t>(max_parents),
                              ptrs.data(), max_iters, init_temp, cooling_rate,
                              is_deterministic ? 1 : 0, seed, static_cast<uint32_t>(num_chains),
                              parents.data(), count.data(), nullptr));
  }
  PDAG g(n);
  for (size_t v = 0; v < n; ++v)
    for (size_t j = 0; j < count[v]; ++j) g.add_edge(parents[v * width + j], v);
  return g;
}

void save_score_tables(const std::string& path, const DataframeWrapper& df,
                       const ScoreType& score_type, size_t max_parents) {
  if (df.num_of_vars == 0 || max_parents > df.num_of_vars - 1)
    throw std::invalid_argument("max_parents must be in [0, num_of_vars-1]");
  const auto tables = score_tables(df, score_type, max_parents);
  std::vector<const double*> ptrs;
  for (const auto& t : tables) ptrs.push_back(t.data());
  std::vector<uint32_t> card(df.num_of_values.begin(), df.num_of_values.end());
  check(obnsl_scores_save(path.c_str(), static_cast<uint32_t>(df.num_of_vars), df.num_of_datapoints,
                          card.data(), score_kind(score_type), score_ess(score_type),
                          static_cast<uint32_t>(max_parents), ptrs.data()));
}

LoadedScores load_score_tables(const std::string& path) {
  LoadedScores out;
  uint32_t n = 0, mp = 0;
  uint64_t rows = 0;
  obnsl_score_kind kind = OBNSL_SCORE_BDEU;
  double ess = 1.0;
  check(obnsl_scores_info(path.c_str(), &n, &rows, nullptr, 0, &kind, &ess, &mp));
  std::vector<uint32_t> card(n);
  check(obnsl_scores_info(path.c_str(), nullptr, nullptr, card.data(), n, nullptr, nullptr, nullptr));
  out.num_of_vars = n;
  out.num_of_datapoints = rows;
  out.max_parents = mp;
  out.num_of_values.assign(card.begin(), card.end());
  switch (kind) {
    case OBNSL_SCORE_BDEU: out.score_type = BDeu(ess); break;
    case OBNSL_SCORE_K2: out.score_type = K2{}; break;
    case OBNSL_SCORE_BIC: out.score_type = BIC{}; break;
    default: out.score_type = AIC{}; break;
  }
  out.tables.resize(mp + 1);
  for (uint32_t k = 1; k <= mp + 1; ++k) {
    out.tables[k - 1].resize(obnsl_binomial(n, k) * k);
    check(obnsl_scores_load(path.c_str(), k, out.tables[k - 1].data(), out.tables[k - 1].size()));
  }
  return out;
}

}  // namespace obnsl
