// Host-side C++ types of the drop-in `openbnsllib` module.  They keep the
// reference's names, fields and error behaviour for the local-score path, and
// route every computation to the HIP engine through include/obnsl_engine.h.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "obnsl_engine.h"

namespace py = pybind11;

namespace obnsl {

// Turns an engine status into the exception type the reference would raise
// (pybind11 maps invalid_argument -> ValueError, out_of_range -> IndexError,
// runtime_error -> RuntimeError).
void check(obnsl_status st);

// Owns one obnsl_data handle.
class EngineData {
 public:
  EngineData(const uint8_t* codes, uint64_t rows, uint32_t vars, const uint32_t* card, int device);
  ~EngineData();
  EngineData(const EngineData&) = delete;
  EngineData& operator=(const EngineData&) = delete;
  obnsl_data* get() const { return h_; }
  int device() const { return device_; }

 private:
  obnsl_data* h_ = nullptr;
  int device_ = 0;
};

// reference: include/base/dataframe_wrapper.h:24-46.  Same public fields; the
// two data matrices are kept as ONE contiguous column-major uint8 buffer
// (`codes`) and materialised as nested lists only when Python asks for them.
struct DataframeWrapper {
  explicit DataframeWrapper(const py::object& dataframe);
  // engine-side ingest for data that is already coded (uint8, column-major)
  DataframeWrapper(std::vector<uint8_t> codes, size_t rows, std::vector<std::string> columns,
                   std::vector<std::vector<std::string>> values);

  size_t num_of_vars = 0;
  size_t num_of_datapoints = 0;
  std::vector<std::string> col_idx2str;
  std::map<std::string, size_t> col_str2idx;
  std::vector<std::vector<std::string>> val_idx2str;
  std::vector<std::map<std::string, uint8_t>> val_str2idx;
  std::vector<size_t> num_of_values;
  std::vector<uint8_t> codes;  // [num_of_vars][num_of_datapoints]

  std::vector<std::vector<uint8_t>> data_column_major() const;
  std::vector<std::vector<uint8_t>> data_row_major() const;

  // device copy, created on first use (throws std::runtime_error without a GPU)
  EngineData& engine() const;
  int engine_device() const;

 private:
  void finish_maps();
  void ingest_integer_columns(const char* base, ssize_t s0, ssize_t s1, ssize_t isz, char kind);
  mutable std::shared_ptr<EngineData> engine_;
};

// reference: include/base/contingency_table.h:22-86 (the Deterministic = true
// instantiation, the only one bound or used by exhaustive_search).
struct ContingencyTable {
  std::vector<size_t> var_ids;
  std::vector<size_t> cardinalities;
  std::vector<size_t> radix_weights;
  std::map<size_t, size_t> counts;  // non-zero cells only, ascending key

  ContingencyTable(const std::vector<size_t>& var_ids, const DataframeWrapper& df);

  size_t radix_weight(size_t idx) const noexcept { return radix_weights[idx]; }
  size_t state_of(size_t key, size_t idx) const noexcept {
    return (key / radix_weights[idx]) % cardinalities[idx];
  }
  size_t strip(size_t key, size_t idx) const noexcept {
    return key - state_of(key, idx) * radix_weights[idx];
  }
  bool contains(size_t key) const noexcept { return counts.count(key) != 0; }
  int device = 0;
};

// reference: include/score/score_type.h:9-63
struct BDeu {
  double ess;
  explicit BDeu(double e = 1.0) : ess(e) {}
};
struct K2 {};
struct BIC {};
struct AIC {};
using ScoreType = std::variant<BDeu, K2, BIC, AIC>;
obnsl_score_kind score_kind(const ScoreType& s);
double score_ess(const ScoreType& s);

// reference: include/score/local_score.h:22-26
double calculate_local_score(size_t child_var, const std::vector<size_t>& parent_set,
                             const ContingencyTable& ct, const ScoreType& score_type);

// reference: include/graph/pdag.h:16-37
struct PDAG {
  size_t num_vars;
  std::vector<std::set<size_t>> parents;
  explicit PDAG(size_t n) : num_vars(n), parents(n) {}
  bool has_edge(size_t from, size_t to) const { return parents.at(to).count(from) != 0; }
  void add_edge(size_t from, size_t to) { parents.at(to).insert(from); }
  void remove_edge(size_t from, size_t to) { parents.at(to).erase(from); }
};

// reference: include/structure_learning/exhaustive_search.h:17-20
// `devices`: HIP devices to shard the scoring phase over (not in the reference,
// which is single-device); empty = OPENBNSL_DEVICES if set, else the device the
// DataframeWrapper lives on.
PDAG exhaustive_search(const DataframeWrapper& df, const ScoreType& score_type,
                       size_t max_parents, bool is_deterministic, std::vector<int> devices = {});

// reference: include/structure_learning/simulated_annealing.h:20-28
PDAG simulated_annealing(const DataframeWrapper& df, const ScoreType& score_type,
                         size_t max_parents, size_t max_iters, double init_temp,
                         double cooling_rate, bool is_deterministic, uint64_t seed,
                         size_t num_chains);

// The score tables the search consumes: tables[k-1][rank*k + i], k = 1..max_parents+1.
std::vector<std::vector<double>> score_tables(const DataframeWrapper& df,
                                              const ScoreType& score_type, size_t max_parents);

// Score tables on disk (obnsl_scores_save / obnsl_scores_load).
void save_score_tables(const std::string& path, const DataframeWrapper& df,
                       const ScoreType& score_type, size_t max_parents);
struct LoadedScores {
  size_t num_of_vars = 0, num_of_datapoints = 0, max_parents = 0;
  std::vector<size_t> num_of_values;
  ScoreType score_type;
  std::vector<std::vector<double>> tables;
};
LoadedScores load_score_tables(const std::string& path);

// The annealing chains on given score tables (host only).
PDAG anneal_from_tables(size_t n, size_t max_parents, const std::vector<std::vector<double>>& tables,
                        size_t max_iters, double init_temp, double cooling_rate,
                        bool is_deterministic, uint64_t seed, size_t num_chains);

// The dynamic program of the search on given score tables (host only).
PDAG search_from_tables(size_t n, size_t max_parents,
                        const std::vector<std::vector<double>>& tables, double* total_score);

}  // namespace obnsl
