// ref_bind.cpp -- driver that exposes the REFERENCE's own hot-path code to tests.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no algorithm: it includes the
// reference headers where they lie (-I/root/reference/include) and is linked
// with the reference's src/base/dataframe_wrapper.cpp by oracle/Makefile.  The
// result (oracle/_ref/openbnsl_ref*.so) is used to pin oracle.c and to
// generate tests/golden, and as the "reference" CPU baseline of bench.py.
// src/structure_learning/exhaustive_search.cpp is NOT part of this build: it
// includes <boost/multiprecision/cpp_int.hpp>, which this image lacks.
// src/structure_learning/simulated_annealing.cpp (a caller of the same table +
// score pair) needs nothing the image lacks and is linked in as it is.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <chrono>

#include "base/contingency_table.h"    // reference: include/base/contingency_table.h
#include "base/dataframe_wrapper.h"    // reference: include/base/dataframe_wrapper.h
#include "score/local_score.h"         // reference: include/score/local_score.h
#include "structure_learning/simulated_annealing.h"  // reference: include/structure_learning/simulated_annealing.h
#include "utils/combmask2vec.h"        // reference: include/utils/combmask2vec.h

namespace py = pybind11;

PYBIND11_MODULE(openbnsl_ref, m) {
  m.doc() = "OpenBNSL reference hot path, compiled from /root/reference sources";

  py::class_<DataframeWrapper>(m, "DataframeWrapper")
      .def(py::init<const py::object&>())
      .def_readonly("num_of_vars", &DataframeWrapper::num_of_vars)
      .def_readonly("num_of_datapoints", &DataframeWrapper::num_of_datapoints)
      .def_readonly("col_idx2str", &DataframeWrapper::col_idx2str)
      .def_readonly("col_str2idx", &DataframeWrapper::col_str2idx)
      .def_readonly("val_idx2str", &DataframeWrapper::val_idx2str)
      .def_readonly("val_str2idx", &DataframeWrapper::val_str2idx)
      .def_readonly("num_of_values", &DataframeWrapper::num_of_values)
      .def_readonly("data_column_major", &DataframeWrapper::data_column_major)
      .def_readonly("data_row_major", &DataframeWrapper::data_row_major);

  py::class_<ContingencyTable<true>>(m, "ContingencyTable")
      .def(py::init<const std::vector<size_t>&, const DataframeWrapper&>(),
           py::arg("var_ids"), py::arg("df"))
      .def_readonly("var_ids", &ContingencyTable<true>::var_ids)
      .def_readonly("cardinalities", &ContingencyTable<true>::cardinalities)
      .def_readonly("radix_weights", &ContingencyTable<true>::radix_weights)
      .def_readonly("counts", &ContingencyTable<true>::counts);

  py::class_<BDeu>(m, "BDeu").def(py::init<double>(), py::arg("ess") = 1.0);
  py::class_<K2>(m, "K2").def(py::init<>());
  py::class_<BIC>(m, "BIC").def(py::init<>());
  py::class_<AIC>(m, "AIC").def(py::init<>());

  m.def("calculate_local_score", &calculate_local_score<double, true>,
        py::arg("child_var"), py::arg("parent_set"), py::arg("ct"),
        py::arg("score_type"));

  m.def("combmask2vec", [](uint64_t mask) { return combmask2vec<uint64_t>(mask); });

  // One table plus one score per member, for each subset, exactly the calls of
  // exhaustive_search.cpp:55-73; timed inside C++ so that Python call overhead
  // is not part of the CPU baseline.  Returns (seconds, scores).
  m.def("score_subsets_timed",
        [](const DataframeWrapper& df, const std::vector<std::vector<size_t>>& subsets,
           double ess) {
          std::vector<std::vector<double>> scores(subsets.size());
          const auto t0 = std::chrono::steady_clock::now();
          for (size_t t = 0; t < subsets.size(); ++t) {
            ContingencyTable<true> ct(subsets[t], df);
            for (size_t child : subsets[t]) {
              std::vector<size_t> parent_set;
              for (size_t v : subsets[t])
                if (v != child) parent_set.push_back(v);
              scores[t].push_back(
                  calculate_local_score<double>(child, parent_set, ct, BDeu{ess}));
            }
          }
          const auto t1 = std::chrono::steady_clock::now();
          return py::make_tuple(std::chrono::duration<double>(t1 - t0).count(), scores);
        },
        py::arg("df"), py::arg("subsets"), py::arg("ess") = 1.0);

  // The reference's simulated_annealing (src/structure_learning/simulated_annealing.cpp:100-159)
  // with the defaults of its binding (bindings/submodules/structure_learning.cpp:19-30);
  // returns (seconds, parents per node) so that the same call serves as a CPU timing.
  m.def("simulated_annealing",
        [](const DataframeWrapper& df, double ess, size_t max_parents, size_t max_iters,
           double init_temp, double cooling_rate, bool is_deterministic, uint64_t seed,
           size_t num_chains) {
          const auto t0 = std::chrono::steady_clock::now();
          PDAG g = simulated_annealing(df, BDeu{ess}, max_parents, max_iters, init_temp,
                                       cooling_rate, is_deterministic, seed, num_chains);
          const auto t1 = std::chrono::steady_clock::now();
          std::vector<std::vector<size_t>> parents(g.num_vars);
          for (size_t v = 0; v < g.num_vars; ++v)
            parents[v].assign(g.parents[v].begin(), g.parents[v].end());
          return py::make_tuple(std::chrono::duration<double>(t1 - t0).count(), parents);
        },
        py::arg("df"), py::arg("ess"), py::arg("max_parents"), py::arg("max_iters") = 10000,
        py::arg("init_temp") = 1.0, py::arg("cooling_rate") = 0.9995,
        py::arg("is_deterministic") = false, py::arg("seed") = 0, py::arg("num_chains") = 0);
}
