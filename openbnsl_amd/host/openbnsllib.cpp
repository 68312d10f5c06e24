// Python module `openbnsllib`: the reference's Python surface for the
// local-score path (reference: bindings/pybind_module.cpp:13-21 and
// bindings/submodules/{base,score,structure_learning}.cpp), backed by the HIP
// engine.  Names, argument names, defaults and exception types are the
// reference's; `hip_enabled` is this build's analogue of `cuda_enabled`.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <optional>

#include "openbnsl_host.h"

using namespace obnsl;

PYBIND11_MODULE(openbnsllib, m) {
  m.doc() = "OpenBNSL local-score path on AMD Instinct MI355X (drop-in for openbnsllib).";
  m.attr("cuda_enabled") = false;  // reference: pybind_module.cpp:15
  int ndev = 0;
  obnsl_device_count(&ndev);
  m.attr("hip_enabled") = ndev > 0;
  m.attr("hip_device_count") = ndev;

  auto base = m.def_submodule("base", "Base submodule");
  py::class_<DataframeWrapper>(base, "DataframeWrapper")
      .def(py::init<const py::object&>())
      // Not in the reference: data that is already coded, e.g. a generator's
      // uint8 matrix of 10^7 rows that could never pass through Python strings.
      // codes: uint8 [num_of_vars, num_of_datapoints]; values[v] = labels of
      // variable v in code order (default: zero-padded decimal codes, cardinality
      // = largest code + 1).
      .def_static("from_codes",
                  [](py::array_t<uint8_t, py::array::c_style | py::array::forcecast> a,
                     std::vector<std::string> columns,
                     std::optional<std::vector<std::vector<std::string>>> values) {
                    if (a.ndim() != 2) throw std::invalid_argument("Input must be a 2D array");
                    const size_t nv = static_cast<size_t>(a.shape(0)), rows = static_cast<size_t>(a.shape(1));
                    if (columns.size() != nv)
                      throw std::invalid_argument("coded data does not match its column/value lists");
                    std::vector<uint8_t> c(a.data(), a.data() + nv * rows);
                    std::vector<std::vector<std::string>> vals;
                    if (values) {
                      vals = std::move(*values);
                    } else {
                      vals.resize(nv);
                      for (size_t v = 0; v < nv; ++v) {
                        unsigned top = 0;
                        for (size_t i = 0; i < rows; ++i) top = std::max<unsigned>(top, c[v * rows + i]);
                        for (unsigned s = 0; s <= top; ++s) {
                          std::string t = std::to_string(s);
                          vals[v].push_back(std::string(3 - t.size(), '0') + t);
                        }
                      }
                    }
                    return DataframeWrapper(std::move(c), rows, std::move(columns), std::move(vals));
                  },
                  py::arg("codes"), py::arg("columns"), py::arg("values") = py::none())
      .def_readonly("num_of_vars", &DataframeWrapper::num_of_vars)
      .def_readonly("num_of_datapoints", &DataframeWrapper::num_of_datapoints)
      .def_readonly("col_idx2str", &DataframeWrapper::col_idx2str)
      .def_readonly("col_str2idx", &DataframeWrapper::col_str2idx)
      .def_readonly("val_idx2str", &DataframeWrapper::val_idx2str)
      .def_readonly("val_str2idx", &DataframeWrapper::val_str2idx)
      .def_readonly("num_of_values", &DataframeWrapper::num_of_values)
      .def_property_readonly("data_column_major", &DataframeWrapper::data_column_major)
      .def_property_readonly("data_row_major", &DataframeWrapper::data_row_major)
      // the coded matrix as one uint8[num_of_vars, num_of_datapoints] array
      .def_property_readonly("codes",
                             [](const DataframeWrapper& df) {
                               return py::array_t<uint8_t>(
                                   {df.num_of_vars, df.num_of_datapoints}, df.codes.data());
                             })
      .def("__repr__", [](const DataframeWrapper& df) {
        std::string r = "DataframeWrapper(\n  num_of_datapoints: " +
                        std::to_string(df.num_of_datapoints) + ",\n  num_of_vars: " +
                        std::to_string(df.num_of_vars) + ",\n  columns: [";
        for (size_t i = 0; i < df.col_idx2str.size(); ++i)
          r += (i ? ", " : "") + df.col_idx2str[i];
        return r + "]\n)";
      });

  py::class_<ContingencyTable>(base, "ContingencyTable")
      .def(py::init<const std::vector<size_t>&, const DataframeWrapper&>(), py::arg("var_ids"),
           py::arg("df"))
      .def_readonly("var_ids", &ContingencyTable::var_ids)
      .def_readonly("cardinalities", &ContingencyTable::cardinalities)
      .def_readonly("counts", &ContingencyTable::counts);

  py::class_<PDAG>(base, "PDAG")
      .def(py::init<size_t>())
      .def(py::init<const PDAG&>())
      .def_readonly("num_vars", &PDAG::num_vars)
      .def_readonly("parents", &PDAG::parents)
      .def("has_edge", &PDAG::has_edge, py::arg("from"), py::arg("to"))
      .def("add_edge", &PDAG::add_edge, py::arg("from"), py::arg("to"))
      .def("remove_edge", &PDAG::remove_edge, py::arg("from"), py::arg("to"));

  auto score = m.def_submodule("score", "Score submodule");
  py::class_<BDeu>(score, "BDeu")
      .def(py::init<double>(), py::arg("ess") = 1.0)
      .def_readonly("ess", &BDeu::ess)
      .def("__repr__", [](const BDeu& b) { return "BDeu(" + std::to_string(b.ess) + ")"; });
  py::class_<K2>(score, "K2").def(py::init<>()).def("__repr__", [](const K2&) { return "K2()"; });
  py::class_<BIC>(score, "BIC").def(py::init<>()).def("__repr__", [](const BIC&) { return "BIC()"; });
  py::class_<AIC>(score, "AIC").def(py::init<>()).def("__repr__", [](const AIC&) { return "AIC()"; });
  score.def("calculate_local_score", &calculate_local_score, py::arg("child_var"),
            py::arg("parent_set"), py::arg("ct"), py::arg("score_type"));

  auto sl = m.def_submodule("structure_learning", "Structure learning submodule");
  sl.def("exhaustive_search", &exhaustive_search, "Run exhaustive search algorithm",
         py::arg("df"), py::arg("score_type"), py::arg("max_parents"),
         py::arg("is_deterministic") = false, py::arg("devices") = std::vector<int>{});
  // reference: bindings/submodules/structure_learning.cpp:19-30 (same defaults)
  sl.def("simulated_annealing", &simulated_annealing, "Run simulated annealing algorithm",
         py::arg("df"), py::arg("score_type"), py::arg("max_parents"),
         py::arg("max_iters") = 10000, py::arg("init_temp") = 1.0,
         py::arg("cooling_rate") = 0.9995, py::arg("is_deterministic") = false,
         py::arg("seed") = 0, py::arg("num_chains") = 0);
  // Not in the reference: the score tables the search consumes, and the search
  // on given tables (host only; lets the dynamic program be tested without a GPU).
  sl.def("score_tables", &score_tables, py::arg("df"), py::arg("score_type"),
         py::arg("max_parents"));
  sl.def("save_score_tables", &save_score_tables, py::arg("path"), py::arg("df"),
         py::arg("score_type"), py::arg("max_parents"));
  py::class_<LoadedScores>(sl, "LoadedScores")
      .def_readonly("num_of_vars", &LoadedScores::num_of_vars)
      .def_readonly("num_of_datapoints", &LoadedScores::num_of_datapoints)
      .def_readonly("max_parents", &LoadedScores::max_parents)
      .def_readonly("num_of_values", &LoadedScores::num_of_values)
      .def_readonly("score_type", &LoadedScores::score_type)
      .def_readonly("tables", &LoadedScores::tables);
  sl.def("load_score_tables", &load_score_tables, py::arg("path"));
  sl.def("anneal_from_tables", &anneal_from_tables, py::arg("num_of_vars"), py::arg("max_parents"),
         py::arg("tables"), py::arg("max_iters") = 10000, py::arg("init_temp") = 1.0,
         py::arg("cooling_rate") = 0.9995, py::arg("is_deterministic") = false,
         py::arg("seed") = 0, py::arg("num_chains") = 0);
  sl.def("search_from_tables",
         [](size_t n, size_t max_parents, const std::vector<std::vector<double>>& tables) {
           if (n == 0 || n > 30 || max_parents > n - 1 || tables.size() != max_parents + 1)
             throw std::invalid_argument("max_parents must be in [0, num_of_vars-1]");
           double total = 0.0;
           PDAG g = search_from_tables(n, max_parents, tables, &total);
           return py::make_tuple(g, total);
         },
         py::arg("num_of_vars"), py::arg("max_parents"), py::arg("tables"));
}
