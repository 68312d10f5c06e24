// DataframeWrapper ingest and the engine handle it owns.
//
// Behaviour kept from the reference (src/base/dataframe_wrapper.cpp:5-79):
// column order as given; per column the distinct values are turned into
// strings, sorted lexicographically, and a value's code is its rank (so "10"
// sorts before "2"); cardinality = number of OBSERVED values; the same error
// messages.  Differences, on purpose: the array is read through its strides, so
// C-ordered and Fortran-ordered frames both decode correctly (the reference
// assumes Fortran order, :42,67); integer and bool frames are accepted directly
// (the reference reinterprets any buffer as PyObject*, :30); a column with more
// than 256 distinct values is an error instead of a silent uint8 wrap.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

#include "openbnsl_host.h"

namespace obnsl {

void check(obnsl_status st) {
  if (st == OBNSL_OK) return;
  const std::string msg = obnsl_last_error();
  switch (st) {
    case OBNSL_ERR_INVALID_ARGUMENT:
    case OBNSL_ERR_UNSUPPORTED:
      throw std::invalid_argument(msg);
    case OBNSL_ERR_OUT_OF_RANGE:
      throw std::out_of_range(msg);
    default:
      throw std::runtime_error(msg);
  }
}

EngineData::EngineData(const uint8_t* codes, uint64_t rows, uint32_t vars, const uint32_t* card,
                       int device)
    : device_(device) {
  check(obnsl_data_create(codes, rows, rows, vars, card, device, 0, &h_));
}

EngineData::~EngineData() {
  if (h_) obnsl_data_destroy(h_);
}

namespace {

// Assigns codes by lexicographic rank of the distinct labels of one column.
// first_seen[i] = id of row i in order of first appearance; labels[id] = string.
void rank_codes(const std::vector<uint32_t>& first_seen, const std::vector<std::string>& labels,
                uint8_t* out, std::vector<std::string>& sorted_labels) {
  std::vector<uint32_t> order(labels.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(),
            [&](uint32_t a, uint32_t b) { return labels[a] < labels[b]; });
  std::vector<uint8_t> code_of(labels.size());
  sorted_labels.resize(labels.size());
  for (uint32_t r = 0; r < order.size(); ++r) {
    code_of[order[r]] = static_cast<uint8_t>(r);
    sorted_labels[r] = labels[order[r]];
  }
  for (size_t i = 0; i < first_seen.size(); ++i) out[i] = code_of[first_seen[i]];
}

int64_t read_int(const char* p, char kind, ssize_t size) {
  if (kind == 'b') return *reinterpret_cast<const uint8_t*>(p) != 0;
  if (kind == 'i') {
    switch (size) {
      case 1: return *reinterpret_cast<const int8_t*>(p);
      case 2: { int16_t v; std::memcpy(&v, p, 2); return v; }
      case 4: { int32_t v; std::memcpy(&v, p, 4); return v; }
      default: { int64_t v; std::memcpy(&v, p, 8); return v; }
    }
  }
  switch (size) {
    case 1: return *reinterpret_cast<const uint8_t*>(p);
    case 2: { uint16_t v; std::memcpy(&v, p, 2); return v; }
    case 4: { uint32_t v; std::memcpy(&v, p, 4); return v; }
    default: { uint64_t v; std::memcpy(&v, p, 8); return static_cast<int64_t>(v); }
  }
}

}  // namespace

DataframeWrapper::DataframeWrapper(const py::object& dataframe) {
  if (!py::hasattr(dataframe, "columns"))
    throw std::invalid_argument("Input must be a pandas dataframe");
  for (auto column : dataframe.attr("columns")) {
    if (!py::isinstance<py::str>(column))
      throw std::invalid_argument("Failed to cast column name to string");
    col_idx2str.push_back(column.cast<std::string>());
  }
  if (!py::hasattr(dataframe, "values"))
    throw std::invalid_argument("Input must be a pandas dataframe");
  py::array arr = py::array::ensure(dataframe.attr("values"));
  if (!arr || arr.ndim() != 2) throw std::invalid_argument("Input must be a 2D array");
  char kind = arr.dtype().kind();
  if (kind != 'O' && kind != 'i' && kind != 'u' && kind != 'b') {
    // fixed-width strings become str objects; anything else (floats ...) then
    // fails the string check below exactly as a non-str object does upstream
    arr = py::array::ensure(arr.attr("astype")(py::dtype("O")));
    kind = 'O';
  }
  num_of_datapoints = static_cast<size_t>(arr.shape(0));
  num_of_vars = static_cast<size_t>(arr.shape(1));
  if (num_of_vars != col_idx2str.size())
    throw std::invalid_argument("Input must be a pandas dataframe");
  const ssize_t s0 = arr.strides(0), s1 = arr.strides(1), isz = arr.itemsize();
  const char* base = static_cast<const char*>(arr.data());

  codes.assign(num_of_vars * num_of_datapoints, 0);
  val_idx2str.resize(num_of_vars);
  std::vector<uint32_t> ids(kind == 'O' ? num_of_datapoints : 0);
  for (size_t v = 0; v < num_of_vars && kind == 'O'; ++v) {
    std::vector<std::string> labels;
    const char* col = base + static_cast<ssize_t>(v) * s1;
    {
      std::unordered_map<std::string, uint32_t> seen;
      // the cells of a column usually share a few string objects: a small
      // direct-mapped cache by object address answers those without hashing the
      // text again (every object stays alive while the array holds it, so an
      // address names one string for the whole pass)
      struct Slot { PyObject* obj; uint32_t id; };
      Slot cache[256] = {};
      for (size_t i = 0; i < num_of_datapoints; ++i) {
        PyObject* o = *reinterpret_cast<PyObject* const*>(col + static_cast<ssize_t>(i) * s0);
        Slot& slot = cache[(reinterpret_cast<uintptr_t>(o) >> 4) & 255];
        if (o && slot.obj == o) {
          ids[i] = slot.id;
          continue;
        }
        Py_ssize_t len = 0;
        const char* s = (o && PyUnicode_Check(o)) ? PyUnicode_AsUTF8AndSize(o, &len) : nullptr;
        if (!s) {
          PyErr_Clear();
          throw std::invalid_argument("Failed to cast value to string");
        }
        auto it = seen.find(std::string(s, static_cast<size_t>(len)));
        if (it == seen.end()) {
          if (labels.size() == 256)
            throw std::invalid_argument("more than 256 distinct values in column " + col_idx2str[v]);
          it = seen.emplace(std::string(s, static_cast<size_t>(len)), labels.size()).first;
          labels.push_back(it->first);
        }
        ids[i] = it->second;
        slot = Slot{o, it->second};
      }
    }
    rank_codes(ids, labels, codes.data() + v * num_of_datapoints, val_idx2str[v]);
  }
  if (kind != 'O') ingest_integer_columns(base, s0, s1, isz, kind);
  finish_maps();
}

// Integer and bool frames touch no Python object, so the columns are decoded
// by a pool of threads with the GIL released.  Each value goes through a small
// direct-mapped cache in front of the (value -> first-seen id) map.
void DataframeWrapper::ingest_integer_columns(const char* base, ssize_t s0, ssize_t s1,
                                              ssize_t isz, char kind) {
  const size_t rows = num_of_datapoints;
  std::vector<std::string> errors(num_of_vars);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    std::vector<uint32_t> ids(rows);
    struct Slot { int64_t value; uint32_t id; bool used; };
    for (size_t v = next++; v < num_of_vars; v = next++) {
      std::vector<std::string> labels;
      std::unordered_map<int64_t, uint32_t> seen;
      std::vector<Slot> cache(1024, Slot{0, 0, false});
      const char* col = base + static_cast<ssize_t>(v) * s1;
      bool ok = true;
      for (size_t i = 0; i < rows && ok; ++i) {
        const int64_t x = read_int(col + static_cast<ssize_t>(i) * s0, kind, isz);
        Slot& slot = cache[static_cast<uint64_t>(x) & 1023];
        if (!slot.used || slot.value != x) {
          auto it = seen.find(x);
          if (it == seen.end()) {
            if (labels.size() == 256) {
              errors[v] = "more than 256 distinct values in column " + col_idx2str[v];
              ok = false;
              break;
            }
            it = seen.emplace(x, static_cast<uint32_t>(labels.size())).first;
            labels.push_back(kind == 'b' ? (x ? "True" : "False") : std::to_string(x));
          }
          slot = Slot{x, it->second, true};
        }
        ids[i] = slot.id;
      }
      if (ok) rank_codes(ids, labels, codes.data() + v * rows, val_idx2str[v]);
    }
  };
  {
    py::gil_scoped_release nogil;
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) {
      const long t = std::strtol(e, nullptr, 10);
      if (t > 0) nthreads = static_cast<unsigned>(t);
    }
    nthreads = std::max(1u, std::min<unsigned>(nthreads, static_cast<unsigned>(num_of_vars)));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  }
  for (const auto& e : errors)
    if (!e.empty()) throw std::invalid_argument(e);
}

// Data that is already coded: uint8 codes[num_of_vars][num_of_datapoints] with
// code = rank of the value's label, and the label lists per variable.
DataframeWrapper::DataframeWrapper(std::vector<uint8_t> c, size_t rows,
                                   std::vector<std::string> columns,
                                   std::vector<std::vector<std::string>> values)
    : num_of_vars(columns.size()),
      num_of_datapoints(rows),
      col_idx2str(std::move(columns)),
      val_idx2str(std::move(values)),
      codes(std::move(c)) {
  if (codes.size() != num_of_vars * num_of_datapoints || val_idx2str.size() != num_of_vars)
    throw std::invalid_argument("coded data does not match its column/value lists");
  for (size_t v = 0; v < num_of_vars; ++v) {
    const size_t r = val_idx2str[v].size();
    if (r == 0 || r > 256) throw std::invalid_argument("a variable needs 1..256 values");
    const uint8_t* col = codes.data() + v * num_of_datapoints;
    for (size_t i = 0; i < num_of_datapoints; ++i)
      if (col[i] >= r) throw std::invalid_argument("code out of range in column " + col_idx2str[v]);
  }
  finish_maps();
}

void DataframeWrapper::finish_maps() {
  col_str2idx.clear();
  for (size_t v = 0; v < num_of_vars; ++v) col_str2idx[col_idx2str[v]] = v;
  val_str2idx.assign(num_of_vars, {});
  num_of_values.assign(num_of_vars, 0);
  for (size_t v = 0; v < num_of_vars; ++v) {
    for (size_t s = 0; s < val_idx2str[v].size(); ++s)
      val_str2idx[v][val_idx2str[v][s]] = static_cast<uint8_t>(s);
    num_of_values[v] = val_idx2str[v].size();
  }
}

std::vector<std::vector<uint8_t>> DataframeWrapper::data_column_major() const {
  std::vector<std::vector<uint8_t>> out(num_of_vars);
  for (size_t v = 0; v < num_of_vars; ++v)
    out[v].assign(codes.begin() + v * num_of_datapoints, codes.begin() + (v + 1) * num_of_datapoints);
  return out;
}

std::vector<std::vector<uint8_t>> DataframeWrapper::data_row_major() const {
  std::vector<std::vector<uint8_t>> out(num_of_datapoints, std::vector<uint8_t>(num_of_vars));
  for (size_t v = 0; v < num_of_vars; ++v)
    for (size_t i = 0; i < num_of_datapoints; ++i) out[i][v] = codes[v * num_of_datapoints + i];
  return out;
}

// The device engine() uploads to (or has uploaded to): OPENBNSL_DEVICE, default 0.
int DataframeWrapper::engine_device() const {
  if (engine_) return engine_->device();
  int device = 0;
  if (const char* e = std::getenv("OPENBNSL_DEVICE")) {
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < 0 || v > 1023)
      throw std::invalid_argument("OPENBNSL_DEVICE must be a device index");
    device = static_cast<int>(v);
  }
  return device;
}

EngineData& DataframeWrapper::engine() const {
  if (!engine_) {
    if (num_of_vars == 0 || num_of_datapoints == 0)
      throw std::invalid_argument("empty data set");
    const int device = engine_device();
    std::vector<uint32_t> card(num_of_values.begin(), num_of_values.end());
    engine_ = std::make_shared<EngineData>(codes.data(), num_of_datapoints,
                                           static_cast<uint32_t>(num_of_vars), card.data(), device);
  }
  return *engine_;
}

}  // namespace obnsl
