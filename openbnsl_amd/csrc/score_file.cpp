// score_file.cpp -- the score tables as a file.
//
// The reference keeps every local score inside one exhaustive_search call
// (src/structure_learning/exhaustive_search.cpp:61-73) and throws them away; a
// second search, an annealing run or a notebook pays for all the counting
// again.  The tables obnsl_score_all produces are small (8 bytes per score:
// 2.3 MB for 37 variables and at most 3 parents) and are what every consumer
// of the path reads, so they can be written once and reloaded.
//
// Layout, little-endian:
//   char     magic[8]   = "OBNSLSC1"
//   uint32   num_vars, max_parents, score_kind, reserved
//   uint64   num_rows
//   double   ess
//   uint32   cardinalities[num_vars]          (then padding to 8 bytes)
//   for k = 1 .. max_parents + 1:
//     uint64 num_scores = k * C(num_vars, k)
//     double scores[num_scores]              scores[rank * k + i] = score(v_i | S \ {v_i})
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "engine_internal.h"

using obnsl_internal::fail;

namespace {

constexpr char MAGIC[8] = {'O', 'B', 'N', 'S', 'L', 'S', 'C', '1'};

struct FileCloser {
  void operator()(std::FILE* f) const {
    if (f) std::fclose(f);
  }
};
using File = std::unique_ptr<std::FILE, FileCloser>;

struct Header {
  uint32_t num_vars = 0, max_parents = 0, kind = 0, reserved = 0;
  uint64_t num_rows = 0;
  double ess = 0.0;
};

obnsl_status read_header(std::FILE* f, const char* path, Header* h, std::vector<uint32_t>* card) {
  char magic[8];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, MAGIC, 8) != 0)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, std::string(path) + " is not a score-table file");
  if (std::fread(h, sizeof(Header), 1, f) != 1 || h->num_vars == 0 || h->num_vars > 65535 ||
      h->max_parents + 1 > OBNSL_MAX_SUBSET || h->max_parents >= h->num_vars)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, std::string(path) + ": damaged header");
  card->resize(h->num_vars + (h->num_vars & 1));
  if (std::fread(card->data(), sizeof(uint32_t), card->size(), f) != card->size())
    return fail(OBNSL_ERR_INVALID_ARGUMENT, std::string(path) + ": truncated header");
  card->resize(h->num_vars);
  return OBNSL_OK;
}

}  // namespace

extern "C" {

obnsl_status obnsl_scores_save(const char* path, uint32_t num_vars, uint64_t num_rows,
                               const uint32_t* cardinalities, obnsl_score_kind kind, double ess,
                               uint32_t max_parents, const double* const* tables) {
  if (!path || !cardinalities || !tables) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  if (num_vars == 0 || num_vars > 65535 || max_parents >= num_vars ||
      max_parents + 1 > OBNSL_MAX_SUBSET)
    return fail(OBNSL_ERR_INVALID_ARGUMENT, "max_parents must be in [0, num_of_vars-1]");
  File f(std::fopen(path, "wb"));
  if (!f) return fail(OBNSL_ERR_RUNTIME, std::string("cannot write ") + path);
  Header h;
  h.num_vars = num_vars;
  h.max_parents = max_parents;
  h.kind = static_cast<uint32_t>(kind);
  h.num_rows = num_rows;
  h.ess = ess;
  std::vector<uint32_t> card(cardinalities, cardinalities + num_vars);
  if (num_vars & 1) card.push_back(0);
  bool ok = std::fwrite(MAGIC, 1, 8, f.get()) == 8 && std::fwrite(&h, sizeof(h), 1, f.get()) == 1 &&
            std::fwrite(card.data(), sizeof(uint32_t), card.size(), f.get()) == card.size();
  for (uint32_t k = 1; ok && k <= max_parents + 1; ++k) {
    if (!tables[k - 1]) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null score table");
    const uint64_t count = obnsl_binomial(num_vars, k) * k;
    ok = std::fwrite(&count, sizeof(count), 1, f.get()) == 1 &&
         std::fwrite(tables[k - 1], sizeof(double), count, f.get()) == count;
  }
  if (!ok || std::fflush(f.get()) != 0) return fail(OBNSL_ERR_RUNTIME, std::string("write failed: ") + path);
  return OBNSL_OK;
}

obnsl_status obnsl_scores_info(const char* path, uint32_t* num_vars, uint64_t* num_rows,
                               uint32_t* cardinalities, uint32_t cardinalities_capacity,
                               obnsl_score_kind* kind, double* ess, uint32_t* max_parents) {
  if (!path) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  File f(std::fopen(path, "rb"));
  if (!f) return fail(OBNSL_ERR_RUNTIME, std::string("cannot read ") + path);
  Header h;
  std::vector<uint32_t> card;
  obnsl_status st = read_header(f.get(), path, &h, &card);
  if (st != OBNSL_OK) return st;
  if (num_vars) *num_vars = h.num_vars;
  if (num_rows) *num_rows = h.num_rows;
  if (kind) *kind = static_cast<obnsl_score_kind>(h.kind);
  if (ess) *ess = h.ess;
  if (max_parents) *max_parents = h.max_parents;
  if (cardinalities) {
    if (cardinalities_capacity < h.num_vars)
      return fail(OBNSL_ERR_INVALID_ARGUMENT, "cardinalities_capacity is smaller than num_vars");
    std::memcpy(cardinalities, card.data(), sizeof(uint32_t) * h.num_vars);
  }
  return OBNSL_OK;
}

obnsl_status obnsl_scores_load(const char* path, uint32_t k, double* scores_out, uint64_t capacity) {
  if (!path || !scores_out) return fail(OBNSL_ERR_INVALID_ARGUMENT, "null argument");
  File f(std::fopen(path, "rb"));
  if (!f) return fail(OBNSL_ERR_RUNTIME, std::string("cannot read ") + path);
  Header h;
  std::vector<uint32_t> card;
  obnsl_status st = read_header(f.get(), path, &h, &card);
  if (st != OBNSL_OK) return st;
  if (k == 0 || k > h.max_parents + 1)
    return fail(OBNSL_ERR_OUT_OF_RANGE, "the file holds no table of this subset size");
  for (uint32_t j = 1; j <= k; ++j) {
    uint64_t count = 0;
    if (std::fread(&count, sizeof(count), 1, f.get()) != 1 || count != obnsl_binomial(h.num_vars, j) * j)
      return fail(OBNSL_ERR_INVALID_ARGUMENT, std::string(path) + ": damaged table header");
    if (j < k) {
      if (std::fseek(f.get(), static_cast<long>(count * sizeof(double)), SEEK_CUR) != 0)
        return fail(OBNSL_ERR_RUNTIME, std::string("seek failed: ") + path);
      continue;
    }
    if (capacity < count) return fail(OBNSL_ERR_INVALID_ARGUMENT, "capacity is smaller than the table");
    if (std::fread(scores_out, sizeof(double), count, f.get()) != count)
      return fail(OBNSL_ERR_INVALID_ARGUMENT, std::string(path) + ": truncated table");
  }
  return OBNSL_OK;
}

}  // extern "C"
