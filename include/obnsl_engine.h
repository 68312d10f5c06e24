/*
 * obnsl_engine.h -- C ABI of the MI355X local-score engine (libobnsl_engine.so).
 *
 * The reference has no C ABI: its boundary is a pybind11 module over C++ free
 * functions and structs.  Each entry point below names the reference interface
 * it replaces (file:line relative to the reference checkout).  The pybind11
 * module openbnsllib (openbnsl_amd/host/) is a thin shim over these calls and
 * keeps the reference's Python names, argument meaning and exception types.
 *
 * Conventions
 *  - every function returns an obnsl_status; nothing throws across the ABI;
 *    obnsl_last_error() gives the message of the calling thread's last failure;
 *  - the data matrix is uint8 state codes, column-major: code of variable v in
 *    row i is codes[v * col_stride + i]  (reference: data_column_major,
 *    include/base/dataframe_wrapper.h:42-43);
 *  - a variable subset is an ascending list of variable ids; its joint table is
 *    dense, keyed as the reference keys it: the LAST variable has radix weight 1
 *    (reference: include/base/contingency_table.h:43-49,77-85);
 *  - size-k subsets are numbered by colex rank, rank(S) = sum_j C(v_j, j) for
 *    v_1 < ... < v_k, which is the order in which the reference visits them
 *    (Gosper order, src/structure_learning/exhaustive_search.cpp:51-52,129);
 *  - an obnsl_data handle serves one call at a time (the caller serialises calls
 *    on one handle; different handles may be used from different threads).  The
 *    reference holds the GIL for the whole of every call
 *    (bindings/pybind_module.cpp: no gil_scoped_release), so it makes no promise
 *    beyond that either;
 *  - all compute runs on the GPU; there is no CPU fallback.  Without a usable
 *    HIP device every compute call fails with OBNSL_ERR_NO_DEVICE.
 */
#ifndef OBNSL_ENGINE_H
#define OBNSL_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  OBNSL_OK = 0,
  OBNSL_ERR_INVALID_ARGUMENT = 1, /* reference: std::invalid_argument -> ValueError */
  OBNSL_ERR_OUT_OF_RANGE = 2,     /* reference: std::out_of_range    -> IndexError */
  OBNSL_ERR_RUNTIME = 3,          /* reference: std::runtime_error   -> RuntimeError */
  OBNSL_ERR_NO_DEVICE = 4,        /* no HIP device / HIP runtime failure at init */
  OBNSL_ERR_UNSUPPORTED = 5       /* "Unsupported score type." (local_score.h:42-44) */
} obnsl_status;

/* reference: std::variant<BDeu,K2,BIC,AIC>, include/score/score_type.h:39.
 * BDeu is what the reference implements; BIC is this engine's addition (the
 * reference throws for it); K2 and AIC are rejected as the reference rejects
 * them. */
typedef enum {
  OBNSL_SCORE_BDEU = 0,
  OBNSL_SCORE_K2 = 1,
  OBNSL_SCORE_BIC = 2,
  OBNSL_SCORE_AIC = 3
} obnsl_score_kind;

/* Largest subset size (max_parents + 1) the engine counts and scores.  The
 * reference has no such bound (contingency_table.h:34-56 takes any var_ids); a
 * table here is dense, so it also has to stay within 2^26 cells.  Sizes up to 8
 * go through the bit-plane contraction, larger ones through the per-row
 * histogram kernel. */
#define OBNSL_MAX_SUBSET 32
#define OBNSL_NOALLOC_MAX_VARS 4096 /* variables obnsl_best_parents_device_noalloc serves */

typedef struct obnsl_data obnsl_data; /* device-resident data set (opaque) */

const char* obnsl_last_error(void);
int obnsl_abi_version(void);

/* Number of visible HIP devices (0 when there is none; never fails for that). */
obnsl_status obnsl_device_count(int* count);

/* Replaces the storage half of DataframeWrapper
 * (include/base/dataframe_wrapper.h:33-45): uploads the coded data matrix and
 * the observed cardinalities (num_of_values) to `device` and builds the
 * per-(variable,state) row bitmaps the counting kernels read.
 * codes_on_device != 0: `codes` is a device pointer on `device` (no host copy). */
obnsl_status obnsl_data_create(const uint8_t* codes, uint64_t num_rows, uint64_t col_stride,
                               uint32_t num_vars, const uint32_t* cardinalities, int device,
                               int codes_on_device, obnsl_data** out);
obnsl_status obnsl_data_destroy(obnsl_data* d);

/* Replaces the fields num_of_datapoints / num_of_vars of DataframeWrapper
 * (include/base/dataframe_wrapper.h:35-36).  Either output may be null. */
obnsl_status obnsl_data_shape(const obnsl_data* d, uint64_t* num_rows, uint32_t* num_vars);

/* Work is queued on this HIP stream (a hipStream_t; NULL = the null stream).
 * Calls that return results to host memory synchronise it before returning. */
obnsl_status obnsl_data_set_stream(obnsl_data* d, void* hip_stream);

/* Rebuilds the row bitmaps from the resident uint8 matrix (one pass over the
 * data; obnsl_data_create already did it once).  Asynchronous on the stream. */
obnsl_status obnsl_data_pack(obnsl_data* d);

/* Replaces ContingencyTable<true>::ContingencyTable(var_ids, df)
 * (include/base/contingency_table.h:34-56): dense joint counts of the subset,
 * prod(card) cells written to counts_out (host).  Unsorted var_ids ->
 * OBNSL_ERR_INVALID_ARGUMENT "var_ids must be sorted" (:37-38). */
obnsl_status obnsl_count(obnsl_data* d, const uint32_t* var_ids, uint32_t k,
                         uint64_t* counts_out, uint64_t counts_capacity);

/* Replaces calculate_local_score<double,true>(child, parents, ct, score)
 * (include/score/local_score.h:22-62) for a table already counted: `counts`
 * (host, dense, keyed as above) over variables with cardinalities cards[0..k),
 * child = position child_idx in that list.  The fp64 evaluation runs on
 * `device`.  K2/AIC -> OBNSL_ERR_UNSUPPORTED. */
obnsl_status obnsl_local_score(int device, const uint64_t* counts, const uint32_t* cards,
                               uint32_t k, uint32_t child_idx, obnsl_score_kind kind,
                               double ess, double* score_out);

/* Replaces the scoring phase of exhaustive_search
 * (src/structure_learning/exhaustive_search.cpp:47-73) for one subset size:
 * for every size-k subset in colex-rank order, one joint table and one score
 * per member taken as the child (members ascending):
 *   scores_out[rank * k + i] = score(v_i | S \ {v_i}).
 * shard/num_shards split the subsets over GPUs: this call fills only the
 * entries of its shard (the partition is by cost, deterministic in
 * (k, num_shards)) and leaves the others untouched; (0, 1) = everything.
 * out_on_device != 0: scores_out is a device pointer and the call is
 * asynchronous on the data set's stream. */
obnsl_status obnsl_score_all(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t k,
                             uint32_t shard, uint32_t num_shards, double* scores_out,
                             int out_on_device);

/* Scheduling hint with no counterpart in the reference: queue the counting of
 * every subset size 1..kmax at once (sizes below kmax whole, size kmax for this
 * shard), the small launches on an internal second stream so that they overlap
 * the large one.  obnsl_score_all calls that follow, on the same row bitmaps,
 * reuse these counts.  Calling it is optional; results are the same without. */
obnsl_status obnsl_precount(obnsl_data* d, uint32_t kmax, uint32_t shard, uint32_t num_shards);

/* Multi-GPU form of obnsl_precount.  The scorer of a size-k table reads the
 * count tables of all smaller sizes, so obnsl_precount counts those whole on
 * every rank.  This call shards the second-largest size as well: the rank
 * counts only its slice of that table (the rest is zero) and reports
 * *sum_size_out = that subset size and *sum_cells_out = the table's number of
 * uint32 counters; the caller then sums the table over the ranks --
 * obnsl_counts_export into a device buffer, all-reduce(sum) over the ranks
 * (RCCL), obnsl_counts_import -- before the obnsl_score_all calls, which refuse
 * to run on an unsummed table.  *sum_size_out = 0: nothing to sum (one shard,
 * a small problem, or a size served by the per-row histogram path).
 * No counterpart in the reference (single process). */
obnsl_status obnsl_precount_sharded(obnsl_data* d, uint32_t kmax, uint32_t shard,
                                    uint32_t num_shards, uint32_t* sum_size_out,
                                    uint64_t* sum_cells_out);
/* The whole count table of subset size k, as counted for the current row
 * bitmaps: its number of uint32 counters, and copies to / from a device buffer
 * of that size, asynchronous on the data set's stream.  Import marks the table
 * as summed. */
obnsl_status obnsl_counts_cells(obnsl_data* d, uint32_t k, uint64_t* cells_out);
/* For callers that can run the all-reduce on the engine's own memory: the device
 * address of that table (stable while the data set lives; work queued on the
 * data set's stream so far is ordered before anything the caller queues there),
 * and the acknowledgement that it now holds the sum over the ranks. */
obnsl_status obnsl_counts_pointer(obnsl_data* d, uint32_t k, uint32_t** d_table_out,
                                  uint64_t* cells_out);
obnsl_status obnsl_counts_mark_summed(obnsl_data* d, uint32_t k); /* k = 0: every size */
obnsl_status obnsl_counts_export(obnsl_data* d, uint32_t k, uint32_t* d_dst);
obnsl_status obnsl_counts_import(obnsl_data* d, uint32_t k, const uint32_t* d_src);

/* Ranks that split the ROWS instead of the subsets (no counterpart in the reference,
 * whose one loop runs over all rows, contingency_table.h:52-55).  For a short pass
 * -- few tiles per subset size, where a split of the subsets leaves ranks without
 * work and every rank with the same fixed part -- each rank creates its data set
 * over a slice of the rows (obnsl_data_create on codes + first_row, the same
 * col_stride and cardinalities on every rank) and names the rows of the whole
 * data set once with obnsl_data_set_total_rows.  Per pass: obnsl_data_pack (the
 * slice only), obnsl_precount_rows -- every subset size 1..kmax counted over the
 * slice in one launch, the count tables in ONE contiguous range of device memory
 * (*d_counts_out, *words_out uint32 words, the same layout on every rank) -- one
 * all-reduce(sum) of that range over the ranks on the data set's stream,
 * obnsl_counts_mark_summed(d, 0) (0 = every size), then obnsl_score_pass with
 * (shard, num_shards) = (rank, world): every rank now holds every count and scores
 * its share of the tables with the total number of rows.  Scoring calls refuse to
 * run before the sum is acknowledged.  Every size up to kmax must be served by the
 * bit-plane contraction (else OBNSL_ERR_UNSUPPORTED: counts of the per-row path are
 * scored as they are produced and never leave the call); the choice is made from
 * total_rows, so it is the same on every rank.  *words_out = 0: the tables of the
 * current row bitmaps were counted and summed by an earlier call (nothing to add). */
obnsl_status obnsl_data_set_total_rows(obnsl_data* d, uint64_t total_rows);
obnsl_status obnsl_precount_rows(obnsl_data* d, uint32_t kmax, uint32_t** d_counts_out,
                                 uint64_t* words_out);

/* Inspection of what the batch counting kernel produced (no counterpart in the
 * reference, which keeps one table at a time, contingency_table.h:24-27): the
 * slice of the resident size-k count table that belongs to the subset var_ids
 * (ascending) -- the counts of the joint states in which EVERY variable is in
 * one of its states 1..r-1, prod(r_v - 1) words, first variable most
 * significant (state s of a variable is digit s - 1).  The full table follows
 * from these slices of all sub-subsets by inclusion-exclusion (what the scorer
 * does).  Counts the size first if the current row bitmaps have not been
 * counted yet.  OBNSL_ERR_UNSUPPORTED for a size served by the per-row path. */
obnsl_status obnsl_reduced_table(obnsl_data* d, const uint32_t* var_ids, uint32_t k,
                                 uint32_t* table_out, uint64_t capacity);

/* Number of size-k subsets, C(num_vars, k), and of scores, k * C(num_vars, k). */
obnsl_status obnsl_num_subsets(const obnsl_data* d, uint32_t k, uint64_t* num_subsets);

/* Explicit-list form of the same pair of calls (what a caller such as
 * simulated_annealing.cpp:77-81 issues one at a time): subsets[m][k] ascending
 * ids; scores_out[m][k] on the host. */
obnsl_status obnsl_score_subsets(obnsl_data* d, obnsl_score_kind kind, double ess,
                                 const uint32_t* subsets, uint32_t k, uint64_t m,
                                 double* scores_out);

/* Per-node best parent set over all parent sets of size <= max_parents, from
 * device score tables produced by obnsl_score_all (d_scores[k-1] = table of
 * subset size k, k = 1..max_parents+1, device pointers).  Entries still equal to
 * the fill value -inf (other shards) are ignored.  Outputs (host):
 *   best_score[v]; best_size[v] = number of parents; best_rank[v] = colex rank
 *   of the parent set among subsets of its size.  Ties: fewer parents, then
 *   lower rank.  This is the per-GPU partial result that ranks all-gather. */
obnsl_status obnsl_best_parents(obnsl_data* d, const double* const* d_scores,
                                uint32_t max_parents, double* best_score,
                                uint32_t* best_size, uint64_t* best_rank);

/* The same reduction left on the device and asynchronous on the data set's
 * stream: d_out (device, 3 * num_vars doubles) receives, per node, the best
 * score, the number of parents and the colex rank of the parent set (integers
 * stored exactly as doubles; -inf, 0, 0 for a node this shard has no score for).
 * This is the record that ranks all-gather. */
obnsl_status obnsl_best_parents_device(obnsl_data* d, const double* const* d_scores,
                                       uint32_t max_parents, double* d_out);

/* obnsl_best_parents_device for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Same inputs, same d_out, same
 * kernels; the per-node arg-max words that obnsl_best_parents_device keeps in a
 * buffer it allocates live here in a fixed array of the library's code object
 * (2 * OBNSL_NOALLOC_MAX_VARS 8-byte words = 64 KiB per GPU, held until the
 * process ends).  Calls hipMalloc / hipFree never; the kernels clear the words
 * they use, with a fill kernel (no memset node).  A data set with more than
 * OBNSL_NOALLOC_MAX_VARS variables -> OBNSL_ERR_UNSUPPORTED.  Every call on one
 * GPU uses the same array; the library orders them on the device (each call's
 * kernels wait for the previous call's through an event that obnsl_data_create
 * set up, whatever their streams and data sets), so calls from two streams or
 * two handles are safe but do not overlap. */
obnsl_status obnsl_best_parents_device_noalloc(obnsl_data* d, const double* const* d_scores,
                                               uint32_t max_parents, double* d_out);

/* The scoring phase of exhaustive_search
 * (src/structure_learning/exhaustive_search.cpp:47-73) for EVERY subset size
 * 1..max_parents+1 in one call, followed by the per-node reduction of
 * obnsl_best_parents_device: the same outputs as one obnsl_score_all per size
 * (d_tables[k-1] = device table of subset size k; shard / num_shards as there)
 * and, when d_best_out is not null, one obnsl_best_parents_device (3 * num_vars
 * doubles, device).  Where every size goes through the bit-plane contraction and
 * max_parents + 1 <= 5, the sizes share their launches (one evaluates every
 * table's functional, one turns them into scores and finds each node's best);
 * otherwise the call is the per-size calls in a row.  The counts come from the
 * current row bitmaps (obnsl_precount / obnsl_precount_sharded first when the
 * counting should overlap or is sharded).  Asynchronous on the data set's stream. */
obnsl_status obnsl_score_pass(obnsl_data* d, obnsl_score_kind kind, double ess, uint32_t max_parents,
                              uint32_t shard, uint32_t num_shards, double* const* d_tables,
                              double* d_best_out);

/* The combine step after ranks have all-gathered their obnsl_best_parents_device
 * records: d_parts (device) = `world` records of 3 * num_vars doubles, rank
 * after rank; d_out (device, 3 * num_vars doubles) = per node the best of them
 * by the same rule (score, then fewer parents, then lower rank).  Asynchronous
 * on the data set's stream.  No counterpart in the reference (single process). */
obnsl_status obnsl_combine_best_device(obnsl_data* d, const double* d_parts, uint32_t world,
                                       double* d_out);

/* Replaces exhaustive_search(df, score_type, max_parents, is_deterministic)
 * (src/structure_learning/exhaustive_search.cpp:14-165) end to end on the
 * device: the score tables of sizes 1..max_parents+1, then the subset dynamic
 * program (one launch per subset size, the reference's tie rules) and the
 * back-track.  parents_out[v] (host, num_vars entries) = bitmask of the parents
 * of v in the optimal DAG; total_score_out (may be null) = its score.
 * Errors as the reference: max_parents > num_vars-1 and num_vars > 64 are
 * OBNSL_ERR_INVALID_ARGUMENT with its messages; so is a problem whose n * 2^n
 * tables do not fit the device (the reference cannot run those either). */
obnsl_status obnsl_exhaustive_search(obnsl_data* d, obnsl_score_kind kind, double ess,
                                     uint32_t max_parents, uint64_t* parents_out,
                                     double* total_score_out);

/* The second half of obnsl_exhaustive_search alone: the subset dynamic program
 * and the back-track (exhaustive_search.cpp:75-148) on score tables that are
 * already on the data set's device -- d_tables[k-1] = the table of subset size
 * k as obnsl_score_all writes it, k = 1..max_parents+1.  This is what runs on
 * one device after several have scored their shards. */
obnsl_status obnsl_exhaustive_search_tables(obnsl_data* d, const double* const* d_tables,
                                            uint32_t max_parents, uint64_t* parents_out,
                                            double* total_score_out);

/* ---- several GPUs behind one call --------------------------------------
 * The reference's entry point (bindings/submodules/structure_learning.cpp:12-18
 * -> exhaustive_search.cpp:14) is one thread on one device.  A group is one
 * host process driving `num_devices` GPUs: rank r = devices[r] holds a replica
 * of the data matrix (obnsl_data_create's arguments, host memory) and scores
 * shard r of every subset size; the ranks exchange over RCCL (xGMI): an
 * all-reduce of one count table before scoring, an all-gather of the per-node
 * records (or an all-reduce of the score tables) after it.  A group in which a
 * device appears twice (ranks sharing a GPU: how a one-GPU box rehearses the
 * sharded pass) or OBNSL_GROUP_EXCHANGE=copy exchanges by device copies
 * instead.  Calls on a group are serialised by the caller, like calls on a
 * data set. */
typedef struct obnsl_group obnsl_group;
obnsl_status obnsl_group_create(const uint8_t* codes, uint64_t num_rows, uint64_t col_stride,
                                uint32_t num_vars, const uint32_t* cardinalities,
                                const int* devices, uint32_t num_devices, obnsl_group** out);
obnsl_status obnsl_group_destroy(obnsl_group* g);
/* Either output may be null.  *uses_rccl: 1 = RCCL collectives, 0 = device copies. */
obnsl_status obnsl_group_info(const obnsl_group* g, uint32_t* num_devices, int* uses_rccl);

/* Host time (milliseconds) the last obnsl_group_* call spent queueing the ranks'
 * launches -- every rank on a host thread of its own, all at once -- as opposed to
 * waiting for the devices.  No counterpart in the reference. */
obnsl_status obnsl_group_host_time(const obnsl_group* g, double* enqueue_ms);

/* The hot path over the whole group: count, score every (child, parent set)
 * with at most max_parents parents (sharded), reduce to per-node best parent
 * sets, combine over the ranks.  Outputs (host) as obnsl_best_parents.
 * repack != 0 rebuilds the row bitmaps first (nothing cached is reused). */
obnsl_status obnsl_group_best_parents(obnsl_group* g, obnsl_score_kind kind, double ess,
                                      uint32_t max_parents, int repack, double* best_score,
                                      uint32_t* best_size, uint64_t* best_rank);
/* The complete score tables (the ranks' shards merged): tables_out[k-1] (host,
 * k * C(num_vars, k) doubles, or null to skip a size), k = 1..max_parents+1. */
obnsl_status obnsl_group_score_tables(obnsl_group* g, obnsl_score_kind kind, double ess,
                                      uint32_t max_parents, int repack, double* const* tables_out);
/* exhaustive_search(df, score_type, max_parents) with the scoring phase
 * (exhaustive_search.cpp:47-73) sharded over the group and the subset dynamic
 * program (:75-148, not shardable) on rank 0.  Outputs and errors as
 * obnsl_exhaustive_search. */
obnsl_status obnsl_group_exhaustive_search(obnsl_group* g, obnsl_score_kind kind, double ess,
                                           uint32_t max_parents, uint64_t* parents_out,
                                           double* total_score_out);

/* Replaces simulated_annealing(df, score_type, max_parents, max_iters,
 * init_temp, cooling_rate, is_deterministic, seed, num_chains)
 * (src/structure_learning/simulated_annealing.cpp:100-159; defaults of its
 * binding: bindings/submodules/structure_learning.cpp:19-30).  The reference
 * builds one contingency table over all rows per step (:77-81); here the local
 * scores come from the GPU (every parent set of at most max_parents parents in
 * one pass, or one table per step when that is cheaper) and the chains run on
 * host threads with the reference's generator, seeds (seed + c * 1234567) and
 * draw order.  num_chains == 0: one chain per host thread (OMP_NUM_THREADS if
 * set).  Outputs (host): num_parents_out[v] and the ascending parent ids
 * parents_out[v * max(max_parents,1) + j] of the best graph of the best chain;
 * best_score_out (may be null) = its total score.
 * max_parents >= num_vars -> OBNSL_ERR_INVALID_ARGUMENT "max_parents out of range"
 * (:109-111).  max_parents == 0 returns the empty graph (the reference's chain
 * does not honour the bound in that one case). */
obnsl_status obnsl_simulated_annealing(obnsl_data* d, obnsl_score_kind kind, double ess,
                                       uint32_t max_parents, uint64_t max_iters,
                                       double init_temp, double cooling_rate,
                                       int is_deterministic, uint64_t seed, uint32_t num_chains,
                                       uint32_t* parents_out, uint32_t* num_parents_out,
                                       double* best_score_out);

/* The same chains on score tables the caller already holds (host memory, no
 * GPU): tables[k-1][rank * k + i] = score(v_i | S \ {v_i}) for the size-k subset
 * S of colex rank `rank`, k = 1..max_parents+1 -- the layout obnsl_score_all
 * writes. */
obnsl_status obnsl_anneal_tables(uint32_t num_vars, uint32_t max_parents,
                                 const double* const* tables, uint64_t max_iters,
                                 double init_temp, double cooling_rate, int is_deterministic,
                                 uint64_t seed, uint32_t num_chains, uint32_t* parents_out,
                                 uint32_t* num_parents_out, double* best_score_out);

/* The score tables as a file (host, no GPU).  No counterpart in the reference,
 * which recomputes every score inside each exhaustive_search call
 * (src/structure_learning/exhaustive_search.cpp:61-73); consumers of the tables
 * (obnsl_anneal_tables, the subset dynamic program) can reload them instead.
 * `tables` as obnsl_anneal_tables takes them.  Little-endian; the layout is
 * documented in openbnsl_amd/csrc/score_file.cpp. */
obnsl_status obnsl_scores_save(const char* path, uint32_t num_vars, uint64_t num_rows,
                               const uint32_t* cardinalities, obnsl_score_kind kind, double ess,
                               uint32_t max_parents, const double* const* tables);
/* Header of such a file; every output may be null. */
obnsl_status obnsl_scores_info(const char* path, uint32_t* num_vars, uint64_t* num_rows,
                               uint32_t* cardinalities, uint32_t cardinalities_capacity,
                               obnsl_score_kind* kind, double* ess, uint32_t* max_parents);
/* The table of subset size k: k * C(num_vars, k) doubles into scores_out. */
obnsl_status obnsl_scores_load(const char* path, uint32_t k, double* scores_out,
                               uint64_t capacity);

/* colex rank <-> subset helpers (host, no GPU). */
uint64_t obnsl_binomial(uint32_t n, uint32_t k);
obnsl_status obnsl_unrank(uint64_t rank, uint32_t k, uint32_t* var_ids_out);
uint64_t obnsl_rank(const uint32_t* var_ids, uint32_t k);

/* Timing of the counting kernels (no counterpart in the reference).  Off by
 * default: a pair of timing events around a launch costs the stream a few
 * microseconds, which a short pass notices (about 2 % of one over 37 variables
 * and 10^6 rows).  obnsl_kernel_timing(d, n): n = 0 off; n >= 1: the count
 * launches of every n-th set of row bitmaps (every n-th obnsl_data_pack) are
 * bracketed by HIP events on the stream they run on.
 * obnsl_kernel_time: device time spent in the counting kernel (the bit-GEMM)
 * of subset size k in the timed launches since the last reset: total
 * milliseconds and number of timed launches.  Waits for them if they are
 * still running.  reset != 0 clears the totals of every subset size after
 * reading. */
obnsl_status obnsl_kernel_timing(obnsl_data* d, uint32_t every_n);
obnsl_status obnsl_kernel_time(obnsl_data* d, uint32_t k, double* total_ms, uint32_t* launches,
                               int reset);

/* Shape of the last matrix-core count launch that was timed under subset size k (no
 * counterpart in the reference): info_out[0..5] = tiles, work units, workgroups, cells
 * the launch computed (tile rows x the 32-column blocks that hold a valid column),
 * cells it was asked for, and the number of subset sizes it counted (a short pass
 * counts every size 1..k in the one launch of size k).  All zero before any launch. */
obnsl_status obnsl_count_launch_info(obnsl_data* d, uint32_t k, uint64_t* info_out);

/* Which kernel counts the tables of subset size k on this data set (no
 * counterpart in the reference, whose one loop is contingency_table.h:52-55):
 * the bit-plane contraction on the matrix cores, the same on the vector ALU
 * (OBNSL_GEMM=valu), or the per-row histogram for tables with many cells. */
#define OBNSL_PATH_MFMA 0
#define OBNSL_PATH_VALU 1
#define OBNSL_PATH_HISTOGRAM 2
obnsl_status obnsl_count_path(obnsl_data* d, uint32_t k, int* path_out);

/* Calibration helper with no counterpart in the reference: one plain streaming
 * read (16 bytes per lane) of `bytes` bytes of device memory at d_src (16-byte
 * aligned), queued on hip_stream; d_sink is one device word the kernel may
 * write.  The caller times it (events on that stream) to obtain the read
 * bandwidth the memory system delivers to a kernel that does nothing else --
 * the achievable figure quoted beside the nominal HBM peak. */
obnsl_status obnsl_stream_read(const void* d_src, uint64_t bytes, void* hip_stream, uint32_t* d_sink);

#ifdef __cplusplus
}
#endif
#endif /* OBNSL_ENGINE_H */
