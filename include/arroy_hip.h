/*
 * arroy_hip.h — C ABI of libarroy_hip.so: the MI355X (gfx950) implementation of arroy's
 * distance-kernel hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * arroy (Rust, /root/reference) has no FFI of its own: the boundary of the hot path is
 * the crate-internal `Distance` trait (src/distance/mod.rs:40-124) driven by three
 * batched loops.  Per-pair FFI is useless for a GPU, so every entry point below replaces
 * one *loop* (or one storage-to-memory staging step) of the reference; the citation after
 * each declaration names the reference code it stands in for.  INTEGRATION.md shows the
 * `extern "C"` block and the call-site patches a maintainer would add to arroy.
 *
 * Conventions
 *   - every function returns an `ah_status` (0 = ok); no C++ exception crosses the ABI
 *     (the reference catches worker panics at src/writer.rs:799-827);
 *   - `ah_last_error()` returns the thread's last error text (-> Error::Panic(String),
 *     src/error.rs:84-85);
 *   - the library never keeps a host pointer past the call that received it (LMDB pages
 *     move after writes, src/writer.rs:512-513); output buffers are caller-allocated;
 *   - a finalized dataset is immutable and may be used by any number of host threads
 *     concurrently (Reader is Sync); `ah_build_forest` is single-caller per dataset;
 *   - results never depend on thread / stream / workgroup scheduling.
 *
 * Numerics contract: every f32 result is bit-identical to arroy's x86-64 AVX+FMA tier
 * (src/spaces/simple_avx.rs) for dims >= 32, its SSE tier for 16 <= dims < 32
 * (src/spaces/simple_sse.rs) and its scalar tier below (src/spaces/simple.rs:49-51,81-83).
 */
#ifndef ARROY_HIP_H
#define ARROY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AH_ABI_VERSION 7   /* v2: ah_node.tree is 32 bits, ah_build_options.margin_mode, ah_build_stats.margin_mode_launches,
                                  ah_last_error_detail, ah_dataset_replicate, ah_dataset_upload_flush
                              v3: AH_MARGIN_DENSE_MFMA, ah_build_stats.dense_launches / dense_columns (appended)
                              v4: ah_forest_digest, ah_tuning_set / _get / _reset, ah_debug_launch_coverage,
                                  ah_build_stats.rows_* / screen8_* / screen_unavailable (appended)
                              v5: ah_search_stats / ah_index_search_stats, ah_build_options.max_host_threads (appended),
                                  ah_host_cache_trim, ah_device_cache_trim, ah_dataset_reserve_build, ah_synth_rows_host, ah_build_forest_stream (the node sink during the
                                  build), ah_build_stats.seconds_setup / seconds_after_device / host_blob_recycled
                              v6: ah_debug_dense_tiles, ah_device_cache_stats, ah_forest_digest_keyed, ah_search_stats.descent_block
                              v7: ah_build_stats.seconds_reserve / seconds_reserve_wait (appended), ah_rerank_stats /
                                  ah_dataset_rerank_stats, ah_search_stats.rerank_screened8 / screen8_retried_chunks / descent_multi, AH_SYNTH_CLUSTERED / AH_SYNTH_LOW_RANK (arroy_hip_policy.h),
                                  ah_dataset_replicate falls back to a copy through pinned host memory when the two devices
                                  have no peer access */

/* every entry point is exported from the shared object (it is built with -fvisibility=hidden) */
#if defined(__GNUC__)
#define AH_API __attribute__((visibility("default")))
#else
#define AH_API
#endif

/* arroy::distances (src/lib.rs:145-150).  The value is also the tag used in files/tests. */
typedef enum ah_metric {
    AH_EUCLIDEAN = 0,              /* src/distance/euclidean.rs                 header {bias:f32}           */
    AH_MANHATTAN = 1,              /* src/distance/manhattan.rs                 header {bias:f32}           */
    AH_COSINE = 2,                 /* src/distance/cosine.rs                    header {norm:f32}           */
    AH_DOT_PRODUCT = 3,            /* src/distance/dot_product.rs               header {extra_dim,norm:f32} */
    AH_BQ_EUCLIDEAN = 4,           /* src/distance/binary_quantized_euclidean.rs header {bias:f32}          */
    AH_BQ_MANHATTAN = 5,           /* src/distance/binary_quantized_manhattan.rs header {bias:f32}          */
    AH_BQ_COSINE = 6               /* src/distance/binary_quantized_cosine.rs    header {norm:f32}          */
} ah_metric;

/* Mapping onto arroy::Error (src/error.rs:6-85). */
typedef enum ah_status {
    AH_OK = 0,
    AH_ERR_INVALID_DIMENSION = 1,  /* Error::InvalidVecDimension  (error.rs:17-23)              */
    AH_ERR_CANCELLED = 2,          /* Error::BuildCancelled       (error.rs:54-55)              */
    AH_ERR_DEVICE = 3,             /* HIP runtime failure      -> Error::Panic(ah_last_error()) */
    AH_ERR_OUT_OF_MEMORY = 4,      /* host or HBM allocation   -> Error::Panic                  */
    AH_ERR_INVALID_ARGUMENT = 5,   /* contract violation       -> Error::Panic                  */
    AH_ERR_MISSING_ITEM = 6,       /* Error::MissingKey           (error.rs:38-46)              */
    AH_ERR_NOT_FINALIZED = 7,      /* dataset used before ah_dataset_finalize -> Error::Panic   */
    AH_ERR_NEED_PREPROCESS = 8     /* DotProduct dataset searched/built before ah_preprocess_dot */
} ah_status;

typedef struct ah_dataset ah_dataset;   /* opaque: the HBM-resident image of ImmutableLeafs */
typedef struct ah_forest ah_forest;     /* opaque: host-side result of one forest build     */

/* Size in bytes of D::Header for the metric (4, or 8 for DotProduct). src/distance/<metric>.rs `Header`. */
AH_API size_t ah_header_size(int metric);
/* Size in bytes of one stored vector: 4*dims for f32 codecs (src/unaligned_vector/f32.rs),
 * ceil(dims/64)*8 for the 1-bit codec (src/unaligned_vector/binary_quantized.rs:80-91). */
AH_API size_t ah_vector_size(int metric, uint32_t dimensions);

AH_API int ah_abi_version(void);
AH_API int ah_device_count(int *out_count);
/* Text of the calling thread's last failure ("" if none).  Never NULL. */
AH_API const char *ah_last_error(void);
/* Structured part of the calling thread's last failure, so that the Rust side can rebuild the typed variants of
 * arroy::Error instead of a string: AH_ERR_INVALID_DIMENSION -> Error::InvalidVecDimension { expected, received }
 * (src/error.rs:17-23), AH_ERR_MISSING_ITEM -> Error::MissingKey { mode: "Item", item } (src/error.rs:58-67). */
typedef struct ah_error_detail {
    int status;          /* the ah_status the failing call returned (AH_OK if the thread has not failed yet) */
    uint32_t item;       /* AH_ERR_MISSING_ITEM: the item id that does not exist                             */
    uint64_t expected;   /* AH_ERR_INVALID_DIMENSION: dimensions (or record bytes) of the dataset            */
    uint64_t received;   /* AH_ERR_INVALID_DIMENSION: what the caller passed                                 */
} ah_error_detail;
AH_API int ah_last_error_detail(ah_error_detail *out);

/* ------------------------------------------------------------------------------------------
 * Dataset = what `ImmutableLeafs::new` builds (src/parallel.rs:271-293): instead of a map
 * id -> pointer into an LMDB page, the records are split into three arrays in HBM (ids,
 * headers, 128-byte-aligned vector rows) in ascending item-id order.
 * ---------------------------------------------------------------------------------------- */

/* `capacity` = number of items that will be uploaded (exact or an upper bound). */
AH_API int ah_dataset_create(int metric, uint32_t dimensions, uint64_t capacity, int device, ah_dataset **out);

/* Stage `n` stored item records `[0u8][header][vector]` (src/node.rs:224-228,252-258) straight
 * from LMDB pages.  record_ptrs[i] may be arbitrarily (mis)aligned; every record is
 * `record_len` bytes (ImmutableLeafs asserts a constant length, src/parallel.rs:286).
 * Ids must be strictly ascending within and across calls (RoaringBitmap iteration order,
 * src/parallel.rs:283).  Records are copied into pinned staging buffers and sent with
 * hipMemcpyAsync, double-buffered; the pointers are not used after return. */
AH_API int ah_dataset_upload_records(ah_dataset *ds, const uint32_t *item_ids, const uint8_t *const *record_ptrs,
                              size_t record_len, size_t n);

/* `Writer::add_item` for a batch (src/writer.rs:380-394): f32 vectors, row-major n x dims; the
 * library applies the codec (`UnalignedVector::from_slice`) and `D::new_header` on device. */
AH_API int ah_dataset_upload_vectors(ah_dataset *ds, const uint32_t *item_ids, const float *vectors, size_t n);

/* Benchmark harness only: materialise `n_items` synthetic items (ids 0..n-1) directly in HBM with
 * the generator of arroy_hip_policy.h, then codec + new_header as in ah_dataset_upload_vectors. */
AH_API int ah_dataset_fill_synthetic(ah_dataset *ds, uint64_t seed, int distribution, uint64_t n_items);

/* Uploads are asynchronous: a call returns once the records are copied out of the caller's pages into the pinned
 * staging ring, the last DMA transfers may still be in flight.  ah_dataset_finalize (and every call that reads the
 * dataset) waits for them; ah_dataset_upload_flush does only that and reports a failed transfer. */
AH_API int ah_dataset_upload_flush(ah_dataset *ds);
/* DotProduct only: records staged with ah_dataset_upload_records carry whatever header the database holds.  Items
 * written by `Writer::add_item` have {extra_dim: 0, norm: 0} until `DotProduct::preprocess` ran
 * (src/distance/dot_product.rs:119-165), so such a dataset still needs ah_preprocess_dot; a caller that staged an
 * already built (preprocessed) database says so here (src/writer.rs:964-976 runs preprocess inside every build). */
AH_API int ah_dataset_set_preprocessed(ah_dataset *ds, int preprocessed);
/* Freeze the dataset (build the id -> row index).  Required before any query/build call. */
AH_API int ah_dataset_finalize(ah_dataset *ds);
/* A replica of `src` on another GPU of the node, copied device to device (xGMI) instead of staged again over PCIe:
 * what the one-tree-batch-per-GPU build needs (`Writer::build` shares ONE ImmutableLeafs between all its tasks,
 * src/writer.rs:530,556-591).  The replica has its own handle and lifetime; destroy it with ah_dataset_destroy. */
AH_API int ah_dataset_replicate(ah_dataset *src, int device, ah_dataset **out);
/* Optional, right after ah_dataset_create: obtain — on a helper thread, while the records are staged over PCIe — the device
 * memory the first `n_trees`-tree build of this dataset will ask for (the binary16 / int8 copies of the rows, the build's
 * scratch) and park it in the library's device cache.  Fresh HBM can cost the driver 20+ ms per GB (a box whose memory was
 * just released by another process scrubs it); `Writer::build` knows its tree count before it collects the items
 * (src/writer.rs:518-519).  split_after 0 = dimensions.  Never fails for lack of memory: the build then allocates itself. */
AH_API int ah_dataset_reserve_build(ah_dataset *ds, uint32_t n_trees, uint32_t split_after);
AH_API int ah_dataset_len(const ah_dataset *ds, uint64_t *out_n_items);
/* `Reader::item_vector` (src/reader.rs:266-276): decoded f32 vector (dims floats; +-1.0 for BQ). */
AH_API int ah_dataset_item_vector(ah_dataset *ds, uint32_t item_id, float *out_vector);
/* Copy stored headers of rows [first_row, first_row+n) to the host (header_size bytes each): what the
 * host needs to rewrite LMDB after `ah_preprocess_dot`. */
AH_API int ah_dataset_read_headers(ah_dataset *ds, uint64_t first_row, uint64_t n, void *out_headers);
AH_API int ah_dataset_destroy(ah_dataset *ds);

/* `DotProduct::preprocess` (src/distance/dot_product.rs:119-165): max norm over all items, then
 * header.norm = max^2, header.extra_dim = sqrt(max^2 - |v|^2) for every item, on device. */
AH_API int ah_preprocess_dot(ah_dataset *ds, float *out_max_norm);

/* ------------------------------------------------------------------------------------------
 * Search side (src/reader.rs:376-400, 607-640)
 * ---------------------------------------------------------------------------------------- */

/* Raw batched `D::built_distance(query, item)` (non-normalized), `QueryBuilder::by_vector`
 * semantics for the query (src/reader.rs:64-75: codec + `D::new_header`).  `item_ids == NULL`
 * scans rows 0..n-1 of the dataset in id order; otherwise one distance per listed id. */
AH_API int ah_distances_by_vector(ah_dataset *ds, const float *query, const uint32_t *item_ids, size_t n, float *out);
/* Same with the query taken from a stored item (`by_item`, src/reader.rs:46-51). */
AH_API int ah_distances_by_item(ah_dataset *ds, uint32_t query_item, const uint32_t *item_ids, size_t n, float *out);

/* The re-rank loop + `median_based_top_k` + `D::normalized_distance` (src/reader.rs:381-399):
 * `sorted_ids` ascending and unique (src/reader.rs:378-379) or NULL for "all items".  Writes
 * min(k, n) pairs ordered by (OrderedFloat(distance), id) ascending. */
AH_API int ah_rerank_by_vector(ah_dataset *ds, const float *query, const uint32_t *sorted_ids, size_t n, size_t k,
                        uint32_t *out_ids, float *out_distances, size_t *out_n);
AH_API int ah_rerank_by_item(ah_dataset *ds, uint32_t query_item, const uint32_t *sorted_ids, size_t n, size_t k,
                      uint32_t *out_ids, float *out_distances, size_t *out_n);
/* Many queries in one submission (candidate lists concatenated; list q is
 * ids[offsets[q] .. offsets[q+1])).  Outputs are n_queries x k, short lists padded with id
 * 0xFFFFFFFF / distance NaN; out_counts[q] = min(k, len_q) says how many are real (0xFFFFFFFF is
 * also a legal item id, src/tests/writer.rs:161-179).  Results are identical to n_queries single
 * calls; a submission with >= 2 candidates per stored row is evaluated row-major on the device
 * (every row leaves HBM once per submission). */
AH_API int ah_rerank_batch(ah_dataset *ds, const float *queries, size_t n_queries, const uint32_t *ids,
                    const uint64_t *offsets, size_t k, uint32_t *out_ids, float *out_distances,
                    uint32_t *out_counts);

/* ABI v7.  Where the wall time of ah_rerank_batch went, summed over the calls of every thread since the last reset — kept
 * only while the tunable AH_RERANK_TIMING is 1 (ah_tuning_set; two extra events and a few clock reads per submission).  A
 * submission that is slow on one box and not on another shows here whether the host (prep / ids / enqueue) or the device
 * (sync_wait) paid: bench.py prints it next to the re-rank figures. */
typedef struct ah_rerank_stats {
    uint64_t calls;               /* ah_rerank_batch calls (their sub-batches summed)                                     */
    uint64_t queries, candidates;
    double seconds_wall;          /* entry -> return                                                                      */
    double seconds_prep;          /* entry -> first enqueue: argument checks, segment / tile tables, scratch, query copy  */
    double seconds_ids;           /* host copies of the candidate ids into pinned memory (they overlap the device's work) */
    double seconds_enqueue;       /* launching copies and kernels (the calls themselves, not their execution)             */
    double seconds_sync_wait;     /* blocked in hipStreamSynchronize: the device still had work when the host was done    */
    double seconds_device_span;   /* HIP events: first enqueue of the submission -> its stream idle                       */
    /* how the certified top-k screen went (kept whatever AH_RERANK_TIMING says) */
    uint64_t queries_screened;    /* queries answered through the screen (int8 or binary16 rows first, f32 for the survivors) */
    uint64_t survivors;           /* ... candidates of theirs evaluated in f32                                            */
    uint64_t chunks_int8;         /* sub-batches whose screen started on the int8 copy of the rows and stayed there       */
    uint64_t chunks_int8_retried; /* ... that left more survivors than the selection holds and ran again on binary16 rows */
} ah_rerank_stats;
AH_API int ah_dataset_rerank_stats(ah_dataset *ds, ah_rerank_stats *out, int reset);

/* ------------------------------------------------------------------------------------------
 * Build side (src/writer.rs:1193-1233, 1398-1531; src/distance/mod.rs:126-223)
 * ---------------------------------------------------------------------------------------- */

#define AH_SPLIT_SAMPLES 12   /* choose_two + 10 x choose (src/distance/mod.rs:139,149-152) */

/* The margin loop (src/writer.rs:1201-1207,1424-1431,1494-1501): `D::side(normal, item)` for every
 * listed item.  `normal_vector` is in the metric's codec (ah_vector_size bytes), `normal_header`
 * is D::Header.  Bit i (LSB first) of side_bits = 1 when item i goes Right
 * (`margin.is_sign_positive()`, src/distance/mod.rs:103-110); (n+7)/8 bytes. Optionally also
 * returns the margins themselves (out_margins may be NULL). */
AH_API int ah_split_sides(ah_dataset *ds, const void *normal_vector, const void *normal_header,
                   const uint32_t *sorted_ids, size_t n, uint8_t *side_bits, uint64_t *out_n_left,
                   float *out_margins);

/* The margins of the tree descent (src/reader.rs:366-369): `D::margin(&normal_i, query_leaf)` for many split-plane
 * normals at once.  `normals` is a dataset whose ITEMS are normals (stage them with ah_dataset_upload_records:
 * header = the normal's header); `leaf_vector` / `leaf_header` is the query leaf in the metric's codec
 * (by_vector: codec + D::new_header; by_item: the stored item).  item_ids NULL = every normal in row order. */
AH_API int ah_margins(ah_dataset *normals, const void *leaf_vector, const void *leaf_header, const uint32_t *item_ids,
                      size_t n, float *out_margins);

/* `D::create_split` (src/distance/{euclidean.rs:55-77,cosine.rs:73-85,dot_product.rs:98-113,...}) with the
 * randomness supplied by the host: sample_ids[0..1] = choose_two, [2..11] = the ten `choose` draws
 * (the RNG policy stays in the caller, as with `R: Rng`). */
AH_API int ah_create_split(ah_dataset *ds, const uint32_t sample_ids[AH_SPLIT_SAMPLES], void *out_normal_vector,
                    void *out_normal_header);

typedef void (*ah_progress_fn)(void *user, uint32_t level, uint64_t nodes_done, uint64_t items_routed);

/* How the margin pass of a level (src/writer.rs:1201-1207 for every pending node of every tree) is laid out on the
 * device.  Every mode produces the same forest bit for bit; AUTO picks per level with a cost model.  The other values
 * pin one kernel family for every level where it is legal (f32 metrics, dimensions >= 32, full-dataset trees, >= 2
 * trees; LDS modes additionally need the group's normals of the level to fit in LDS) and use the node-major kernel
 * elsewhere: they exist so that tests can compare EVERY kernel instantiation with the oracle, and for A/B timing. */
typedef enum ah_margin_mode {
    AH_MARGIN_AUTO = 0,
    AH_MARGIN_NODE_MAJOR = 1,      /* one tree at a time, tiles of one node, its normal in LDS                       */
    AH_MARGIN_ROWS_2 = 2,          /* row-major: one pass over the rows serves 2 / 4 / 8 / 16 trees, normals via L2   */
    AH_MARGIN_ROWS_4 = 4,
    AH_MARGIN_ROWS_8 = 8,
    AH_MARGIN_ROWS_16 = 16,
    AH_MARGIN_ROWS_LDS_8 = 0x108,  /* row-major, all normals of a group of 8 / 16 trees resident in LDS               */
    AH_MARGIN_ROWS_LDS_16 = 0x110,
    /* Dense screen on the matrix units (ABI v3): the binary16 screen values of ALL (row, node) pairs of a level as one
     * matrix product rows x normals^T (v_mfma_f32_32x32x16_f16), of which every row uses one entry per tree; pairs the
     * screen cannot decide are recomputed in the reference's f32 arithmetic, so the forest is the same bit for bit.
     * Cheaper than the row-major passes while a tree has few nodes (top ~6 levels); needs the screen, i.e. it is
     * ignored (node-major instead) together with AH_MARGIN_EXACT_ONLY and beyond AH_DENSE_MAX_COLS nodes per level. */
    AH_MARGIN_DENSE_MFMA = 0x200,
    /* Flag, OR-ed into any of the above: evaluate every margin in the reference's f32 arithmetic only.  Without it the
     * f32 metrics first evaluate a *certified screen*: the same dot product on a binary16 shadow copy of the rows and
     * of the level's normals (half the bytes), with a rigorous bound E on |screen - reference f32 margin| derived
     * from the measured quantisation errors ("certified screening"); only the SIGN of a margin is ever
     * used (`D::side`, src/distance/mod.rs:103-110), so |screen| > E decides the side and every other pair (about 1 %)
     * is recomputed with the reference arithmetic in the same kernel.  Sides — hence forests — are identical. */
    AH_MARGIN_EXACT_ONLY = 0x1000
} ah_margin_mode;

typedef struct ah_build_options {
    uint32_t n_trees;              /* trees built by THIS call (the caller shards trees over GPUs)   */
    uint32_t split_after;          /* 0 = dimensions (src/writer.rs:474-477)                         */
    const uint64_t *tree_seeds;    /* n_trees seeds (src/writer.rs:575: one RNG per root task)       */
    const volatile int *cancel;    /* polled while a level runs (the reference polls per node and per item,
                                      src/writer.rs:1178,1196): non-zero -> launches that have not started yet drain
                                      without work, the call returns AH_ERR_CANCELLED when the level's stream is idle
                                      (at most one level: <= 0.25 s at 10M x 768 x 100 trees) */
    ah_progress_fn progress;       /* may be NULL (src/writer.rs:53-69 SubStep).  Called once per digested level with
                                      `level` = depth + 1: ascending, except at the bottom of a build whose last levels
                                      run group of trees by group of trees (AH_BUILD_TAIL_GROUPS), where every group
                                      reports its own last levels; nodes_done / items_routed never decrease */
    void *progress_user;
    uint32_t max_trees_in_flight;  /* 0 = as many as HBM allows                                      */
    uint32_t margin_mode;          /* ah_margin_mode; 0 = AH_MARGIN_AUTO                             */
    uint32_t max_host_threads;     /* ABI v5: host threads this build may keep busy for its output path (page commits,
                                      bounce copies, node list); 0 = 8.  The reference gives a build its rayon pool
                                      (src/writer.rs:538-548); eight concurrent builds of an 8-GPU node share the host */
    uint32_t reserved0;            /* 0 */
} ah_build_options;

/* Whole-forest build: `make_tree_in_file` for every tree (src/writer.rs:556-591,1167-1261) with the
 * randomness policy of arroy_hip_policy.h.  Level-synchronous on device; nothing is visible to the
 * caller until the whole forest exists (src/writer.rs:597-607). */
AH_API int ah_build_forest(ah_dataset *ds, const ah_build_options *options, ah_forest **out);

/* The same build over caller-given item subsets: `incremental_index_large_descendant` (src/writer.rs:660-739), i.e.
 * `make_tree_in_file` for every Descendants node that grew beyond split_after during an incremental insert.  Tree t
 * of the result covers the ascending id list item_ids[offsets[t] .. offsets[t+1]) and uses options->tree_seeds[t];
 * options->n_trees = number of subsets.  A subset that fits in one Descendants node yields a one-node tree. */
AH_API int ah_build_subtrees(ah_dataset *ds, const ah_build_options *options, const uint32_t *item_ids,
                             const uint64_t *offsets, ah_forest **out);

enum { AH_NODE_DESCENDANTS = 1, AH_NODE_SPLIT = 2 };   /* node tags, src/node.rs:216-241 */

typedef struct ah_node {
    uint8_t kind;           /* AH_NODE_DESCENDANTS | AH_NODE_SPLIT                                   */
    uint8_t has_normal;     /* 0 = `normal: None` (random split fallback, src/writer.rs:1220-1227)   */
    uint16_t reserved;      /* 0                                                                    */
    uint32_t tree;          /* tree index inside this forest (arroy's target_n_trees, src/writer.rs:1371-1380,
                               can exceed 65 535, hence 32 bits since ABI v2)                         */
    uint32_t left, right;   /* forest-local node indices (SPLIT)                                    */
    uint64_t offset;        /* SPLIT: byte offset into the normals blob; DESCENDANTS: first id index */
    uint32_t count;         /* DESCENDANTS: number of item ids; SPLIT: items under the node         */
    uint32_t depth;
} ah_node;

typedef struct ah_forest_view {
    uint32_t n_trees;
    uint64_t n_nodes;
    const uint32_t *roots;        /* n_trees forest-local node indices                               */
    const ah_node *nodes;         /* n_nodes, children before parents (post-order) inside each tree  */
    /* Normals: one fixed-size record per split node at byte `ah_node.offset` of `normals`.  Inside a record
     * the vector (metric codec, ah_vector_size bytes) sits at `normal_vector_offset` and D::Header
     * (ah_header_size bytes) at `normal_header_offset`; `normal_stride` may exceed their sum (the f32 metrics pad a
     * record to whole 128-byte lines, the padding is zero).  (The records are the device layout copied back
     * verbatim, so nothing is repacked on the host; the Rust side copies both parts into the LMDB value
     * `[2u8][left][right][header][vector]` anyway, src/node.rs:229-237.) */
    const uint8_t *normals;
    uint64_t normals_len;
    uint64_t normal_stride;
    uint64_t normal_vector_offset;
    uint64_t normal_header_offset;
    const uint32_t *descendants;  /* item ids, ascending inside each Descendants node                */
    uint64_t descendants_len;
} ah_forest_view;

typedef struct ah_build_stats {
    double seconds_total;         /* wall time of ah_build_forest                                    */
    double seconds_device;        /* HIP-event time of all kernels                                   */
    double seconds_margin;        /* HIP-event time of the margin/side kernel only                   */
    uint64_t margin_evaluations;  /* number of (item, node-visit) units incl. retries                */
    uint64_t margin_launches;
    uint64_t margin_row_passes;   /* row-major passes (each streams all rows once and serves several trees)  */
    uint64_t split_nodes, descendant_nodes, dummy_normals, retries;
    uint32_t levels;
    /* margin launches per kernel family: [0] node-major f32, [1] rows x2, [2] rows x4, [3] rows x8, [4] rows x16,
     * [5] LDS x8, [6] LDS x16, [7] node-major 1-bit */
    uint64_t margin_mode_launches[8];
    uint64_t screened_launches;   /* margin launches that ran the certified binary16 screen                  */
    uint64_t screen_fallbacks;    /* (item, node) pairs the screen could not decide (recomputed in f32)       */
    uint64_t screen_violations;   /* AH_SCREEN_VERIFY=1 only: decided pairs whose f32 side differs (must be 0) */
    uint64_t dense_launches;      /* ABI v3: levels whose first attempt ran as one MFMA product (AH_MARGIN_DENSE_MFMA)  */
    uint64_t dense_columns;       /* ABI v3: normals (columns) those products covered, summed over the levels          */
    /* ABI v4: which schedule variants of the row-major pass ran (kernel launches, not passes) */
    uint64_t rows_xcd_launches;   /* launches that pinned every (chunk, tree group) to one XCD                          */
    uint64_t rows_nt_launches;    /* ... of which streamed the rows with non-temporal loads                             */
    uint64_t rows_split_launches; /* extra launches of passes cut at the work-item limit of one dispatch                */
    uint64_t screen8_pairs;       /* (item, node) pairs that met the int8 first stage of the node-major screen          */
    uint64_t screen8_decided;     /* ... of which its first digit decided (768 bytes of a 768-d row)                    */
    uint64_t screen8b_decided;    /* ... and of the rest, decided by the rows' second int8 digit (768 more bytes); what is
                                     left goes on to the binary16 stage                                                  */
    uint32_t screen_unavailable;  /* the build wanted the screen but its copies could not be allocated: f32 arithmetic  */
    uint32_t tail_groups;         /* (was reserved) groups of trees the last big level and what followed it ran in, summed
                                     over the batches: their ids and normals left the device under the next group's
                                     kernels (AH_BUILD_TAIL_GROUPS); 0 = every level for all trees                      */
    /* ABI v5: where the wall time outside the kernels went (summed over the batches of the build) */
    double seconds_setup;         /* entry of a batch -> its first launch (device buffers, pinned memory, host blobs)    */
    double seconds_after_device;  /* last launch of a batch -> its return (ids' read-back, node list, teardown)          */
    uint64_t host_blob_recycled;  /* output blobs (normals, ids) taken committed from the pool of destroyed forests     */
    /* ABI v7: ah_dataset_reserve_build, as the first build after it saw it */
    double seconds_reserve;       /* run time of the helper thread that obtained the build's device memory under the staging */
    double seconds_reserve_wait;  /* ... and how long this build waited for it before it started (NOT in seconds_total)      */
} ah_build_stats;

AH_API int ah_forest_view_get(const ah_forest *forest, ah_forest_view *out);
/* 64-bit digest of the CONTENT of a forest, independent of how it was laid out in memory (batching, record padding):
 * per tree, in the node order of ah_forest_view (post-order), kind / has_normal / children / count / depth of every
 * node, header + vector bytes of every split plane, item ids of every Descendants node.  Two builds of the same
 * dataset with the same seeds must agree whatever the margin mode or tuning: bench.py and the GPU tests compare the
 * screened and the f32-only build of the 10M x 100-tree forest this way.  out_per_tree: n_trees values or NULL. */
AH_API int ah_forest_digest(const ah_forest *forest, uint64_t *out_per_tree, uint64_t *out_total);
/* ABI v6: the per-tree digests with the caller's key in place of the tree's index inside THIS forest — e.g. the tree's
 * index in the whole index when the forest is one GPU's share of it (trees t = device mod G, src/writer.rs:556-591): the
 * digest of tree t is then the same whichever share, batch or device built it, and the union of the shares' digests can be
 * compared with the digests of a one-GPU build (bench.py does, `build_10m.union`).  tree_keys / out_per_tree: n_trees values. */
AH_API int ah_forest_digest_keyed(const ah_forest *forest, const uint64_t *tree_keys, uint64_t *out_per_tree);
AH_API int ah_forest_stats(const ah_forest *forest, ah_build_stats *out);
/* Node sink in the shape `TmpNodes::put` expects (src/parallel.rs:130-147): children first, parent
 * last (post-order), per tree. `payload`: SPLIT -> the normal record of ah_forest_view (vector at
 * normal_vector_offset, header at normal_header_offset) or NULL for `normal: None`; DESCENDANTS -> u32 ids. */
typedef int (*ah_node_sink_fn)(void *user, uint32_t tree, uint32_t node, uint8_t kind, uint32_t left,
                               uint32_t right, const void *payload, size_t payload_len);
AH_API int ah_forest_visit(const ah_forest *forest, ah_node_sink_fn sink, void *user);
AH_API int ah_forest_destroy(ah_forest *forest);

/* ------------------------------------------------------------------------------------------
 * Streaming build (ABI v5): the node sink DURING the build.  `Writer::build` hands every finished tree node to
 * `TmpNodes::put` (src/parallel.rs:130-147) and drains the tmp files into LMDB at the end (src/writer.rs:597-607);
 * nothing in the reference ever needs the whole forest in one piece.  ah_build_forest_stream therefore never
 * materialises it: the split planes of a level (as soon as the level is final) and the item-id lists (after the last
 * level) travel device -> pinned ring -> `sink`, batch by batch, while the next levels are computed.  The host keeps
 * 64 MiB of pinned memory and the node table instead of the 9.4 GB a 10M x 768 x 100-tree forest occupies.
 *
 * Order: breadth-first — a split node arrives BEFORE its children (its record names their ids; the reference's ids are
 * arbitrary too, src/parallel.rs:239-254, and TmpNodes is an append-only log keyed by id).  The Descendants nodes of a
 * tree arrive after all of its split nodes, in ascending (tree, position) order over the whole call: after the last
 * level, or — when the last big level and what follows it run in groups of trees (AH_BUILD_TAIL_GROUPS, the default
 * for builds whose id lists are worth it) — a group's Descendants as soon as that group is complete, i.e. between the
 * split planes of the groups after it.  `id`s are unique over the whole call, dense from 0, assigned in creation
 * order (a node's children are id-consecutive: left, left + 1).  `sink` is called from ONE library thread, one batch at
 * a time; `nodes` and `payload` are valid only during the call (the payload is the pinned DMA buffer itself: copy or
 * encode out of it).  A non-zero return stops the build: AH_ERR_CANCELLED.
 * ---------------------------------------------------------------------------------------- */
typedef struct ah_stream_node {
    uint32_t id;             /* this node                                                                      */
    uint32_t tree;           /* tree index inside this build                                                   */
    uint8_t kind;            /* AH_NODE_SPLIT | AH_NODE_DESCENDANTS                                            */
    uint8_t has_normal;      /* SPLIT: 0 = `normal: None` (src/writer.rs:1220-1227): no payload                 */
    uint16_t reserved;
    uint32_t left, right;    /* SPLIT: ids of the children                                                     */
    uint32_t count;          /* DESCENDANTS: item ids in the payload; SPLIT: items under the node              */
    uint32_t depth;
    uint64_t payload_offset; /* into ah_node_batch.payload.  SPLIT: the normal record (vector at normal_vector_offset,
                                D::Header at normal_header_offset, as in ah_forest_view); DESCENDANTS: `count` u32 item
                                ids, ascending                                                                  */
} ah_stream_node;

typedef struct ah_node_batch {
    uint32_t kind;           /* all nodes of a batch are of one kind                                           */
    uint32_t level;          /* SPLIT: depth of the nodes; DESCENDANTS: 0                                      */
    uint64_t n_nodes;
    const ah_stream_node *nodes;
    const uint8_t *payload;
    uint64_t payload_len;
    uint64_t normal_stride, normal_vector_offset, normal_header_offset;
} ah_node_batch;

typedef int (*ah_node_batch_fn)(void *user, const ah_node_batch *batch);

/* `options` as for ah_build_forest.  out_roots: options->n_trees node ids (the roots, in tree order); out_stats may be
 * NULL.  The forest is the one ah_build_forest builds for the same seeds, node for node (only the numbering differs:
 * ah_forest_view numbers children before parents). */
AH_API int ah_build_forest_stream(ah_dataset *ds, const ah_build_options *options, ah_node_batch_fn sink, void *user,
                                  uint32_t *out_roots, ah_build_stats *out_stats);

/* ------------------------------------------------------------------------------------------
 * Whole search on device (src/reader.rs:317-401): the forest mirrored in HBM next to its dataset, best-first
 * descent + candidate collection + sort/dedup + re-rank + top-k for a batch of queries in one call.
 * Node identity (the tie-break of the reference's BinaryHeap<(OrderedFloat<f32>, NodeId)>) is the forest-local
 * node index; a host that hands out global node ids in that same order (children before parents, as
 * INTEGRATION.md does) preserves every tie-break.
 * ---------------------------------------------------------------------------------------- */
typedef struct ah_index ah_index;

/* Upload the forest (nodes, split-plane normals, descendants) to the dataset's device.  The ah_forest may be
 * destroyed afterwards; the dataset must outlive the index. */
AH_API int ah_index_create(ah_dataset *ds, const ah_forest *forest, ah_index **out);
/* Same from caller-owned arrays in the ah_forest_view shape (e.g. tree nodes decoded from LMDB by `Reader::open`):
 * validated, copied to the device, not retained. */
AH_API int ah_index_create_from_view(ah_dataset *ds, const ah_forest_view *view, ah_index **out);
AH_API int ah_index_destroy(ah_index *index);

/* `QueryBuilder::by_vector` (queries = nq x dims f32, query_items = NULL) or `by_item` (queries = NULL,
 * query_items = nq ids) with `search_k` (0 = count * n_trees, src/reader.rs:330), `oversampling`
 * (0 = D::DEFAULT_OVERSAMPLING) and optional `candidates` (have_filter != 0: ascending ids).
 * Any count (`Reader::nns(count)`); counts above 2048 leave the batched top-k for the single-query kernels, query by
 * query.  Outputs nq x count, padded with id 0xFFFFFFFF / NaN; out_counts[q] = results of query q. */
AH_API int ah_search_batch(ah_index *index, const float *queries, const uint32_t *query_items, size_t nq, size_t count,
                           size_t search_k, size_t oversampling, const uint32_t *filter_sorted, size_t n_filter,
                           int have_filter, uint32_t *out_ids, float *out_distances, uint32_t *out_counts);

/* Which device path served the searches of an index (ABI v5).  ah_search_batch has several tiers — four descents, three
 * dedup paths, two re-rank paths — that return the same bits; these counters are how a caller (and the parity tests)
 * can tell that the intended one ran instead of a silent fall-back.  All values count since the index was created or
 * since the last call with reset != 0. */
typedef struct ah_search_stats {
    uint64_t calls;                 /* ah_search_batch calls that reached the device                                   */
    uint64_t chunks;                /* sub-batches they were cut into                                                  */
    uint64_t queries;
    /* the descent that produced a query's candidates (src/reader.rs:341-374); these four and `descent_block` below sum to
     * `queries` */
    uint64_t descent_wave_small;    /* one wave per query, 256 queue entries / 64 leaves per octet                     */
    uint64_t descent_wave_big;      /* ... its second pass, 1024 / 128                                                 */
    uint64_t descent_octet_lds;     /* one octet per query, the sequential queue in LDS                                */
    uint64_t descent_octet_global;  /* ... the queue in global memory (capacity = number of nodes)                     */
    /* nns.sort_unstable(); nns.dedup() (src/reader.rs:378-379), in queries */
    uint64_t dedup_flag_bitmap;     /* leaf tiles: duplicates flagged through one bit per id in LDS                    */
    uint64_t dedup_flag_hash;       /* leaf tiles: ... through a hash set of the candidates in LDS (big id spaces)     */
    uint64_t dedup_sorted_bitmap;   /* sorted path: the LDS bitmap walked in order                                     */
    uint64_t dedup_sort_lds;        /* sorted path: bitonic sort in LDS                                                */
    uint64_t dedup_sort_global;     /* sorted path: bitonic sort in global memory                                      */
    /* the re-rank (src/reader.rs:381-399), in queries */
    uint64_t rerank_tiles;          /* rows of a leaf x the queries that reached it                                    */
    uint64_t rerank_sorted;         /* from the sorted candidate lists (the kernels of ah_rerank_batch)                */
    uint64_t tile_visits;           /* (query, leaf) pairs the leaf tiles served                                       */
    uint64_t tile_units_16;         /* work units of 9..16 visits of one leaf (operands through the LDS ring)          */
    uint64_t tile_units_8;          /* ... of 5..8 visits (LDS ring)                                                   */
    uint64_t tile_units_4;          /* ... of 1..4 visits (registers)                                                  */
    /* sub-batches the leaf tiles handed to the sorted path, and why (one chunk may count under several reasons) */
    uint64_t fallback_chunks;
    uint64_t fallback_non_finite;   /* a non-finite distance: src/reader.rs:611-621 looks at positions                 */
    uint64_t fallback_select;       /* the selection / the hash set did not fit its LDS buffers                        */
    uint64_t fallback_queue;        /* a queue outgrew LDS (the sorted path has the global-memory queue)               */
    uint64_t fallback_visits;       /* more leaf visits than the visit buffer holds                                    */
    uint64_t fallback_launch;       /* the runtime rejected a launch of the tile path                                  */
    uint64_t filtered_queries;      /* queries under a candidate filter                                                */
    uint64_t leaf_kept_passes;      /* passes over every Descendants id for |leaf & candidates| (once per filtered call) */
    /* certified top-k screen of the tile re-rank (Cosine / DotProduct): candidates evaluated on the binary16 copy of the
     * rows first, only those whose proven distance interval reaches the top `count` in f32 */
    uint64_t rerank_screened;       /* queries whose top-k went through the screen                                      */
    uint64_t screen_survivors;      /* candidates of those queries evaluated in f32 (the rest: 2 x dims bytes each)     */
    uint64_t descent_block;         /* ABI v6: one block (32 octets, one tree each) per query: submissions of few queries;
                                       a fifth tier of the descent — the five sum to `queries` */
    /* ABI v7: the int8 copy of the rows as the first stage of that screen (big submissions; 1 byte per dimension) */
    uint64_t rerank_screened8;      /* queries (of rerank_screened) whose candidates were evaluated on the int8 rows first */
    uint64_t screen8_retried_chunks; /* sub-batches whose int8 stage left more survivors than the selection holds: done again
                                       with the binary16 rows first (eight of them switch the int8 stage of the index off)  */
    uint64_t descent_multi;         /* queries (of descent_block) whose trees were dealt over several blocks, one wave of
                                       eight octets each: one query on more than one compute unit (submissions of <= 32 queries) */
} ah_search_stats;
AH_API int ah_index_search_stats(ah_index *index, ah_search_stats *out, int reset);

/* Incremental insert routing, `insert_items_in_descendants_from_frozen_reader` (src/writer.rs:1398-1459), for
 * every tree of the index at once: each of the `n` items (they must already be rows of the index's dataset — the
 * reference also re-creates `ImmutableLeafs` over all current items for every build, src/writer.rs:530) walks from
 * each root to the Descendants node it lands in, by `D::side` at split planes and by the coin of
 * arroy_hip_policy.h (`ah_route_side_is_left`, keyed by tree_seeds[t]) at `normal: None` nodes.
 * out_leaf[t * n + i] = forest-local node index reached by item i in tree t.  The host merges the items into those
 * Descendants nodes and re-splits the ones that grew beyond split_after (src/writer.rs:1411-1414, 546-561). */
AH_API int ah_route_items(ah_index *index, const uint32_t *item_ids, size_t n, const uint64_t *tree_seeds,
                          uint32_t *out_leaf);

/* ------------------------------------------------------------------------------------------
 * Measurement helpers (bench.py).  They time with hipEvents recorded on the same stream the
 * kernels run on and never touch the host data path.
 * ---------------------------------------------------------------------------------------- */

/* Launch the Q=1 distance scan `iterations` times back to back over the first `n` rows and report
 * the total kernel time in milliseconds (HIP events on the launch stream).  The query is
 * item `query_item`.  Distances of the last iteration are left in an internal device buffer; if
 * `out` is non-NULL they are copied to it (n floats) after the timed region. */
AH_API int ah_bench_scan(ah_dataset *ds, uint32_t query_item, uint64_t n, uint32_t iterations, float *out,
                  double *out_ms_total);
/* Device-to-device copy of `bytes` bytes, `iterations` times: the measured streaming ceiling. */
AH_API int ah_bench_memcpy(int device, uint64_t bytes, uint32_t iterations, double *out_ms_total);
/* Read-only stream over `bytes` bytes (16-byte loads, integer checksum, nothing written), `iterations` times: the
 * measured read ceiling of the device, the practical bound of the scan kernels next to the 8 TB/s spec peak. */
AH_API int ah_bench_read(int device, uint64_t bytes, uint32_t iterations, double *out_ms_total);
/* Name of the device (hipDeviceProp_t.name / gcnArchName) into buf. */
AH_API int ah_device_name(int device, char *buf, size_t buf_len);

/* The blobs (normals, item ids) of destroyed forests are kept committed in a process-wide pool and handed to the next
 * build, which then takes no page faults for its 9.4 GB of output (AH_HOST_CACHE_MB bounds the pool, default 16 GiB;
 * 0 = keep nothing).  This returns the pool's memory to the system; *out_bytes (may be NULL) = committed bytes released. */
AH_API int ah_host_cache_trim(uint64_t *out_bytes);
/* Device memory is cached the same way: HBM the library has obtained (datasets, their binary16 / int8 copies, the ~28 GB of
 * scratch of a 10M x 100-tree build) goes back to the driver only under memory pressure or here — a hipMalloc that lands on
 * memory the driver is still scrubbing after a hipFree was measured to stall a build's first launch by a second
 * (AH_DEVICE_CACHE_MB bounds the idle bytes, default 96 GiB; 0 = plain hipMalloc / hipFree).  device < 0: every device. */
AH_API int ah_device_cache_trim(int device, uint64_t *out_bytes);
/* ABI v6: what the library holds on `device` (< 0: every device) right now — bytes handed out to its datasets, indexes,
 * builds and per-thread scratch (`live`), and bytes parked in the cache (`idle`).  Either pointer may be NULL.  The caches
 * live as long as a dataset does: when the last dataset of a device is destroyed its idle blocks go back to the driver, and
 * with the last dataset of the process the host blob pool goes back to the system (AH_CACHE_KEEP_IDLE=1 keeps them). */
AH_API int ah_device_cache_stats(int device, uint64_t *out_live_bytes, uint64_t *out_idle_bytes);
/* Benchmark harness only: n x dims synthetic rows of the generator of arroy_hip_policy.h (the rows
 * ah_dataset_fill_synthetic makes in HBM) in HOST memory, items first_item .. first_item + n - 1, on all host cores. */
AH_API int ah_synth_rows_host(uint64_t seed, int distribution, uint64_t first_item, uint64_t n, uint32_t dims, float *out);

/* Tunables: measurement / test aids that steer the SCHEDULE of the kernels (which family a level takes, grids, cache
 * policy), never a result.  `name` is the environment variable that initialises the tunable when the library is loaded
 * ("AH_ROWS_XCD", "AH_DENSE", "AH_SCREEN8", ...: DESIGN.md lists them).  Set them while no call is running. */
AH_API int ah_tuning_set(const char *name, int64_t value);
AH_API int ah_tuning_get(const char *name, int64_t *out_value, int64_t *out_default);
AH_API int ah_tuning_reset(void);   /* every tunable back to its built-in default (not to the environment's value) */

/* Test aid: run the block -> work-item maps of the build's launches on the device — the same device functions the
 * margin kernels call and the same host-side launch plans the build uses, under the current tunables — over a shape,
 * and count how often every work item is served (every count must be 1).
 *   kind 0: screened row-major pass, a = trees per group (2 / 4 / 8 / 16), b = groups; out_counts[b][n_rows]
 *   kind 1: dense MFMA screen, a = columns (normals of the level);               out_counts[row tiles][column tiles] (ah_debug_dense_tiles)
 *   kind 2: exact-pairs pass after the dense screen, a = trees;                   out_counts[a][ceil(n_rows / 1024)]
 * `dims` sizes the rows as the build would.  out_len = number of counters the caller allocated (checked). */
AH_API int ah_debug_launch_coverage(int device, int kind, uint64_t n_rows, uint32_t dims, uint32_t a, uint32_t b,
                                    uint32_t *out_counts, uint64_t out_len);
/* Test aid (ABI v6): the tile shape the dense MFMA screen takes for a level of `n_cols` normals under the current tunables
 * (the narrow kernel: 160 rows x 64 / 128 columns; the wide one: 256 x 128 / 256) — kind 1 of ah_debug_launch_coverage
 * counts [ceil(n_rows / tile_rows)][ceil(n_cols / tile_cols)] tiles. */
AH_API int ah_debug_dense_tiles(uint64_t n_rows, uint32_t n_cols, uint32_t *out_tile_rows, uint32_t *out_tile_cols);

#ifdef __cplusplus
}
#endif

#endif /* ARROY_HIP_H */
