// common.h — internal declarations shared by the translation units of libarroy_hip.so.
// gfx950 (MI355X) only: wave = 64 lanes, hard-coded.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/arroy_hip.h"
#include "../../include/arroy_hip_policy.h"

namespace ah {

// ---------------------------------------------------------------------------------------------
// errors: thread-local text, integer codes across the ABI
// ---------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
const char *last_error();
// structured part of the last failure (ah_last_error_detail): set_error() clears it, AH_REQUIRE / AH_HIP record the
// status, the sites that know more add the item id or the expected / received sizes
void set_error_status(int status);
void set_error_detail(uint32_t item, uint64_t expected, uint64_t received);

// Every `extern "C"` entry point runs its body through guarded(): no C++ exception crosses the ABI (the reference catches
// worker panics at src/writer.rs:799-827 and turns them into Error::Panic, src/error.rs:84-85).  std::bad_alloc becomes
// AH_ERR_OUT_OF_MEMORY, anything else AH_ERR_DEVICE, both with ah_last_error() text.  While a thread is inside guarded()
// the library's own `operator new` (api.hip) honours AH_FAIL_ALLOC_AFTER — only there: helper threads have nobody to
// catch for them.
// Release paths (destructors, ah_*_destroy, the allocators' free side) must never be the allocation AH_FAIL_ALLOC_AFTER fails:
// a vector that grows by one element while a block is parked would throw out of a destructor.  NoFailScope suspends the
// fault injection of the calling thread for its lifetime.
struct NoFailScope {
    int saved;
    NoFailScope();
    ~NoFailScope();
};
int guard_enter();            // returns the previous depth
void guard_leave();
int guard_failed(const char *what, int kind, const char *text) noexcept;  // kind 0: bad_alloc, 1: std::exception, 2: unknown
template <class F>
inline int guarded(const char *what, F &&f) noexcept {
    struct Scope {
        Scope() { guard_enter(); }
        ~Scope() { guard_leave(); }
    } scope;
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return guard_failed(what, 0, nullptr);
    } catch (const std::exception &e) {
        return guard_failed(what, 1, e.what());
    } catch (...) {
        return guard_failed(what, 2, nullptr);
    }
}
#define AH_GUARDED(name) return ::ah::guarded(name, [&]() -> int {
#define AH_GUARDED_END \
    });

#define AH_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::ah::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            ::ah::set_error_status((_e == hipErrorOutOfMemory) ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE); \
            return (_e == hipErrorOutOfMemory) ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE;             \
        }                                                                                          \
    } while (0)

#define AH_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != AH_OK) return _s;   \
    } while (0)

#define AH_REQUIRE(cond, code, ...)       \
    do {                                  \
        if (!(cond)) {                    \
            ::ah::set_error(__VA_ARGS__); \
            ::ah::set_error_status(code); \
            return (code);                \
        }                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Tunables: measurement and test aids.  Every one is initialised from the environment variable of the same name when
// the library is loaded and can be changed at run time through ah_tuning_set (include/arroy_hip.h) — that is how the
// GPU tests drive every kernel-selecting switch in-process and compare the forests.  RESULTS ARE BIT-IDENTICAL UNDER
// EVERY SETTING; only the schedule (which kernel family, which grid, which cache policy) changes.
//   X(identifier, "NAME", default)
// ---------------------------------------------------------------------------------------------
#define AH_TUNABLES(X)                                                                                                  \
    X(DEBUG, "AH_DEBUG", 0)                     /* 1: synchronise after every launch of the build and name the kernel */  \
    X(TIMING, "AH_TIMING", 0)                   /* 1: per-batch / per-upload timing on stderr; 2: also one line per level */ \
    X(ROWMAJOR, "AH_ROWMAJOR", -1)              /* 0: never the row-major margin pass, 1: whenever legal, -1: cost model */ \
    X(ROWMAJOR_CACHE_MB, "AH_ROWMAJOR_CACHE_MB", -1) /* > 0: budget for one group's normals of a level */                \
    X(ROWMAJOR_MAX_TC, "AH_ROWMAJOR_MAX_TC", 16) /* largest tree group of a row-major pass */                            \
    X(ROWMAJOR_ADVANCE, "AH_ROWMAJOR_ADVANCE", 1) /* 0: node_of re-scattered every level instead of advanced in row order */ \
    X(ROWMAJOR_LDS, "AH_ROWMAJOR_LDS", 1)       /* 0: no LDS-resident variant of the row-major pass */                    \
    X(FOREST_TILE_BLOCKS, "AH_FOREST_TILE_BLOCKS", 1 << 20) /* grid caps (grid-stride beyond) */                         \
    X(FOREST_NODE_BLOCKS, "AH_FOREST_NODE_BLOCKS", 0)       /* node-major margin kernels; 0 = automatic */               \
    X(FOREST_SPLIT_BLOCKS, "AH_FOREST_SPLIT_BLOCKS", 65536) /* create_split: one wave per node */                        \
    X(FOREST_ROW_BLOCKS, "AH_FOREST_ROW_BLOCKS", 1 << 20)   /* f32 row-major passes */                                   \
    X(ROWS_XCD, "AH_ROWS_XCD", 1)               /* 0: row-major groups spread over all XCDs (never one XCD per group) */  \
    X(ROWS_XCD_MIN_GROUPS, "AH_ROWS_XCD_MIN_GROUPS", 16) /* fewest groups of <= 4 trees that get one XCD each */         \
    X(ROWS_NT, "AH_ROWS_NT", -1)                /* 0 / 1: never / always stream the rows non-temporally; -1: by size */    \
    X(ROWS_NT_BYTES, "AH_ROWS_NT_BYTES", 5 << 20) /* bytes of one group's normals beyond which the rows go non-temporal */ \
    X(ROWS_PER_BLOCK, "AH_ROWS_PER_BLOCK", 0)   /* rows per block of the screened row-major pass (0 = 32) */              \
    X(ROWS_CHUNK_MB, "AH_ROWS_CHUNK_MB", 48)    /* binary16 rows per chunk of the chunk-major schedule, in MiB */         \
    X(ROWS_CHUNK_ROWS, "AH_ROWS_CHUNK_ROWS", 0) /* > 0: rows per chunk, overrides ROWS_CHUNK_MB (small test shapes) */    \
    X(LAUNCH_MAX_ITEMS, "AH_LAUNCH_MAX_ITEMS", 0xFFFFFFFFll) /* work-items one row-major launch may carry */             \
    X(SCREEN, "AH_SCREEN", 1)                   /* 0: reference f32 arithmetic only (as AH_MARGIN_EXACT_ONLY) */          \
    X(SCREEN_VERIFY, "AH_SCREEN_VERIFY", 0)     /* 1: screened kernels evaluate f32 for EVERY pair, count violations */  \
    X(NODE_PREFETCH, "AH_NODE_PREFETCH", 1)     /* 0: no software pipeline in the int8 stage of the node-major screen */    \
    X(SCREEN8_LO, "AH_SCREEN8_LO", 1)           /* 0: no second int8 digit of the rows (stage 1 = the binary16 row) */      \
    X(SCREEN8, "AH_SCREEN8", -1)                /* 0: no int8 first stage; 1: keep it whatever the data; -1: by quality */ \
    X(MASK_BITS, "AH_MASK_BITS", -1)            /* 0 / 1: never / always pack a row-major level's sides into bits before the tiles gather them; -1: by size */ \
    X(DENSE, "AH_DENSE", -1)                    /* 0: never the dense MFMA screen; 1: whenever legal; -1: cost model */    \
    X(DENSE_MAX_COLS, "AH_DENSE_MAX_COLS", 16384)                                                                        \
    X(DENSE_NARROW, "AH_DENSE_NARROW", -1)      /* 0: never the narrow dense kernel (rows straight to registers) */         \
    X(DENSE_NARROW_MAX_COLS, "AH_DENSE_NARROW_MAX_COLS", 64) /* most columns of a level the narrow dense kernel takes */   \
    X(DENSE_NARROW_STREAM, "AH_DENSE_NARROW_STREAM", 0) /* 1: non-temporal row loads there (measured slower: 3.4 -> 4.1 ms per level) */ \
    X(DENSE_GMACS, "AH_DENSE_GMACS", 495000)    /* sustained multiply-add rate the cost model assumes, 1e9 MAC/s */       \
    X(MARGIN_MODE, "AH_MARGIN_MODE", 0)         /* ah_margin_mode for callers that pass AH_MARGIN_AUTO */                 \
    X(READBACK_DIRECT, "AH_READBACK_DIRECT", 0) /* 1: let the runtime stage the device -> pageable copies */             \
    X(READBACK_PRIORITY, "AH_READBACK_PRIORITY", 1) /* 0: the read-back worker's copy stream has the default priority (it may then share a hardware queue with the build's compute stream once more than four streams are alive) */ \
    X(READBACK_MB, "AH_READBACK_MB", 256)       /* pinned double buffer of a build's read-back worker, MiB (both halves) */ \
    X(NUMA, "AH_NUMA", 1)                       /* 0: no NUMA placement: a build's host blobs are first-touched and its read-back / page-commit threads run wherever the scheduler puts them (until round 6) */ \
    X(RETRY_GATE, "AH_RETRY_GATE", 1)           /* 0: the retry attempts of a level run over every node and tile even when the attempt before left none pending (rounds 1-6a) */ \
    X(TAIL_GROUPS, "AH_BUILD_TAIL_GROUPS", 5)   /* the last big level of a build and what follows it run tree group by tree group, each group's item ids and normals travelling under the next group's kernels (0 / 1: all trees level by level to the end) */ \
    X(TAIL_MIN_MB, "AH_BUILD_TAIL_MIN_MB", 64)  /* ... when the ids still under splitting nodes are at least this many MiB */ \
    X(TAIL_NODE_ITEMS, "AH_BUILD_TAIL_NODE_ITEMS", 2) /* ... from the first node-major level on whose nodes hold at most this many x split_after items on average (2: most children are Descendants nodes) */ \
    X(TAIL_RATIO, "AH_BUILD_TAIL_RATIO", 76)    /* ... each group this many percent of the trees of the group before it (100: equal groups): a group's ids and normals take ~0.76 of the time its kernels do, and what is left after the last launch is the LAST group's */ \
    X(SCAN_BLOCKS, "AH_SCAN_BLOCKS", 0)         /* grid cap of the distance scan (0 = built-in) */                        \
    X(MANHATTAN_ROWS, "AH_MANHATTAN_ROWS", 1)                                                                            \
    X(RERANK_INVERT, "AH_RERANK_INVERT", -1)    /* 0 / 1: never / always the row-major re-rank of big submissions */      \
    X(RERANK_SMALL, "AH_RERANK_SMALL", 1)       /* 0: ah_rerank_by_vector / _by_item never take the one-launch selection of short lists (k_topk_small) */ \
    X(PAIR_GROUP, "AH_PAIR_GROUP", 0)                                                                                    \
    X(PAIR_RUNS, "AH_PAIR_RUNS", 1)                                                                                      \
    X(SEARCH_BITMAP, "AH_SEARCH_BITMAP", 1)     /* 0: sort + dedup of the candidates always by the bitonic network */     \
    X(SEARCH_TILES, "AH_SEARCH_TILES", 1)       /* 0: never the leaf-tile re-rank of ah_search_batch */                   \
    X(SEARCH_WAVE, "AH_SEARCH_WAVE", 1)         /* 0: the descent always one octet per query (k_descend) */               \
    X(SEARCH_BLOCK_MAX_QUERIES, "AH_SEARCH_BLOCK_MAX_QUERIES", 64) /* submissions of at most this many queries descend with one BLOCK (32 octets) per query; 0: never */ \
    X(SEARCH_SMALL_GATE, "AH_SEARCH_SMALL_GATE", 1) /* 0: small submissions start on the block descent / k_units_small whatever the estimate of the leaves they open says */ \
    X(SEARCH_SMALL_UNITS_MAX_QUERIES, "AH_SEARCH_SMALL_UNITS_MAX_QUERIES", 64) /* up to this many queries a call: one block places the leaf visits (k_units_small) */ \
    X(SEARCH_FUSED_FLAG, "AH_SEARCH_FUSED_FLAG", 1) /* 0: a small submission flags its duplicate candidates with k_flag_duplicates, not inside the selection */ \
    X(SEARCH_FUSED_PREPARE, "AH_SEARCH_FUSED_PREPARE", 1) /* 0: a small submission prepares its query leaves with k_prepare_queries, not inside the block descent */ \
    X(SEARCH_SINGLE_FUSED, "AH_SEARCH_SINGLE_FUSED", 1) /* 0: a one-query submission places its leaf visits with k_units_small like the other small ones */ \
    X(SEARCH_SMALL_TILES_MAX_QUERIES, "AH_SEARCH_SMALL_TILES_MAX_QUERIES", 8) /* up to this many queries a call: leaf tiles in slabs of 64 rows, whole rows in flight */ \
    X(SEARCH_MULTI, "AH_SEARCH_MULTI", 1)       /* 0: a small submission never deals a query's trees over several blocks (k_descend_multi) */ \
    X(SEARCH_MULTI_TREES_PER_BLOCK, "AH_SEARCH_MULTI_TREES_PER_BLOCK", 8) /* ... trees per block (1 - 8: one per octet of its descent wave) */ \
    X(SEARCH_MULTI_IDS_BY_TILES, "AH_SEARCH_MULTI_IDS_BY_TILES", 1) /* 0: the last block of k_descend_multi copies a single query's ids itself */ \
    X(SEARCH_SPIN_WAIT, "AH_SEARCH_SPIN_WAIT", 1) /* 0: a small submission waits with hipStreamSynchronize instead of polling the status word its last kernel writes into pinned memory */ \
    X(SEARCH_FLAT_TILES, "AH_SEARCH_FLAT_TILES", 1) /* 0: a single query's tile launch keeps the 2-D grid (units x slabs) of the small submissions */ \
    X(SEARCH_STATUS_WIPE, "AH_SEARCH_STATUS_WIPE", 1) /* 0: every search submission clears its status block with a memset of its own */ \
    X(SEARCH_ITEM_LIST, "AH_SEARCH_ITEM_LIST", 1) /* 0: the tile launch of a small submission keeps its 2-D grid (units x slabs) instead of the (unit, slab) list k_units_small leaves */ \
    X(SEARCH_MULTI_OWN_UNITS, "AH_SEARCH_MULTI_OWN_UNITS", 1) /* 0: a call of 2 - 8 queries sorts the leaf visits of all its queries by leaf (k_units_small) instead of every query's descent writing its own units */ \
    X(SEARCH_MULTI_TRACE, "AH_SEARCH_MULTI_TRACE", 0) /* 1: ah_search_batch prints where the blocks of query 0 spent their time (stderr) */ \
    X(SEARCH_MULTI_MAX_QUERIES, "AH_SEARCH_MULTI_MAX_QUERIES", 32) /* ... up to this many queries a call (at most 32: the control block's size; 64 queries measured 2 % slower) */ \
    X(EXACT_WIDE, "AH_EXACT_WIDE", 1)           /* 0: k_forest_exact_pairs streams the row eight lines at a time (rounds 2-5) instead of asking for row and normal whole */ \
    X(RERANK_GROUPS, "AH_RERANK_GROUPS", 2)     /* groups a screened ah_rerank_batch submission is cut into (upload of group g + 1 under the kernel of g) */ \
    X(RERANK_SCREEN8, "AH_RERANK_SCREEN8", 1)   /* 0: the screen of ah_rerank_batch starts on the binary16 rows, never on the int8 copy */ \
    X(SEARCH_SCREEN8_MAX_VISITS, "AH_SEARCH_SCREEN8_MAX_VISITS", 4) /* leaves reached by at most this many queries of a call are screened on the int8 rows, the others on the binary16 rows */ \
    X(SEARCH_SCREEN8_MIN_QUERIES, "AH_SEARCH_SCREEN8_MIN_QUERIES", 65) /* ... from this many queries a call (never the in-flight tiles of <= AH_SEARCH_SMALL_TILES_MAX_QUERIES queries; 9 .. 64 measured: no gain) */ \
    X(SEARCH_SCREEN8, "AH_SEARCH_SCREEN8", 1)   /* 0: the tile re-rank of ah_search_batch likewise */ \
    X(RERANK_SCREEN, "AH_RERANK_SCREEN", 1)     /* 0: ah_rerank_batch never screens its candidates (f32 rows for all) */ \
    X(REPLICATE_HOST_BOUNCE, "AH_REPLICATE_HOST_BOUNCE", 0) /* 1: ah_dataset_replicate copies through pinned host memory even where peer access works (test aid) */ \
    X(RERANK_TIMING, "AH_RERANK_TIMING", 0)     /* 1: ah_rerank_batch accounts its wall time by phase (ah_dataset_rerank_stats) */ \
    X(SEARCH_SCREEN, "AH_SEARCH_SCREEN", 1)     /* 0: the re-rank of ah_search_batch never screens its candidates (f32 rows for all) */ \
    X(HOST_THREADS, "AH_HOST_THREADS", 8)       /* host threads one build may use at a time for its output path */        \
    X(DEVICE_CACHE_MB, "AH_DEVICE_CACHE_MB", 98304) /* idle HBM the caching allocator keeps while a dataset lives on the device */ \
    X(HOST_CACHE_MB, "AH_HOST_CACHE_MB", 16384) /* committed host memory of destroyed forests kept for the next build */   \
    X(CACHE_KEEP_IDLE, "AH_CACHE_KEEP_IDLE", 0) /* 1: keep both caches even when the last dataset (of a device / of the process) is destroyed */ \
    X(FAIL_ALLOC_AFTER, "AH_FAIL_ALLOC_AFTER", 0) /* test aid: the n-th host / device allocation from now on fails (0 = off) */ \
    X(STAGE_THREADS, "AH_STAGE_THREADS", 0)                                                                              \
    X(STAGE_MEMCPY, "AH_STAGE_MEMCPY", 0)                                                                                \
    X(STAGE_REGISTER, "AH_STAGE_REGISTER", 0)
enum Tunable {
#define AH_X(id, name, def) TUN_##id,
    AH_TUNABLES(AH_X)
#undef AH_X
        TUN_COUNT
};
long long tun(int id);  // current value (api.hip)

// fn(0) .. fn(n - 1), fn(0) on the calling thread and the others on threads of their own.  A thread that cannot be created
// (std::system_error) must not cross the C ABI: its share simply runs on the caller.
template <class F>
inline void parallel_run(unsigned n, F fn) {
    std::vector<std::thread> pool;
    unsigned started = 1;
    try {
        pool.reserve(n);
        for (; started < n; started++) pool.emplace_back(fn, started);
    } catch (...) {
    }
    if (n) fn(0u);
    for (unsigned t = started; t < n; t++) fn(t);
    for (auto &th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------
// Device memory: a caching allocator (api.hip).  HBM that the library has obtained is handed back to the driver only by
// ah_device_cache_trim or under memory pressure: a build of the 10M x 768 x 100-tree forest takes and returns ~28 GB of
// scratch, and a hipMalloc that lands on memory the driver is still scrubbing after a hipFree was measured to take a
// SECOND (r04: the first launch of every second build waited 0.9 - 1.1 s for its buffers).  dev_free keeps hipFree's
// implicit device synchronisation, so no caller can free a block a queued kernel still uses.
//   AH_DEVICE_CACHE_MB: most bytes kept idle per process (default 98304; 0 = plain hipMalloc / hipFree).
// The cache lives as long as a dataset does: when the LAST dataset of a device is destroyed its idle blocks go back to the
// driver (and the host blob pool with the last dataset of the process), so an embedding application that is done with the
// library holds none of its memory (AH_CACHE_KEEP_IDLE=1 keeps them).
// ---------------------------------------------------------------------------------------------
// optional: an allocation the caller can do without (the screens' copies) — no trim-and-retry when the device is full
hipError_t dev_malloc(void **p, size_t bytes, bool optional = false);  // on the calling thread's current device
template <typename T>
inline hipError_t dev_malloc(T **p, size_t bytes, bool optional = false) {
    return dev_malloc(reinterpret_cast<void **>(p), bytes, optional);
}
// datasets alive per device (ah_dataset_create / _replicate / _destroy): the caches' lifetime
void dataset_born(int device);
void dataset_gone(int device);
size_t host_cache_trim();                    // forest.hip: the pool of destroyed forests' blobs; returns the bytes released
// NUMA placement of a build's output path (api.hip): the host node the device hangs off (-1: unknown, one node, AH_NUMA=0), the
// calling thread onto that node's CPUs (those of them the process may use; false: left alone), a mapping's pages preferred there
// A stream for copies that must run UNDER the kernels of another stream (read-back, pipelined uploads, the build's side stream):
// created with the highest priority, because the runtime multiplexes the streams of one priority onto four hardware queues and a
// copy stream that lands on the compute stream's queue waits for every kernel in front of it (AH_READBACK_PRIORITY=0: default).
hipError_t create_copy_stream(hipStream_t *out);
int numa_node_of_device(int device);
bool numa_bind_thread_to_node(int node);
void numa_prefer_node(void *p, size_t bytes, int node);
void pinned_spare_fill(int device, size_t bytes);  // api.hip: a pinned block obtained ahead of the context that will want it
size_t pinned_spare_trim();                  // ... given back if nobody took it; returns its bytes
size_t dev_cache_live_bytes(int device);     // bytes handed out and not yet freed (ah_device_cache_stats)
// test aid (AH_FAIL_ALLOC_AFTER): true when THIS allocation is the one that must fail
bool fail_alloc_tick();
hipError_t dev_free(void *p);
hipError_t dev_free_unused(void *p);       // a block no kernel or copy ever touched: straight to the idle list, no device wait
size_t dev_cache_trim(int device);         // device < 0: every device; returns the bytes given back
size_t dev_cache_idle_bytes(int device);   // idle bytes cached for `device` (ah_build_forest adds them to hipMemGetInfo's free)

inline bool metric_is_bq(int m) { return m >= AH_BQ_EUCLIDEAN && m <= AH_BQ_COSINE; }
inline bool metric_valid(int m) { return m >= AH_EUCLIDEAN && m <= AH_BQ_COSINE; }
inline uint32_t header_floats(int m) { return m == AH_DOT_PRODUCT ? 2u : 1u; }
inline uint32_t bq_words(uint32_t dims) { return (dims + 63u) / 64u; }

// ---------------------------------------------------------------------------------------------
// Device-side view of a dataset (passed by value to kernels).
//
// HBM layout ("SoA of the LMDB record"): the stored record [tag][header][vector] is split into
//   ids[n]            u32, ascending
//   headers[n*hf]     f32, hf = 1 (bias | norm) or 2 (extra_dim, norm)
//   rows              f32 metrics: n rows of `pitch` floats, pitch = round_up(dims, 32) so every row
//                     starts on a 128-byte line and is read as whole lines by 8-lane groups;
//                     BQ metrics: n rows of `pitch` 64-bit words (pitch = round_up(words, 2): 16-B rows)
// plus an optional id -> row table.
// ---------------------------------------------------------------------------------------------
struct DataView {
    int metric;
    uint32_t dims;
    uint32_t pitch;        // floats (f32 metrics) or u64 words (BQ)
    uint32_t words;        // BQ: ceil(dims/64); else 0
    uint64_t n;
    const float *rows_f32;
    const uint64_t *rows_bq;
    float *headers;
    const uint32_t *ids;
    const uint32_t *lut;   // dense id -> row (0xFFFFFFFF = absent) or nullptr
    uint32_t lut_len;
    int identity_ids;      // ids are exactly 0..n-1
};

// binary16 shadow of an f32 dataset for the certified screen of the forest build (screen_device.h):
//   rows   n x hpitch halves (hpitch = round_up(dims, 64): an octet reads 128-byte lines, 16 B = 8 halves per lane)
//   stats  per row {|x~|, |x - x~|, |x|, 0}: 2-norms of the rounded row, of the rounding error and of the row, each
//          rounded UP (so that the bound built from them is rigorous)
struct ScreenView {
    const uint16_t *rows;
    const float4 *stats;
    uint32_t hpitch;
    float gamma_s, gamma_r;  // accumulation-error factors of the screen / of the reference f32 reduction
    float4 max_stats;        // component-wise maximum of `stats` over all rows: a bound that needs no per-row load
    // int8 copy of the rows for the first stage of the node-major screen (nullptr = stage off): rows8[n][pitch8] = q with
    // x / d ~ s_r q (d: one power of two per dimension, s_r = scale8_rows[r]: one scale per row; inf = "never decide this
    // row here"), max8 = {max |q|, max |x/d/s_r - q|, max |x|/s_r, 0} over the rows, in units of the row's scale
    const int8_t *rows8;
    uint32_t pitch8;  // bytes per row, a multiple of 128
    const float *scale8_rows;
    float4 max8;
    // the rows' second int8 digit (x / d ~ s_r (q + q2 / 256)): read instead of the binary16 row by the pairs the first
    // digit cannot decide; max8b = {max |q + q2/256|, max |x/d/s_r - q - q2/256|, max |x|/s_r, 0}.  nullptr = not built
    const int8_t *rows8_lo;
    float4 max8b;
};

// One per concurrently calling host thread: a stream plus growable device / pinned scratch.
struct Context {
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // uploads that run under the kernels of `stream` (created on first use)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // AH_RERANK_TIMING: first enqueue .. stream idle of one submission (created on first use)
    hipEvent_t ev_ring[4] = {nullptr, nullptr, nullptr, nullptr};  // staging ring (created on first use)
    void *d_scratch = nullptr;
    size_t d_cap = 0;
    void *h_pinned = nullptr;
    size_t h_cap = 0;
    void *d_filter = nullptr;  // candidate filter of a search submission (bitmap + id list): outlives its sub-batches
    size_t d_filter_cap = 0;
    void *d_multi = nullptr;   // control block of k_descend_multi (search.hip): zero between calls, allocated on first use
    // the status block of the last small search submission, if its selection kernel left it zeroed behind itself and nothing has
    // carved the scratch since (ensure_device forgets it): the next submission that lays its status block at the same address
    // skips the memset in front of its first kernel
    void *clean_status = nullptr;
    int ensure_multi(size_t bytes);
    int ensure_device(size_t bytes);
    int ensure_pinned(size_t bytes);
    int ensure_filter(size_t bytes);
    void destroy();
};

}  // namespace ah

struct ah_dataset {
    int metric = 0;
    uint32_t dims = 0;
    uint32_t pitch = 0;
    uint32_t words = 0;
    uint64_t n = 0, capacity = 0;
    int device = 0;
    bool finalized = false;
    bool dot_preprocessed = false;
    float *d_rows_f32 = nullptr;
    uint64_t *d_rows_bq = nullptr;
    float *d_headers = nullptr;
    uint32_t *d_ids = nullptr;
    uint32_t *d_lut = nullptr;
    uint32_t lut_len = 0;
    bool identity_ids = true;
    uint32_t last_id = 0;
    std::vector<uint32_t> h_ids;     // host mirror of ids (ascending) for id validation / lookups
    // lazily built binary16 shadow (first forest build of an f32 dataset); nullptr = not built / not available
    uint16_t *d_rows_h16 = nullptr;
    float4 *d_screen_stats = nullptr;
    float screen_max[4] = {0.f, 0.f, 0.f, 0.f};  // component-wise maximum of the per-row stats (host copy)
    int8_t *d_rows_i8 = nullptr;                 // int8 copy for the first screen stage (nullptr: not built / not useful)
    int8_t *d_rows_i8_lo = nullptr;              // its second digit (stage 1 of the node-major screen), optional
    float *d_scale8_rows = nullptr;              // its scale per row
    float *d_dim_scale = nullptr;                // 2 x pitch8 floats: the power-of-two scale of every dimension, then its inverse
    uint32_t pitch8 = 0;
    float screen8_max[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // {|q|, |y/s - q|, |x|/s, |q + q2/256|, |y/s - q - q2/256|} maxima
    double screen8_quality = 0.0;                // expected undecided share indicator (forest.hip: ensure_screen)
    uint32_t hpitch = 0;
    bool screen_never = false;                   // the screen can never apply to this dataset (1-bit metric, dims < 32)
    bool screen8_decided = false;                // the int8 copy was built or found useless: do not try again
    bool screen_alloc_failed = false;            // the last attempt failed for lack of memory (retried by the next BUILD only)
    // the copies above are published: d_rows_h16 / d_screen_stats / hpitch / screen_max are final and may be read without
    // `mu` (store-release in ensure_screen after the last of them is written, load-acquire by the search paths)
    std::atomic<bool> screen_ready{false};
    std::atomic<uint32_t> rerank8_fails{0};      // ah_rerank_batch submissions whose int8 stage left too many survivors
    std::atomic<bool> rerank8_off{false};        // ... often enough that the dataset's re-rank starts on the binary16 rows from now on
    std::atomic<bool> screen8_ready{false};      // ... and d_rows_i8 / d_scale8_rows / d_dim_scale / pitch8 / screen8_max likewise
    // staging in flight (ah_dataset_upload_*): the context whose stream / pinned ring the uploads use until
    // ah_dataset_finalize (or ah_dataset_upload_flush) waits for them
    ah::Context *up_ctx = nullptr;
    int up_buf = 0;
    bool up_used[4] = {false, false, false, false};
    std::mutex mu;
    std::vector<ah::Context *> pool;
    bool counted = false;                        // dataset_born() ran for this handle (its destroy then runs dataset_gone())
    std::thread reserve_thread;                  // ah_dataset_reserve_build: fills the device cache while records are staged
    bool replicated_through_host = false;        // ah_dataset_replicate made this replica without peer access (diagnostic)
    ah_rerank_stats rr_stats{};                  // AH_RERANK_TIMING=1: where ah_rerank_batch's wall time went (under `mu`)
    double reserve_seconds = 0.0;                // ... how long that helper ran (written by it, read after the join)
    double reserve_wait_seconds = 0.0;           // ... and how long the first build then waited for it (ah_build_stats, ABI v7)

    // Wait for ah_dataset_reserve_build's helper, whoever gets there first (concurrent builds on one dataset are allowed: the
    // handle is moved out under `mu`, so exactly one caller joins it; a failing join must not cross the C ABI)
    void join_reserve() {
        std::thread t;
        {
            std::lock_guard<std::mutex> lk(mu);
            t = std::move(reserve_thread);
        }
        if (t.joinable()) {
            try {
                t.join();
            } catch (...) {
            }
        }
    }
    ah::DataView view() const;
    size_t row_bytes() const { return ah::metric_is_bq(metric) ? (size_t)pitch * 8 : (size_t)pitch * 4; }
    ah::Context *acquire();
    void release(ah::Context *c);
};

namespace ah {

// owning device pointer for short-lived buffers on paths with early error returns
struct DevMem {
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
    ~DevMem() {
        if (p) (void)dev_free(p);
    }
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

struct ContextLease {
    ah_dataset *ds;
    Context *c;
    explicit ContextLease(ah_dataset *d) : ds(d), c(d->acquire()) {}
    ~ContextLease() {
        if (c) ds->release(c);
    }
};

// ---- launchers implemented in the kernel translation units ------------------------------------
// distance.hip
int launch_prepare_query(const DataView &dv, const float *d_query_f32, void *d_qvec, float *d_qhdr, hipStream_t s);
int launch_load_item_as_query(const DataView &dv, uint32_t row, void *d_qvec, float *d_qhdr, hipStream_t s);
// distances of `n` rows: rows given by d_ids (item ids, may be nullptr = rows 0..n-1). d_err: u32 flags.
int launch_distances(const DataView &dv, const void *d_qvec, const float *d_qhdr, const uint32_t *d_ids, uint64_t n,
                     float *d_out, uint32_t *d_err, hipStream_t s);
size_t topk_scratch_bytes(uint64_t n, size_t k);
int launch_topk(const DataView &dv, const float *d_dist, const uint32_t *d_ids, uint64_t n, size_t k, void *d_scratch,
                uint32_t *d_out_ids, float *d_out_dist, hipStream_t s);
// one launch for one short list (n <= 16 384, k <= 1024): results and the status word straight into pinned host memory
bool topk_small_fits(uint64_t n, size_t k);
int launch_topk_small(const DataView &dv, const float *d_dist, const uint32_t *ids, uint64_t n, size_t k, uint32_t *out_ids,
                      float *out_dist, uint32_t *d_err, uint32_t *host_err, hipStream_t s);
int launch_headers_from_vectors(const DataView &dv, uint64_t first_row, uint64_t n, hipStream_t s);
int launch_quantize_rows(const float *d_src, uint32_t src_pitch, uint32_t dims, uint64_t *d_dst, uint32_t dst_pitch,
                         uint32_t words, uint64_t n, hipStream_t s);
int launch_synth_fill(float *d_rows, uint32_t pitch, uint32_t dims, uint64_t first_item, uint64_t n, uint64_t seed,
                      int distribution, hipStream_t s);
int launch_build_lut(const uint32_t *d_ids, uint64_t n, uint32_t *d_lut, uint32_t lut_len, hipStream_t s);
int launch_preprocess_dot(const DataView &dv, float *d_max_norm_bits, hipStream_t s);
int launch_decode_item(const DataView &dv, uint32_t row, float *d_out, hipStream_t s);
int launch_bench_read(const void *d_src, uint64_t bytes, unsigned long long *d_sink, hipStream_t s);

// batch.hip
size_t batch_key_stride(uint32_t max_n);
uint32_t batch_tile_candidates();
bool batch_supported(uint32_t k);
uint32_t batch_rounds(uint32_t n, uint32_t k);
int launch_rerank_batch(const DataView &dv, const float *d_q_f32, uint32_t n_queries, uint8_t *d_qvecs, uint64_t qstride,
                        float *d_qhdrs, const void *d_segs, const void *d_tiles, uint32_t n_tiles, const uint32_t *d_ids,
                        float *d_dist, uint64_t *d_keys_a, uint64_t *d_keys_b, uint64_t kstride, uint32_t max_n,
                        uint32_t k_out, uint32_t max_rounds, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_err,
                        hipStream_t s, uint64_t n_candidates, uint32_t *d_inv_counters);

int launch_prepare_queries_only(const DataView &dv, const float *d_q_f32, uint32_t n_queries, uint8_t *d_qvecs,
                                uint64_t qstride, float *d_qhdrs, hipStream_t s);
int launch_rerank_batch_prepared(const DataView &dv, uint32_t n_queries, const uint8_t *d_qvecs, uint64_t qstride,
                                 const float *d_qhdrs, const void *d_segs, const void *d_tiles, uint32_t n_tiles,
                                 const uint32_t *d_ids, float *d_dist, uint64_t *d_keys_a, uint64_t *d_keys_b,
                                 uint64_t kstride, uint32_t max_n, uint32_t k_out, uint32_t max_rounds,
                                 uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_err, hipStream_t s,
                                 uint64_t n_candidates, uint32_t *d_inv_counters);
// row-major ("inverted") re-rank of big submissions: policy + the counter scratch it needs (see batch.hip)
bool batch_invert_wanted(const DataView &dv, uint64_t n_candidates);
size_t batch_invert_counter_bytes(uint64_t n_rows, uint64_t n_candidates);

// forest.hip: the binary16 shadow of an f32 dataset (+ per-row norms), made once per dataset by whoever needs it first — the
// certified screens of the forest build (want8: also the int8 copies of its node-major stage) or of the search's re-rank.
// false: not applicable (1-bit metric, dims < 32) or no memory for it right now.
// retry_failed: ask for the memory again although an earlier attempt found none (the builds do; the readers do not — a
// read-only process short on HBM must not allocate, fail and free 2 x dims bytes per item on every call).
bool ensure_screen(ah_dataset *ds, hipStream_t s, bool want8, bool retry_failed = false);
bool ensure_screen8_search(ah_dataset *ds, hipStream_t s);

// search.hip: the certified top-k screen for the candidate lists of ah_rerank_batch (binary16 rows first, f32 for the survivors)
// (tile_first .. tile_first + n_tiles: the tiles this call screens — the caller may launch the lists group by group while the
// ids of the next group are still on their way; select = also run the per-query selection, i.e. this was the last group)
int launch_rerank_screened(ah_dataset *ds, uint32_t nq, const uint8_t *d_qvecs, uint64_t qstride, const float *d_qhdrs,
                           const void *d_segs, const void *d_tiles, uint32_t tile_first, uint32_t n_tiles, uint32_t tile_candidates,
                           const uint32_t *d_ids, float *d_dist, float *d_aux, uint16_t *d_q16, float4 *d_qstats, uint32_t k_out,
                           uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_err, hipStream_t s, bool first, bool select,
                           int8_t *d_q8 = nullptr, float4 *d_q8stats = nullptr, float *d_aux8 = nullptr);

// split.hip
int launch_split_sides(const DataView &dv, const void *d_nvec, const float *d_nhdr, const uint32_t *d_ids, uint64_t n,
                       uint8_t *d_side_bits, unsigned long long *d_n_left, float *d_margins, uint32_t *d_err,
                       hipStream_t s, int row_is_normal = 0);
int launch_create_split(const DataView &dv, const uint32_t *d_sample_rows, void *d_out_vec, float *d_out_hdr,
                        hipStream_t s);

}  // namespace ah
