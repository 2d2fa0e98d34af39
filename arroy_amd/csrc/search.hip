// search.hip — `Reader::nns_by_leaf` (src/reader.rs:317-401) entirely on device, for a batch of queries:
// best-first tree descent -> candidate collection -> sort + dedup -> re-rank -> top-k.
//
// The forest built by ah_build_forest is mirrored in HBM as flat arrays: nodes
// {kind, left/right or descendants range, normal row}, the split-plane normals as a second row matrix
// (same layout and the same exact-order margin code as the item rows) and the descendants blob.
//
// Descent: ONE OCTET PER QUERY (8 queries per wave).  Lane 0 of the octet owns the priority queue — a binary
// max-heap of 64-bit keys `orderable(distance) << 32 | node` in LDS, which is exactly the reference's
// BinaryHeap<(OrderedFloat<f32>, NodeId)> order, ties included; all 8 lanes compute the margin of a popped
// split node with the AVX-order dot product (both operands stream from L2/HBM) and copy descendants.
// The pop sequence is inherently sequential per query (best-first), so throughput comes from running
// thousands of queries concurrently; a query whose queue outgrows its LDS slot is re-run by a second kernel
// with the queue in global memory (capacity = number of nodes, a hard bound: every node is pushed at most
// once).
#include <algorithm>
#include <chrono>
#include <new>

#include "common.h"
#include "split_device.h"
#include "screen_device.h"

namespace ah {

static constexpr uint32_t kHeapLds = 1024;   // queue entries per query in LDS (8 KiB)
static constexpr uint32_t kSortLds = 16384;  // candidate ids sorted in LDS per query (64 KiB)

struct DNode {
    uint32_t kind;  // AH_NODE_*; bit 8 = has_normal
    uint32_t a;     // SPLIT: left            DESCENDANTS: first id (index into the blob)
    uint32_t b;     // SPLIT: right           DESCENDANTS: count
    uint32_t c;     // SPLIT: normal row      DESCENDANTS: unused
};

// One popped Descendants node of one query: the leaf-tile re-rank groups these by node.
struct Visit {
    uint32_t node;  // index into nodes
    uint32_t q;     // query
    uint32_t pos;   // where the leaf's ids start in the query's candidate buffer
    uint32_t n;     // how many (the leaf's size, or what the candidate filter kept of it)
};
struct VisitSink {          // all null / 0: the descent records nothing
    Visit *visits;          // appended in pop order of whichever query gets there first
    uint32_t *total;        // number of visits appended (may exceed cap: then bit 5 of *err is set)
    uint32_t cap;
    uint32_t *leaf_count;   // per node: visits of that node
    uint32_t *err;          // bit 4: a queue overflowed its LDS slot; bit 5: more than `cap` visits
};

// Words of the per-chunk device block [error bits][counters of ah_search_stats]: which kernel produced a query's candidates,
// how the leaf tiles were cut.  Proof of the path taken (ah_index_search_stats), never an input of a result.
enum SearchStatSlot {
    SS_ERR = 0, SS_WAVE_SMALL, SS_WAVE_BIG, SS_OCTET_LDS, SS_OCTET_GLOBAL, SS_UNITS_16, SS_UNITS_8, SS_UNITS_4, SS_VISITS,
    SS_SCREENED, SS_SURVIVORS, SS_BLOCK, SS_VISIT_TOTAL, SS_N_UNITS, SS_DONE, SS_MULTI, SS_WORDS = 16
};

struct SearchParams {
    uint32_t *stats;              // SS_WORDS words (may be nullptr)
    const DNode *nodes;
    const uint32_t *roots;
    uint32_t n_trees;
    const uint32_t *desc;
    const uint32_t *filter_bits;  // bitmap over item ids or nullptr
    uint64_t filter_len_bits;  // max item id + 1 (item id u32::MAX is legal, src/tests/writer.rs:161-179)
    uint32_t search_k;            // already multiplied by the oversampling, clamped to the blob size
    uint32_t nns_stride;          // capacity of one query's candidate buffer
    const uint32_t *leaf_kept;    // per node: ids of the leaf the filter keeps (k_leaf_kept), or nullptr
};

__device__ __forceinline__ float key_to_dist(uint32_t k) {
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ void heap_push(uint64_t *h, uint32_t &n, uint64_t key) {
    uint32_t i = n++;
    while (i > 0) {
        const uint32_t p = (i - 1) >> 1;
        if (h[p] >= key) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = key;
}
__device__ __forceinline__ uint64_t heap_pop(uint64_t *h, uint32_t &n) {
    const uint64_t top = h[0];
    const uint64_t last = h[--n];
    uint32_t i = 0;
    for (;;) {
        const uint32_t l = 2 * i + 1;
        if (l >= n) break;
        const uint32_t r = l + 1;
        const uint32_t m = (r < n && h[r] > h[l]) ? r : l;
        if (h[m] <= last) break;
        h[i] = h[m];
        i = m;
    }
    if (n) h[i] = last;
    return top;
}
// Rust f32::min: the non-NaN operand if one is NaN.
__device__ __forceinline__ float rust_min(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}

// `D::margin(&normal, query_leaf)` with the normal a ROW of the normals matrix `nv` and the query leaf in
// global memory (qvec / qh).  All 8 lanes of the octet participate; every lane gets the result.
template <bool Q_LDS = false>
__device__ __forceinline__ float descent_margin(const DataView &nv, uint32_t nrow, const void *qvec, LeafHdr qh,
                                                uint32_t j) {
    if (metric_is_bq_dev(nv.metric)) {
        const uint64_t *np = nv.rows_bq + (uint64_t)nrow * nv.pitch;
        const uint64_t *qp = reinterpret_cast<const uint64_t *>(qvec);
        uint32_t ham = 0;
        for (uint32_t w = 0; w < nv.pitch; w++) ham += (uint32_t)__popcll(np[w] ^ qp[w]);
        const float d = (float)bq_dot_from_hamming(ham, nv.words);
        return nv.metric == AH_BQ_COSINE ? d : f_add(nv.headers[nrow], d);
    }
    const float *np = nv.rows_f32 + (uint64_t)nrow * nv.pitch;
    const float *qp = reinterpret_cast<const float *>(qvec);
    // (the whole normal requested at once: the descent is a chain of such margins, one per pop; qvec: LDS in the wave / block
    // kernels)
    // (the normal's header requested with its vector, not after the reduction: the loads of the wide reduction are fenced)
    float hdr = 0.0f;
    if (nv.metric == AH_EUCLIDEAN || nv.metric == AH_MANHATTAN) hdr = nv.headers[nrow];
    else if (nv.metric == AH_DOT_PRODUCT) hdr = nv.headers[2 * (uint64_t)nrow];
    const float d = nv.dims >= 32 ? octet_reduce_wide<OP_DOT, Q_LDS>(np, qp, nv.dims, j) : octet_reduce_any<OP_DOT>(np, qp, nv.dims, j);
    if (nv.metric == AH_EUCLIDEAN || nv.metric == AH_MANHATTAN) return f_add(hdr, d);
    if (nv.metric == AH_DOT_PRODUCT) return f_add(d, f_mul(hdr, qh.h0));
    return d;
}

// Visit record of one popped leaf (VisitSink), `n` ids written at `pos` of query q's candidate buffer.
__device__ __forceinline__ void record_visit(const VisitSink &sink, uint32_t node, uint32_t q, uint32_t pos, uint32_t n) {
    if (!sink.visits || n == 0) return;
    const uint32_t slot = atomicAdd(sink.total, 1u);
    if (slot < sink.cap) {
        sink.visits[slot] = Visit{node, q, pos, n};
        if (sink.leaf_count) atomicAdd(&sink.leaf_count[node], 1u);
    } else {
        atomicOr(sink.err, 32u);
    }
}
// `descendants & candidates` of one leaf by one octet: the kept ids, in the leaf's order, to dst (nullptr: count only);
// returns how many.  32 ids per step (4 per lane, their bitmap words requested together); a ballot gives every lane the
// number of kept ids before its own.  Octets of a wave may be here with different leaves or not at all: a lane reads only
// its octet's byte of the ballot.
__device__ __forceinline__ uint32_t copy_filtered(const SearchParams &sp, const uint32_t *__restrict__ ids, uint32_t n,
                                                  uint32_t *__restrict__ dst, uint32_t j) {
    const uint32_t shift = threadIdx.x & 56u;  // first lane of the octet
    uint32_t written = 0;
    for (uint32_t base = 0; base < n; base += 32) {
        uint32_t id[4], keep[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = base + 8 * u + j;
            id[u] = i < n ? ids[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = base + 8 * u + j < n && id[u] < sp.filter_len_bits;
            keep[u] = in ? (sp.filter_bits[id[u] >> 5] >> (id[u] & 31)) & 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t mask = (uint32_t)(__ballot(keep[u] != 0) >> shift) & 0xFFu;
            if (keep[u] && dst) dst[written + __popc(mask & ((1u << j) - 1u))] = id[u];
            written += __popc(mask);
        }
    }
    return written;
}

// One octet per query.  HEAP_GLOBAL = false: queue in LDS (kHeapLds entries), overflow reported;
// true: queue in global memory with `heap_cap` entries per query (re-run of the overflowed queries).
template <bool HEAP_GLOBAL>
__global__ __launch_bounds__(64) void k_descend(DataView nv, SearchParams sp, const uint32_t *__restrict__ query_list,
                                                uint32_t n_list, const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                const float *__restrict__ qhdrs, uint32_t *__restrict__ nns,
                                                uint32_t *__restrict__ nns_count, uint32_t *__restrict__ overflow,
                                                uint64_t *heap_global, uint32_t heap_cap, VisitSink sink,
                                                bool only_flagged) {
    extern __shared__ uint64_t s_heap[];  // 8 x kHeapLds entries when the queue lives in LDS
    const uint32_t o = threadIdx.x >> 3, j = threadIdx.x & 7u;
    const uint32_t slot = blockIdx.x * 8 + o;
    bool live = slot < n_list;
    const uint32_t q = live ? (query_list ? query_list[slot] : slot) : 0u;
    if (only_flagged && live) live = overflow[q] != 0;  // the queries k_descend_wave left (its capacities, ties)
    uint64_t *heap = HEAP_GLOBAL ? heap_global + (uint64_t)slot * heap_cap : s_heap + o * kHeapLds;
    const uint32_t cap = HEAP_GLOBAL ? heap_cap : kHeapLds;
    const void *qvec = qvecs + (uint64_t)q * qstride;
    const LeafHdr qh = {qhdrs[2 * (uint64_t)q], qhdrs[2 * (uint64_t)q + 1]};
    uint32_t *my_nns = nns + (uint64_t)q * sp.nns_stride;
    uint32_t hn = 0, nn = 0;
    bool failed = false;
    if (live && j == 0) {  // queue.extend(repeat(+inf).zip(roots)), src/reader.rs:338
        for (uint32_t t = 0; t < sp.n_trees && !failed; t++) {
            if (hn == cap) failed = true;
            else heap_push(heap, hn, ((uint64_t)0xFF800000u << 32) | sp.roots[t]);
        }
    }
    bool active = live;
    while (__any(active)) {
        // lane 0 pops; the decision is broadcast to the octet
        uint32_t action = 0, node = 0;
        float dist = 0.0f;
        if (active && j == 0) {
            if (failed || nn >= sp.search_k || hn == 0) {
                action = 0;
            } else {
                const uint64_t key = heap_pop(heap, hn);
                node = (uint32_t)key;
                dist = key_to_dist((uint32_t)(key >> 32));
                action = sp.nodes[node].kind & 0xFFu;
            }
        }
        action = __shfl(action, 0, 8);
        node = __shfl(node, 0, 8);
        dist = __shfl(dist, 0, 8);
        if (!active || action == 0) {
            active = false;
            continue;
        }
        const DNode nd = sp.nodes[node];
        if (action == AH_NODE_DESCENDANTS) {  // src/reader.rs:354-360
            if ((uint64_t)nn + nd.b > sp.nns_stride) {  // cannot happen for a validated forest; never write out of bounds
                failed = true;
                continue;
            }
            const uint32_t *ids = sp.desc + nd.a;
            if (!sp.filter_bits) {
                for (uint32_t i = j; i < nd.b; i += 8) my_nns[nn + i] = ids[i];
                if (j == 0) record_visit(sink, node, q, nn, nd.b);
                nn += nd.b;
            } else {  // descendants & candidates: order inside nns is irrelevant (sorted afterwards)
                const uint32_t kept = copy_filtered(sp, ids, nd.b, my_nns + nn, j);
                if (j == 0) record_visit(sink, node, q, nn, kept);
                nn += kept;
            }
        } else {  // SplitPlaneNormal, src/reader.rs:361-372
            float margin = 0.0f;
            if (nd.kind & 0x100u) margin = descent_margin(nv, nd.c, qvec, qh, j);
            if (j == 0) {
                if (hn + 2 > cap) {
                    failed = true;
                } else {  // D::pq_distance, src/distance/mod.rs:63-68
                    const float pl = rust_min(-margin, dist), pr = rust_min(margin, dist);
                    heap_push(heap, hn, ((uint64_t)orderable_key(pl) << 32) | nd.a);
                    heap_push(heap, hn, ((uint64_t)orderable_key(pr) << 32) | nd.b);
                }
            }
        }
    }
    if (live && j == 0) {
        nns_count[q] = failed ? 0u : nn;
        overflow[q] = failed ? 1u : 0u;
        if (failed && sink.err) atomicOr(sink.err, 16u);
        if (!failed && sp.stats) atomicAdd(&sp.stats[HEAP_GLOBAL ? SS_OCTET_GLOBAL : SS_OCTET_LDS], 1u);
    }
}

// |descendants & candidates| of every leaf, once per filtered submission (the descent then pays one load per popped leaf
// instead of a walk over its ids): one octet per node.
__global__ __launch_bounds__(256) void k_leaf_kept(SearchParams sp, uint32_t n_nodes, uint32_t *__restrict__ kept) {
    const uint32_t j = threadIdx.x & 7u;
    const uint32_t octets = (gridDim.x * blockDim.x) >> 3;
    for (uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; node < n_nodes; node += octets) {
        const DNode nd = sp.nodes[node];
        uint32_t c = 0;
        if ((nd.kind & 0xFFu) == AH_NODE_DESCENDANTS) c = copy_filtered(sp, sp.desc + nd.a, nd.b, nullptr, j);
        if (j == 0) kept[node] = c;
    }
}

// ---- one WAVE per query ------------------------------------------------------------------------------------------
// The keys a best-first search pops never increase (a child's key is min(parent's key, +-margin)), so the candidates
// are a static set: the leaves in decreasing key order until `search_k` ids are collected.  The 8 octets of the wave
// run the search of the trees t = octet (mod 8) independently, each with its own queue in LDS, and RECORD the leaves
// they pop instead of taking them.  A recorded leaf is settled once its key is strictly above every queue's top (nothing
// unexplored can come before it); when the settled leaves hold >= search_k ids (or every queue is empty) they are
// sorted by key and the prefix the sequential loop would have taken is copied out.  Same candidates as k_descend, about
// an eighth of its chain of dependent pops.  What is left to k_descend (overflow[q] = 1): a queue or leaf list that
// outgrows its LDS slot, and equal keys of two octets across the cut (the reference orders those by node id and by which
// parent was popped first; inside one octet its own pop order settles them -- every descendant on the far side of a split
// whose margin is smaller than the key inherits it, so equal keys are common once a query goes past its own leaves).
// kWaveHeap queue entries and kWaveLeaves recorded leaves per octet: 256 / 64 (30 KiB of LDS per query), or 1024 / 128
// (92 KiB) for submissions whose candidate filter makes a query pop many more nodes.
constexpr size_t wave_lds_bytes(uint32_t heap, uint32_t leaves) { return 8 * (size_t)heap * 8 + 8 * (size_t)leaves * (8 + 8 + 4 + 4 + 4 + 4); }
template <uint32_t kWaveHeap, uint32_t kWaveLeaves>
__global__ __launch_bounds__(64) void k_descend_wave(DataView nv, SearchParams sp, uint32_t nq,
                                                     const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                     const float *__restrict__ qhdrs, uint32_t *__restrict__ nns,
                                                     uint32_t *__restrict__ nns_count, uint32_t *__restrict__ overflow,
                                                     VisitSink sink, bool only_flagged) {
    extern __shared__ uint64_t s_wave_lds[];  // wave_lds_bytes(kWaveHeap, kWaveLeaves)
    if (blockIdx.x >= nq || (only_flagged && overflow[blockIdx.x] == 0)) return;  // the second pass, with the big queues
    uint64_t(*s_heap)[kWaveHeap] = reinterpret_cast<uint64_t(*)[kWaveHeap]>(s_wave_lds);
    uint64_t(*s_leaf)[kWaveLeaves] = reinterpret_cast<uint64_t(*)[kWaveLeaves]>(s_wave_lds + 8 * kWaveHeap);  // key word << 32 | node
    uint64_t *s_sorted = s_wave_lds + 8 * kWaveHeap + 8 * kWaveLeaves;
    uint32_t(*s_leaf_n)[kWaveLeaves] = reinterpret_cast<uint32_t(*)[kWaveLeaves]>(s_sorted + 8 * kWaveLeaves);
    uint32_t *s_sorted_n = reinterpret_cast<uint32_t *>(s_sorted + 8 * kWaveLeaves) + 8 * kWaveLeaves;
    uint32_t *s_pos = s_sorted_n + 8 * kWaveLeaves, *s_sorted_node = s_pos + 8 * kWaveLeaves;
    const uint32_t q = blockIdx.x, o = threadIdx.x >> 3, j = threadIdx.x & 7u, lane = threadIdx.x;
    if (q >= nq) return;
    uint64_t *heap = s_heap[o];
    // the query leaf in LDS (behind the queues; qstride bytes): every margin of the descent reads it
    uint4 *s_q4 = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(s_wave_lds) + wave_lds_bytes(kWaveHeap, kWaveLeaves));
    {
        const uint4 *g_q4 = reinterpret_cast<const uint4 *>(qvecs + (uint64_t)q * qstride);
        for (uint32_t i = lane; i < (uint32_t)(qstride >> 4); i += 64) s_q4[i] = g_q4[i];
        __syncthreads();
    }
    const void *qvec = s_q4;
    const LeafHdr qh = {qhdrs[2 * (uint64_t)q], qhdrs[2 * (uint64_t)q + 1]};
    uint32_t *my_nns = nns + (uint64_t)q * sp.nns_stride;
    uint32_t hn = 0, nl = 0;  // lane 0 of the octet
    bool failed = false;
    if (j == 0)
        for (uint32_t t = o; t < sp.n_trees; t += 8) heap_push(heap, hn, ((uint64_t)0xFF800000u << 32) | sp.roots[t]);
    uint32_t threshold = 0;
    bool all_settled = false;
    for (;;) {
        uint32_t action = 0, node = 0, key_word = 0;
        if (j == 0 && hn > 0) {
            const uint64_t key = heap_pop(heap, hn);
            node = (uint32_t)key;
            key_word = (uint32_t)(key >> 32);
            action = sp.nodes[node].kind & 0xFFu;
        }
        action = __shfl(action, 0, 8);
        node = __shfl(node, 0, 8);
        key_word = __shfl(key_word, 0, 8);
        if (action != 0) {
            const DNode nd = sp.nodes[node];
            if (action == AH_NODE_DESCENDANTS) {
                // the ids this leaf adds (`descendants & candidates` under a filter); a leaf that adds none is not a visit
                const uint32_t kept = sp.leaf_kept ? sp.leaf_kept[node] : nd.b;
                if (j == 0 && kept) {
                    if (nl == kWaveLeaves) {
                        failed = true;
                    } else {
                        s_leaf[o][nl] = ((uint64_t)key_word << 32) | node;
                        s_leaf_n[o][nl] = kept;
                        nl++;
                    }
                }
            } else {
                float margin = 0.0f;
                if (nd.kind & 0x100u) margin = descent_margin<true>(nv, nd.c, qvec, qh, j);
                if (j == 0) {
                    if (hn + 2 > kWaveHeap) {
                        failed = true;
                    } else {
                        const float dist = key_to_dist(key_word);
                        const float pl = rust_min(-margin, dist), pr = rust_min(margin, dist);
                        heap_push(heap, hn, ((uint64_t)orderable_key(pl) << 32) | nd.a);
                        heap_push(heap, hn, ((uint64_t)orderable_key(pr) << 32) | nd.b);
                    }
                }
            }
        }
        if (__any(failed)) {
            if (lane == 0) {
                nns_count[q] = 0;
                overflow[q] = 1;
            }
            return;
        }
        // the largest key still queued anywhere, and the ids held by the leaves strictly above it
        const bool has = j == 0 && hn > 0;
        uint32_t top = has ? (uint32_t)(heap[0] >> 32) : 0u;
        for (uint32_t d = 32; d > 0; d >>= 1) top = max(top, (uint32_t)__shfl_xor((int)top, d, 64));
        const bool any_queued = __any(has);
        const uint32_t nl_o = __shfl(nl, 0, 8);
        uint32_t held = 0;
        for (uint32_t i = j; i < nl_o; i += 8)
            if (!any_queued || (uint32_t)(s_leaf[o][i] >> 32) > top) held += s_leaf_n[o][i];
        for (uint32_t d = 32; d > 0; d >>= 1) held += __shfl_xor(held, d, 64);
        if (!any_queued || held >= sp.search_k) {
            threshold = top;
            all_settled = !any_queued;
            break;
        }
    }
    // the settled leaves, sorted by decreasing key
    const uint32_t nl_o = __shfl(nl, 0, 8);
    uint32_t mine = 0;
    for (uint32_t i = j; i < nl_o; i += 8) mine += (all_settled || (uint32_t)(s_leaf[o][i] >> 32) > threshold) ? 1u : 0u;
    for (uint32_t d = 4; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 8);  // settled leaves of this octet
    uint32_t first = 0, n_settled = 0;
    for (uint32_t oo = 0; oo < 8; oo++) {
        const uint32_t c = __shfl(mine, oo * 8, 64);
        first += oo < o ? c : 0u;
        n_settled += c;
    }
    // Equal keys: inside one octet the order is the octet's own pop order (its queue is the sequential queue restricted
    // to its trees, ties included), so the sort key is key word << 32 | octet << 16 | 0xFFFF - (index in the octet's list).
    if (j == 0) {
        uint32_t w = first;
        for (uint32_t i = 0; i < nl; i++)
            if (all_settled || (uint32_t)(s_leaf[o][i] >> 32) > threshold) {
                s_sorted[w] = (s_leaf[o][i] & 0xFFFFFFFF00000000ull) | (o << 16) | (0xFFFFu - i);
                s_sorted_node[w] = (uint32_t)s_leaf[o][i];
                s_sorted_n[w] = s_leaf_n[o][i];
                w++;
            }
    }
    uint32_t p2 = 64;
    while (p2 < n_settled) p2 <<= 1;
    __syncthreads();
    for (uint32_t t = n_settled + lane; t < p2; t += 64) {
        s_sorted[t] = 0;
        s_sorted_n[t] = 0;
        s_sorted_node[t] = 0;
    }
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t str = size >> 1; str > 0; str >>= 1) {
            __syncthreads();
            for (uint32_t t = lane; t < (p2 >> 1); t += 64) {
                const uint32_t a_i = 2 * t - (t & (str - 1)), b_i = a_i + str;
                const bool down = (a_i & size) == 0;  // descending order
                const uint64_t x = s_sorted[a_i], y = s_sorted[b_i];
                if ((x < y) == down) {
                    s_sorted[a_i] = y;
                    s_sorted[b_i] = x;
                    const uint32_t nx = s_sorted_n[a_i], dx = s_sorted_node[a_i];
                    s_sorted_n[a_i] = s_sorted_n[b_i];
                    s_sorted_n[b_i] = nx;
                    s_sorted_node[a_i] = s_sorted_node[b_i];
                    s_sorted_node[b_i] = dx;
                }
            }
        }
    }
    __syncthreads();
    // `if nns.len() >= search_k { break }` before every pop: leaf i is taken iff the leaves before it hold < search_k ids
    const uint32_t per = p2 / 64;
    uint32_t local = 0;
    for (uint32_t i = 0; i < per; i++) local += s_sorted_n[lane * per + i];
    uint32_t incl = local;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    uint32_t before = incl - local, taken = 0, ids_taken = 0;
    bool tie = false;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t e = lane * per + i;
        if (e < n_settled && before < sp.search_k) {
            s_pos[e] = before;
            taken++;
            ids_taken = before + s_sorted_n[e];
            // The leaf that reaches search_k.  If other settled leaves have its key, how many of them the sequential loop
            // takes depends on their order: inside one octet that is the order above; between octets it is the sequential
            // queue's business (node ids, and which parent was popped first) -- e.g. two leaves of 83 and 227 ids with one
            // key and search_k = 90: the bigger node id pops first and decides whether the other is taken at all.
            if (ids_taken >= sp.search_k) {
                const uint32_t kw = (uint32_t)(s_sorted[e] >> 32), oct = (uint32_t)s_sorted[e] >> 16;
                for (uint32_t g = e + 1; g < n_settled && (uint32_t)(s_sorted[g] >> 32) == kw; g++)
                    if (((uint32_t)s_sorted[g] >> 16) != oct) tie = true;
                for (uint32_t g = e; g-- > 0 && (uint32_t)(s_sorted[g] >> 32) == kw;)
                    if (((uint32_t)s_sorted[g] >> 16) != oct) tie = true;
            }
        }
        before += s_sorted_n[e];
    }
    for (uint32_t d = 32; d > 0; d >>= 1) {
        taken += __shfl_xor(taken, d, 64);
        ids_taken = max(ids_taken, (uint32_t)__shfl_xor((int)ids_taken, d, 64));
    }
    if (__any(tie) || ids_taken > sp.nns_stride) {
        if (lane == 0) {
            nns_count[q] = 0;
            overflow[q] = 1;
        }
        return;
    }
    __syncthreads();
    for (uint32_t e = o; e < taken; e += 8) {  // the taken leaves are the first `taken` of the sorted list
        const uint32_t node = s_sorted_node[e], pos = s_pos[e];
        const DNode nd = sp.nodes[node];
        const uint32_t *ids = sp.desc + nd.a;
        if (!sp.filter_bits) {
            for (uint32_t i = j; i < nd.b; i += 8) my_nns[pos + i] = ids[i];
        } else {
            copy_filtered(sp, ids, nd.b, my_nns + pos, j);
        }
        if (j == 0) record_visit(sink, node, q, pos, s_sorted_n[e]);
    }
    if (lane == 0) {
        nns_count[q] = ids_taken;
        overflow[q] = 0;
        if (sp.stats) atomicAdd(&sp.stats[kWaveHeap == 256 ? SS_WAVE_SMALL : SS_WAVE_BIG], 1u);
    }
}

// A work unit of the leaf-tile re-rank: <= kUnitVisits visits of one node (k_leaf_scan_* / k_units_small / the block descent of
// a single query write them, k_leaf_tiles* read them).
static constexpr uint32_t kUnitVisits = 16;
struct TileUnit {
    uint32_t node, first, n_vis, pad;  // sorted[first .. first + n_vis)
};
static constexpr uint32_t kTileSlab = 128;       // rows per block and unit of more than 8 visits (twice that up to 8): k_leaf_tiles*
static constexpr uint32_t kTileSmallSlab = 64;   // rows per block of k_leaf_tiles16 in a small submission (32 octets x 2 rows)
// rows of a unit's leaf one block of k_leaf_tiles16 takes (small: the small submissions' variant of the kernel)
__host__ __device__ constexpr uint32_t tile_slab_rows(uint32_t n_vis, bool small) {
    return small && n_vis <= 2 ? kTileSmallSlab : (n_vis > 8 ? kTileSlab : 2 * kTileSlab);
}

// Blocks 1 .. n_h16 of the launch are the queries' binary16 copies (k_queries_h16's work, wanted by the same consumer — the
// leaf tiles — and as independent of the descent): one launch less on a path that is mostly launches.
struct QueriesH16 {
    const uint8_t *qvecs;
    uint64_t qstride;
    uint32_t dims, hpitch;
    uint16_t *q16;
    float4 *qstats;
};
// (one wave per query; `lane` = the thread's index in it)
__device__ __forceinline__ void query_h16(uint32_t q, uint32_t lane, const uint8_t *__restrict__ qvecs, uint64_t qstride, uint32_t dims,
                                          uint32_t hpitch, uint16_t *__restrict__ q16, float4 *__restrict__ qstats) {
    const float *v = reinterpret_cast<const float *>(qvecs + (uint64_t)q * qstride);
    uint16_t *out = q16 + (uint64_t)q * hpitch;
    float sa = 0.f, sb = 0.f, sc = 0.f;
    uint32_t xbits = 0u;
    for (uint32_t i = lane; i < hpitch; i += 64) {
        const float x = i < dims ? v[i] : 0.0f;
        const _Float16 h = to_shadow_half(x);
        const float y = (float)h, d = x - y;
        sa += y * y;
        sb += d * d;
        sc += x * x;
        xbits = max(xbits, __float_as_uint(x) & 0x7FFFFFFFu);
        out[i] = __builtin_bit_cast(uint16_t, h);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_xor(sa, off);
        sb += __shfl_xor(sb, off);
        sc += __shfl_xor(sc, off);
        xbits = max(xbits, (uint32_t)__shfl_xor((int)xbits, off));
    }
    if (lane == 0) {
        const float up = 1.0f + (float)(hpitch + 64u) * 1.2e-7f;
        const bool tiny = xbits != 0u && xbits < kTinyBits;  // squares underflow: the measured norms would lie
        const float inf = __uint_as_float(0x7F800000u);
        qstats[q] = tiny ? make_float4(inf, inf, inf, 0.f) : make_float4(sqrtf(sa) * up, sqrtf(sb) * up, sqrtf(sc) * up, 0.f);
    }
}

__global__ __launch_bounds__(64) void k_queries_h16(const uint8_t *__restrict__ qvecs, uint64_t qstride, uint32_t dims,
                                                    uint32_t hpitch, uint16_t *__restrict__ q16, float4 *__restrict__ qstats) {
    query_h16(blockIdx.x, threadIdx.x, qvecs, qstride, dims, hpitch, q16, qstats);
}


// A submission of ONE query on the leaf-tile path: its visits are nobody else's, so every taken leaf is a unit of one visit and
// the block descent writes units and the (trivially sorted) visit list itself — and makes the query's binary16 copy while it
// is at it: the call loses the launch that would sort a dozen visits.  units == nullptr: off.
struct SingleQueryOut {
    TileUnit *units;
    Visit *sorted;
    uint32_t *n_units;
    QueriesH16 h16;  // q16 == nullptr: no binary16 copy wanted
    // k_descend_multi, no candidate filter: the leaves' ids are NOT copied into the candidate buffer by the descent's last block
    // (10 000 ids through one compute unit: 7 us) — TileUnit::pad carries 1 + the leaf's first id in the blob and the ~150
    // blocks of k_leaf_tiles16<true> copy the ids of the rows they evaluate (the selection reads them from the buffer as before)
    bool ids_by_tiles;
    // k_descend_multi, a few queries a call (round 6): query q's units and visits at [q * per_query, ...), its unit count at
    // n_units[q] — every query its own list, no launch to sort the visits of all queries by leaf (0: one query, lists at 0)
    uint32_t per_query;
};
// ---- one BLOCK per query: the small submissions ------------------------------------------------------------------
// arroy's API takes one query per call (src/reader.rs:46-75) and the wave descent's time does not depend on how many queries a
// call brings: ~66 dependent pops of 4.5 us each = 0.3 ms for ONE query, two thirds of the whole call (round-5 kernel trace of
// nq = 1).  With few queries the device is idle, so a query gets kOct = 32 octets instead of 8: the octets search the trees
// t = octet (mod 32) — one tree each for the usual 10 - 30 trees — and the chain shrinks to the pops of ONE tree (~20).  Same
// rules as k_descend_wave, the cross-octet reductions (largest queued key, ids held above it, failure) through LDS and block
// barriers instead of wave shuffles; what it cannot hold (kHeap queue entries / kLeaves leaves per octet, equal keys of two
// octets across the cut) is left to the passes behind it exactly like the wave kernel's leftovers.
template <uint32_t kOct, uint32_t kHeap, uint32_t kLeaves>
constexpr size_t block_descend_lds_bytes() {
    return (size_t)kOct * kHeap * 8 + (size_t)kOct * kLeaves * (8 + 8 + 4 + 4 + 4 + 4 + 4) + 256;
}
template <uint32_t kOct, uint32_t kHeap, uint32_t kLeaves, uint32_t kPops>
__global__ __launch_bounds__(8 * kOct) void k_descend_block(DataView nv, SearchParams sp, uint32_t nq,
                                                            const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                            const float *__restrict__ qhdrs, uint32_t *__restrict__ nns,
                                                            uint32_t *__restrict__ nns_count, uint32_t *__restrict__ overflow,
                                                            VisitSink sink, bool last_pass, const float *__restrict__ raw_queries,
                                                            float *__restrict__ qhdrs_out, SingleQueryOut single) {
    constexpr uint32_t kThreads = 8 * kOct, kWaves = kThreads / 64, kCap = kOct * kLeaves;
    static_assert((kThreads & (kThreads - 1)) == 0 && kCap >= kThreads, "power-of-two block, sort network at least as wide");
    extern __shared__ uint64_t s_blk_lds[];
    uint64_t(*s_heap)[kHeap] = reinterpret_cast<uint64_t(*)[kHeap]>(s_blk_lds);
    uint64_t(*s_leaf)[kLeaves] = reinterpret_cast<uint64_t(*)[kLeaves]>(s_blk_lds + (size_t)kOct * kHeap);  // key word << 32 | node
    uint64_t *s_sorted = s_blk_lds + (size_t)kOct * kHeap + kCap;
    uint32_t(*s_leaf_n)[kLeaves] = reinterpret_cast<uint32_t(*)[kLeaves]>(s_sorted + kCap);
    uint32_t *s_sorted_n = reinterpret_cast<uint32_t *>(s_sorted + kCap) + kCap;
    uint32_t *s_pos = s_sorted_n + kCap, *s_sorted_node = s_pos + kCap;
    uint32_t *s_red = s_sorted_node + kCap;  // 64 words of scratch for the block reductions
    uint32_t(*s_leaf_a)[kLeaves] = reinterpret_cast<uint32_t(*)[kLeaves]>(s_red + 64);  // first id of the leaf in the blob (DNode::a)
    const uint32_t q = blockIdx.x, tid = threadIdx.x, o = tid >> 3, j = tid & 7u, wave = tid >> 6, wl = tid & 63u;
    if (q >= nq) return;
    // block-wide max / sum / or of one value per thread (all threads call; two barriers each)
    auto block_max = [&](uint32_t v) {
        for (uint32_t d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
        __syncthreads();
        if (wl == 0) s_red[wave] = v;
        __syncthreads();
        uint32_t r = 0;
        for (uint32_t w = 0; w < kWaves; w++) r = max(r, s_red[w]);
        return r;
    };
    auto block_sum = [&](uint32_t v) {
        for (uint32_t d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        __syncthreads();
        if (wl == 0) s_red[wave] = v;
        __syncthreads();
        uint32_t r = 0;
        for (uint32_t w = 0; w < kWaves; w++) r += s_red[w];
        return r;
    };
    uint64_t *heap = s_heap[o];
    // the query leaf in LDS (behind everything else; qstride bytes): every margin of the descent reads it
    uint4 *s_q4 = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(s_blk_lds) + block_descend_lds_bytes<kOct, kHeap, kLeaves>());
    LeafHdr qh;
    if (raw_queries) {
        // `QueryBuilder::by_vector` (src/reader.rs:64-75) done here for a small submission of an f32 metric: what k_prepare_queries
        // would have left in qvecs / qhdrs for the kernels behind this one, and the leaf straight into LDS (raw_queries may be the
        // caller's pinned staging buffer, read over the link)
        const float *src = raw_queries + (uint64_t)q * nv.dims;
        float *dst = reinterpret_cast<float *>(const_cast<uint8_t *>(qvecs) + (uint64_t)q * qstride), *s_q = reinterpret_cast<float *>(s_q4);
        for (uint32_t i0 = tid; i0 < nv.pitch; i0 += 8 * kThreads) {  // (eight reads over the link in flight per thread, not one)
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) v[u] = i0 + u * kThreads < nv.dims ? src[i0 + u * kThreads] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                if (i0 + u * kThreads < nv.pitch) {
                    s_q[i0 + u * kThreads] = v[u];
                    dst[i0 + u * kThreads] = v[u];
                }
        }
        __syncthreads();
        if (nv.metric == AH_COSINE && tid < 8) {
            const float norm = f_sqrt(octet_reduce_any<OP_DOT>(s_q, s_q, nv.dims, tid));
            if (tid == 0) s_red[63] = __float_as_uint(norm);
        } else if (tid == 0) {
            s_red[63] = 0u;
        }
        __syncthreads();
        qh = LeafHdr{__uint_as_float(s_red[63]), 0.0f};
        if (tid == 0) {
            qhdrs_out[2 * (uint64_t)q] = qh.h0;
            qhdrs_out[2 * (uint64_t)q + 1] = 0.0f;
        }
    } else {
        const uint4 *g_q4 = reinterpret_cast<const uint4 *>(qvecs + (uint64_t)q * qstride);
        for (uint32_t i = tid; i < (uint32_t)(qstride >> 4); i += kThreads) s_q4[i] = g_q4[i];
        __syncthreads();
        qh = LeafHdr{qhdrs[2 * (uint64_t)q], qhdrs[2 * (uint64_t)q + 1]};
    }
    const void *qvec = s_q4;
    if (single.units && single.h16.q16 && tid < 64)  // (the leaf in global memory was written before the barriers above, or by an earlier kernel)
        query_h16(q, tid, single.h16.qvecs, single.h16.qstride, single.h16.dims, single.h16.hpitch, single.h16.q16, single.h16.qstats);
    uint32_t *my_nns = nns + (uint64_t)q * sp.nns_stride;
    // The octet's queue is an UNSORTED array in LDS, popped by an arg-max over the octet's eight lanes: the keys are unique (the
    // node is their low word), so the pops come in the order a binary heap would give them, and a queue of the ~10 - 30 entries
    // one tree holds costs two or three LDS reads per lane and three shuffles instead of a lane-0 sift of dependent LDS round
    // trips.  hn / nl / failed are octet-uniform (every lane keeps them; lane 0 writes LDS).
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t hn = 0, nl = 0;
    bool failed = false;
    for (uint32_t t = o; t < sp.n_trees; t += kOct) {
        if (hn == kHeap) {
            failed = true;
            break;
        }
        if (j == 0) heap[hn] = ((uint64_t)0xFF800000u << 32) | sp.roots[t];
        hn++;
    }
    auto octet_sync = [] {  // lane 0's LDS stores before the other lanes' loads (one wave: program order is enough for the hardware)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint64_t best = 0;
    uint32_t bi = kNone;
    auto queue_argmax = [&] {  // (best, bi) <- the largest key queued by this octet and where it sits; bi = kNone: nothing queued
        octet_sync();
        best = 0;
        bi = kNone;
        for (uint32_t i = j; i < hn; i += 8) {
            const uint64_t k = heap[i];
            if (bi == kNone || k > best) {
                best = k;
                bi = i;
            }
        }
#pragma unroll
        for (uint32_t d = 1; d < 8; d <<= 1) {
            const uint32_t o_hi = __shfl_xor((uint32_t)(best >> 32), d, 8), o_lo = __shfl_xor((uint32_t)best, d, 8);
            const uint32_t o_i = __shfl_xor(bi, d, 8);
            const uint64_t ok = ((uint64_t)o_hi << 32) | o_lo;
            if (o_i != kNone && (bi == kNone || ok > best)) {
                best = ok;
                bi = o_i;
            }
        }
    };
    // the two children of the split popped last: their records are requested beside the split's normal, and the next pop of
    // this tree is nearly always one of them — one dependent trip to memory per level instead of two
    uint32_t c_id[2] = {kNone, kNone};
    DNode c_nd[2] = {};
    queue_argmax();
    uint32_t threshold = 0;
    bool all_settled = false;
    for (;;) {
        // kPops pops of every octet per block-wide check: a pop beyond the point where the check would have stopped only settles
        // more leaves than needed (the cut below is by the sorted prefix), it never changes which ones are taken
#pragma unroll 1
        for (uint32_t rep = 0; rep < kPops; rep++) {
            if (bi != kNone && !failed) {  // octet-uniform
                const uint32_t node = (uint32_t)best, key_word = (uint32_t)(best >> 32);
                hn--;
                if (j == 0 && bi != hn) heap[bi] = heap[hn];
                DNode nd;
                if (node == c_id[0]) nd = c_nd[0];
                else if (node == c_id[1]) nd = c_nd[1];
                else nd = sp.nodes[node];
                if ((nd.kind & 0xFFu) == AH_NODE_DESCENDANTS) {
                    const uint32_t kept = sp.leaf_kept ? sp.leaf_kept[node] : nd.b;
                    if (kept) {
                        if (nl == kLeaves) {
                            failed = true;
                        } else {
                            if (j == 0) {
                                s_leaf[o][nl] = ((uint64_t)key_word << 32) | node;
                                s_leaf_n[o][nl] = kept;
                                s_leaf_a[o][nl] = nd.a;
                            }
                            nl++;
                        }
                    }
                } else if ((nd.kind & 0xFFu) != 0) {
                    c_id[0] = nd.a;
                    c_id[1] = nd.b;
                    c_nd[0] = sp.nodes[nd.a];
                    c_nd[1] = sp.nodes[nd.b];
                    float margin = 0.0f;
                    if (nd.kind & 0x100u) margin = descent_margin<true>(nv, nd.c, qvec, qh, j);
                    if (hn + 2 > kHeap) {
                        failed = true;
                    } else {
                        const float dist = key_to_dist(key_word);
                        const float pl = rust_min(-margin, dist), pr = rust_min(margin, dist);
                        if (j == 0) {
                            heap[hn] = ((uint64_t)orderable_key(pl) << 32) | nd.a;
                            heap[hn + 1] = ((uint64_t)orderable_key(pr) << 32) | nd.b;
                        }
                        hn += 2;
                    }
                }
                queue_argmax();
            }
        }
        // the largest key still queued anywhere, whether anything is queued at all, a failure — one barrier; then the ids held by
        // the leaves strictly above that key — a second one.  (Two scratch rows alternate: a wave can be at most one reduction
        // ahead of the slowest, so a row is never overwritten before everybody has read it.)
        const bool has = bi != kNone;
        uint32_t top = has ? (uint32_t)(best >> 32) : 0u, flags = (has ? 1u : 0u) | (failed ? 2u : 0u);
        for (uint32_t d = 32; d >= 8; d >>= 1) {
            top = max(top, (uint32_t)__shfl_xor((int)top, d, 64));
            flags |= (uint32_t)__shfl_xor((int)flags, d, 64);
        }
        if (wl == 0) {
            s_red[wave] = top;
            s_red[8 + wave] = flags;
        }
        __syncthreads();
        top = flags = 0;
        for (uint32_t w = 0; w < kWaves; w++) {
            top = max(top, s_red[w]);
            flags |= s_red[8 + w];
        }
        if (flags & 2u) {
            if (tid == 0) {
                nns_count[q] = 0;
                overflow[q] = 1;
                if (last_pass && sink.err) atomicOr(sink.err, 16u);  // nobody behind this kernel: the submission takes the long way
            }
            return;
        }
        const bool any_queued = (flags & 1u) != 0;
        uint32_t held = 0;
        for (uint32_t i = j; i < nl; i += 8)
            if (!any_queued || (uint32_t)(s_leaf[o][i] >> 32) > top) held += s_leaf_n[o][i];
        for (uint32_t d = 32; d > 0; d >>= 1) held += __shfl_xor(held, d, 64);
        if (wl == 0) s_red[16 + wave] = held;
        __syncthreads();
        held = 0;
        for (uint32_t w = 0; w < kWaves; w++) held += s_red[16 + w];
        if (!any_queued || held >= sp.search_k) {
            threshold = top;
            all_settled = !any_queued;
            break;
        }
    }
    // the settled leaves, sorted by decreasing key
    const uint32_t nl_o = __shfl(nl, 0, 8);
    uint32_t mine = 0;
    for (uint32_t i = j; i < nl_o; i += 8) mine += (all_settled || (uint32_t)(s_leaf[o][i] >> 32) > threshold) ? 1u : 0u;
    for (uint32_t d = 4; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 8);  // settled leaves of this octet
    __syncthreads();
    if (j == 0) s_pos[o] = mine;  // (s_pos is free until the prefix below)
    __syncthreads();
    uint32_t first = 0, n_settled = 0;
    for (uint32_t oo = 0; oo < kOct; oo++) {
        const uint32_t c = s_pos[oo];
        first += oo < o ? c : 0u;
        n_settled += c;
    }
    __syncthreads();
    // Equal keys: inside one octet the order is the octet's own pop order (see k_descend_wave): the sort key is
    // key word << 32 | octet << 16 | 0xFFFF - (index in the octet's list)
    if (j == 0) {
        uint32_t w = first;
        for (uint32_t i = 0; i < nl; i++)
            if (all_settled || (uint32_t)(s_leaf[o][i] >> 32) > threshold) {
                s_sorted[w] = (s_leaf[o][i] & 0xFFFFFFFF00000000ull) | (o << 16) | (0xFFFFu - i);
                s_sorted_node[w] = (uint32_t)s_leaf[o][i];
                s_sorted_n[w] = s_leaf_n[o][i];
                w++;
            }
    }
    uint32_t p2 = kThreads;
    while (p2 < n_settled) p2 <<= 1;
    __syncthreads();
    for (uint32_t t = n_settled + tid; t < p2; t += kThreads) {
        s_sorted[t] = 0;
        s_sorted_n[t] = 0;
        s_sorted_node[t] = 0;
    }
    // descending by key: a leaf's place is the number of larger keys (unique: octet and pop index are their low word) — n_settled
    // broadcast reads per leaf and two barriers, where a sorting network over two or three dozen leaves was fifteen
    {
        constexpr uint32_t kMine = kCap / kThreads;
        uint64_t my_key[kMine];
        uint32_t my_n[kMine], my_node[kMine], my_rank[kMine];
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < kMine; r++) {
            const uint32_t e = tid + r * kThreads;
            my_rank[r] = 0;
            if (e < n_settled) {
                my_key[r] = s_sorted[e];
                my_n[r] = s_sorted_n[e];
                my_node[r] = s_sorted_node[e];
                for (uint32_t g = 0; g < n_settled; g++) my_rank[r] += s_sorted[g] > my_key[r] ? 1u : 0u;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < kMine; r++)
            if (tid + r * kThreads < n_settled) {
                s_sorted[my_rank[r]] = my_key[r];
                s_sorted_n[my_rank[r]] = my_n[r];
                s_sorted_node[my_rank[r]] = my_node[r];
            }
    }
    __syncthreads();
    // `if nns.len() >= search_k { break }` before every pop: leaf i is taken iff the leaves before it hold < search_k ids
    const uint32_t per = p2 / kThreads;
    uint32_t local = 0;
    for (uint32_t i = 0; i < per; i++) local += s_sorted_n[tid * per + i];
    uint32_t incl = local;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (wl >= d) incl += v;
    }
    if (wl == 63) s_red[32 + wave] = incl;  // (the upper half of the scratch: block_max / block_sum use the lower)
    __syncthreads();
    uint32_t before = incl - local;
    for (uint32_t w = 0; w < wave; w++) before += s_red[32 + w];
    uint32_t taken = 0, ids_taken = 0;
    bool tie = false;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t e = tid * per + i;
        if (e < n_settled && before < sp.search_k) {
            s_pos[e] = before;
            taken++;
            ids_taken = before + s_sorted_n[e];
            if (ids_taken >= sp.search_k) {  // the leaf that reaches search_k: equal keys of another octet around it -> sequential queue
                const uint32_t kw = (uint32_t)(s_sorted[e] >> 32), oct = (uint32_t)s_sorted[e] >> 16;
                for (uint32_t g = e + 1; g < n_settled && (uint32_t)(s_sorted[g] >> 32) == kw; g++)
                    if (((uint32_t)s_sorted[g] >> 16) != oct) tie = true;
                for (uint32_t g = e; g-- > 0 && (uint32_t)(s_sorted[g] >> 32) == kw;)
                    if (((uint32_t)s_sorted[g] >> 16) != oct) tie = true;
            }
        }
        before += s_sorted_n[e];
    }
    taken = block_sum(taken);
    ids_taken = block_max(ids_taken);
    const uint32_t any_tie = block_max(tie ? 1u : 0u);
    if (any_tie || ids_taken > sp.nns_stride) {
        if (tid == 0) {
            nns_count[q] = 0;
            overflow[q] = 1;
            if (last_pass && sink.err) atomicOr(sink.err, 16u);
        }
        return;
    }
    __syncthreads();
    if (!sp.filter_bits) {
        // The taken leaves are the first `taken` of the sorted list and their ids one flat range [0, ids_taken) of the query's
        // candidate buffer: position p belongs to the last leaf whose first position is <= p.  Every thread copies positions
        // p = tid (mod 256) — a leaf per octet left 12 octets copying ~800 ids each, 8 lanes wide, while 20 watched.
        // (a leaf's first id in the blob was kept when the leaf was popped: the sort key says which octet popped it and when)
        uint32_t my_first[(kCap + kThreads - 1) / kThreads];
#pragma unroll
        for (uint32_t r = 0; r < (kCap + kThreads - 1) / kThreads; r++) {
            const uint32_t e = tid + r * kThreads;
            my_first[r] = 0;
            if (e < taken) {
                const uint32_t w = (uint32_t)s_sorted[e];
                my_first[r] = s_leaf_a[w >> 16][0xFFFFu - (w & 0xFFFFu)];
                if (single.units) {
                    single.units[e] = TileUnit{s_sorted_node[e], e, 1u, 0u};
                    single.sorted[e] = Visit{s_sorted_node[e], q, s_pos[e], s_sorted_n[e]};
                } else {
                    record_visit(sink, s_sorted_node[e], q, s_pos[e], s_sorted_n[e]);
                }
            }
        }
        __syncthreads();
        uint32_t *s_first = reinterpret_cast<uint32_t *>(s_sorted);  // (the sort keys are dead now)
#pragma unroll
        for (uint32_t r = 0; r < (kCap + kThreads - 1) / kThreads; r++)
            if (tid + r * kThreads < taken) s_first[tid + r * kThreads] = my_first[r];
        __syncthreads();
        // positions p = tid, tid + 256, ... rise: the leaf of p is found once and then walked forward (a search per position
        // was 24 000 cycles of dependent LDS reads per thread)
        constexpr uint32_t kFly = 20;
        uint32_t e = 0;
        {
            uint32_t hi = taken;  // the last e with s_pos[e] <= tid
            while (hi - e > 1) {
                const uint32_t mid = (e + hi) >> 1;
                if (s_pos[mid] <= tid) e = mid;
                else hi = mid;
            }
        }
        uint32_t next_first = e + 1 < taken ? s_pos[e + 1] : 0xFFFFFFFFu, base = taken ? s_first[e] - s_pos[e] : 0u;
        for (uint32_t p0 = tid; p0 < ids_taken; p0 += kFly * kThreads) {
            uint32_t id[kFly];
#pragma unroll
            for (uint32_t u = 0; u < kFly; u++) {
                const uint32_t p = p0 + u * kThreads;
                while (p >= next_first) {
                    e++;
                    next_first = e + 1 < taken ? s_pos[e + 1] : 0xFFFFFFFFu;
                    base = s_first[e] - s_pos[e];
                }
                id[u] = p < ids_taken ? sp.desc[base + p] : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t u = 0; u < kFly; u++)
                if (p0 + u * kThreads < ids_taken) my_nns[p0 + u * kThreads] = id[u];
        }
    } else {
        for (uint32_t e = o; e < taken; e += kOct) {
            const uint32_t node = s_sorted_node[e], pos = s_pos[e];
            const DNode nd = sp.nodes[node];
            copy_filtered(sp, sp.desc + nd.a, nd.b, my_nns + pos, j);
            if (j == 0) {
                if (single.units) {
                    single.units[e] = TileUnit{node, e, 1u, 0u};
                    single.sorted[e] = Visit{node, q, pos, s_sorted_n[e]};
                } else {
                    record_visit(sink, node, q, pos, s_sorted_n[e]);
                }
            }
        }
    }
    if (tid == 0) {
        nns_count[q] = ids_taken;
        overflow[q] = 0;
        if (sp.stats) atomicAdd(&sp.stats[SS_BLOCK], 1u);
        if (single.units) {
            *single.n_units = taken;
            if (sink.total) *sink.total = taken;
            if (sp.stats && taken) {
                atomicAdd(&sp.stats[SS_VISITS], taken);
                atomicAdd(&sp.stats[SS_UNITS_4], taken);
            }
        }
    }
}

// ---- several BLOCKS per query: one query on more than one CU (round 6) --------------------------------------------------
// k_descend_block runs a whole query on ONE compute unit: every pop round sends ~20 normals of 6 KB through one texture
// path (0.8 us of a ~3.7 us round) and 255 CUs watch.  The candidate set is order-independent (k_descend_wave: the leaves
// in decreasing key order until search_k ids are held), so the trees can be dealt over G blocks, t = block (mod G), each
// a WAVE of eight octets with a queue each — no block barrier inside the loop — and the blocks only have to agree on when
// to stop.  What block g needs for that: a key x with "the leaves already popped ANYWHERE with key >= x hold >= search_k
// ids" (then the cut T* of the sequential loop is >= x) and every key still queued in g below x (then g has popped all of
// its leaves with key >= T*, ties included).  Leaves another block has not reported yet only delay the stop (more pops
// than needed, never other candidates), so the exchange is one-way and needs no fence: every popped leaf is published as
// ONE 64-bit word (key word << 32 | ids) by an agent-scope atomic store into a slot that was zero, and a second wave of
// every block — the descent wave never waits for memory it does not need — polls the other blocks' next slots, keeps all
// known leaves in LDS and recomputes x.  The last block to finish (a counter) gathers every list, orders the leaves as
// k_descend_block does (octet order inside a list, equal keys of two lists across the cut -> the sequential descent) and
// copies the ids; it also wipes what the query wrote, so the control block (Context::d_multi) is zero between calls.
static constexpr uint32_t kMultiMaxBlocks = 16, kMultiLeaves = 32, kMultiLists = kMultiMaxBlocks * 8, kMultiMaxQueries = 32;
static constexpr uint32_t kMultiKnown = 1024, kMultiCap = 1024;
struct MultiCtl {
    uint32_t done, failed, pad[14];
    unsigned long long leaf[kMultiLists][kMultiLeaves];  // key word << 32 | ids the leaf adds (never 0); list = block * 8 + octet
    unsigned long long info[kMultiLists][kMultiLeaves];  // node << 32 | first id of the leaf in the blob
    uint32_t trace[kMultiMaxBlocks][8];                  // AH_SEARCH_MULTI_TRACE: 10 ns ticks since the block started (see the kernel)
    uint32_t sel_trace[16];                              // ... and of the selection kernel behind it (k_search_select_screened<*, true>)
    uint32_t tile_trace[8];                              // ... and of the tile launch between them (maxima over its blocks)
};
size_t multi_ctl_bytes() { return (size_t)kMultiMaxQueries * sizeof(MultiCtl); }
template <uint32_t kHeap>
constexpr size_t multi_descend_lds_bytes() {
    return (size_t)8 * kHeap * 8 + (size_t)8 * kMultiLeaves * (8 + 4 + 4) + (size_t)kMultiKnown * 8 + (size_t)kMultiCap * (8 + 4 + 4 + 4 + 4) +
           64 * 4 + 32 * 4;
}
enum MultiWord { MW_N_KNOWN = 0, MW_STOP_X, MW_FINISH, MW_REMOTE_FAILED, MW_LOCAL_FAILED, MW_LAST, MW_N_MERGED };
__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long lds_load64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long dev_load64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dev_store64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <uint32_t kHeap, uint32_t kPops>
__global__ __launch_bounds__(256) void k_descend_multi(DataView nv, SearchParams sp, uint32_t nq, const uint8_t *__restrict__ qvecs,
                                                       uint64_t qstride, const float *__restrict__ qhdrs, uint32_t *__restrict__ nns,
                                                       uint32_t *__restrict__ nns_count, uint32_t *__restrict__ overflow, VisitSink sink,
                                                       const float *__restrict__ raw_queries, float *__restrict__ qhdrs_out,
                                                       SingleQueryOut single, MultiCtl *__restrict__ ctls, uint32_t G, uint32_t trace_on) {
    constexpr uint32_t kThreads = 256, kLeaves = kMultiLeaves, kCap = kMultiCap;
    extern __shared__ uint64_t s_multi_lds[];
    uint64_t(*s_heap)[kHeap] = reinterpret_cast<uint64_t(*)[kHeap]>(s_multi_lds);
    uint64_t(*s_leaf)[kLeaves] = reinterpret_cast<uint64_t(*)[kLeaves]>(s_multi_lds + (size_t)8 * kHeap);  // key word << 32 | node
    unsigned long long *s_known = reinterpret_cast<unsigned long long *>(s_multi_lds + (size_t)8 * kHeap + 8 * kLeaves);
    uint64_t *s_sorted = s_multi_lds + (size_t)8 * kHeap + 8 * kLeaves + kMultiKnown;
    uint32_t *s_sorted_n = reinterpret_cast<uint32_t *>(s_sorted + kCap);
    uint32_t *s_pos = s_sorted_n + kCap, *s_sorted_node = s_pos + kCap, *s_sorted_first = s_sorted_node + kCap;
    uint32_t(*s_leaf_n)[kLeaves] = reinterpret_cast<uint32_t(*)[kLeaves]>(s_sorted_first + kCap);
    uint32_t(*s_leaf_a)[kLeaves] = reinterpret_cast<uint32_t(*)[kLeaves]>(s_sorted_first + kCap + 8 * kLeaves);
    uint32_t *s_red = s_sorted_first + kCap + 16 * kLeaves;  // 64 words of scratch for the block reductions
    uint32_t *s_mw = s_red + 64;                             // 32 control words (MultiWord)
    const uint32_t q = blockIdx.x / G, g = blockIdx.x - q * G, tid = threadIdx.x, o = (tid >> 3) & 7u, j = tid & 7u, wave = tid >> 6, wl = tid & 63u;
    if (q >= nq) return;
    MultiCtl *ctl = ctls + q;
    const uint64_t t_start = trace_on ? wall_clock64() : 0ull;
    auto stamp = [&](uint32_t slot) {  // (AH_SEARCH_MULTI_TRACE; thread 0 of the block; never read by the kernels)
        if (trace_on && tid == 0) ctl->trace[g][slot] = (uint32_t)(wall_clock64() - t_start);
    };
    constexpr uint32_t kWaves = kThreads / 64;
    auto block_max = [&](uint32_t v) {
        for (uint32_t d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
        __syncthreads();
        if (wl == 0) s_red[wave] = v;
        __syncthreads();
        uint32_t r = 0;
        for (uint32_t w = 0; w < kWaves; w++) r = max(r, s_red[w]);
        return r;
    };
    auto block_sum = [&](uint32_t v) {
        for (uint32_t d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        __syncthreads();
        if (wl == 0) s_red[wave] = v;
        __syncthreads();
        uint32_t r = 0;
        for (uint32_t w = 0; w < kWaves; w++) r += s_red[w];
        return r;
    };
    for (uint32_t i = tid; i < kMultiKnown; i += kThreads) s_known[i] = 0ull;
    if (tid < 32) s_mw[tid] = 0u;
    // the query leaf in LDS (behind everything else; qstride bytes) — as in k_descend_block; block 0 of the query leaves what the
    // kernels behind the descent read (the prepared leaf, its header, the binary16 copy) in global memory
    uint4 *s_q4 = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(s_multi_lds) + multi_descend_lds_bytes<kHeap>());
    LeafHdr qh;
    if (raw_queries) {
        const float *src = raw_queries + (uint64_t)q * nv.dims;
        float *dst = reinterpret_cast<float *>(const_cast<uint8_t *>(qvecs) + (uint64_t)q * qstride), *s_q = reinterpret_cast<float *>(s_q4);
        for (uint32_t i0 = tid; i0 < nv.pitch; i0 += 8 * kThreads) {
            float v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) v[u] = i0 + u * kThreads < nv.dims ? src[i0 + u * kThreads] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                if (i0 + u * kThreads < nv.pitch) {
                    s_q[i0 + u * kThreads] = v[u];
                    if (g == 0) dst[i0 + u * kThreads] = v[u];
                }
        }
        __syncthreads();
        if (nv.metric == AH_COSINE && tid < 8) {
            const float norm = f_sqrt(octet_reduce_any<OP_DOT>(s_q, s_q, nv.dims, tid));
            if (tid == 0) s_red[63] = __float_as_uint(norm);
        } else if (tid == 0) {
            s_red[63] = 0u;
        }
        __syncthreads();
        qh = LeafHdr{__uint_as_float(s_red[63]), 0.0f};
        if (tid == 0 && g == 0) {
            qhdrs_out[2 * (uint64_t)q] = qh.h0;
            qhdrs_out[2 * (uint64_t)q + 1] = 0.0f;
        }
    } else {
        const uint4 *g_q4 = reinterpret_cast<const uint4 *>(qvecs + (uint64_t)q * qstride);
        for (uint32_t i = tid; i < (uint32_t)(qstride >> 4); i += kThreads) s_q4[i] = g_q4[i];
        __syncthreads();
        qh = LeafHdr{qhdrs[2 * (uint64_t)q], qhdrs[2 * (uint64_t)q + 1]};
    }
    const void *qvec = s_q4;
    stamp(0);
    // (wave 3 idles until the merge: the binary16 copy of a single query is its business; it reads the leaf block 0 wrote above)
    if (g == 0 && single.units && single.h16.q16 && wave == 3)
        query_h16(q, wl, single.h16.qvecs, single.h16.qstride, single.h16.dims, single.h16.hpitch, single.h16.q16, single.h16.qstats);
    uint32_t *my_nns = nns + (uint64_t)q * sp.nns_stride;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    if (wave == 0) {
        // ---- the descent: eight octets, a queue each (k_descend_block's loop, its block-wide reductions as shuffles) ----
        uint64_t *heap = s_heap[o];
        uint32_t hn = 0, nl = 0, published = 0;
        bool failed = false;
        for (uint32_t t = g + G * o; t < sp.n_trees; t += G * 8) {
            if (hn == kHeap) {
                failed = true;
                break;
            }
            if (j == 0) heap[hn] = ((uint64_t)0xFF800000u << 32) | sp.roots[t];
            hn++;
        }
        auto octet_sync = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        uint64_t best = 0;
        uint32_t bi = kNone;
        auto queue_argmax = [&] {
            octet_sync();
            best = 0;
            bi = kNone;
            for (uint32_t i = j; i < hn; i += 8) {
                const uint64_t k = heap[i];
                if (bi == kNone || k > best) {
                    best = k;
                    bi = i;
                }
            }
#pragma unroll
            for (uint32_t d = 1; d < 8; d <<= 1) {
                const uint32_t o_hi = __shfl_xor((uint32_t)(best >> 32), d, 8), o_lo = __shfl_xor((uint32_t)best, d, 8);
                const uint32_t o_i = __shfl_xor(bi, d, 8);
                const uint64_t ok = ((uint64_t)o_hi << 32) | o_lo;
                if (o_i != kNone && (bi == kNone || ok > best)) {
                    best = ok;
                    bi = o_i;
                }
            }
        };
        uint32_t c_id[2] = {kNone, kNone};
        DNode c_nd[2] = {};
        queue_argmax();
        for (;;) {
#pragma unroll 1
            for (uint32_t rep = 0; rep < kPops; rep++) {
                if (bi != kNone && !failed) {  // octet-uniform
                    const uint32_t node = (uint32_t)best, key_word = (uint32_t)(best >> 32);
                    hn--;
                    if (j == 0 && bi != hn) heap[bi] = heap[hn];
                    DNode nd;
                    if (node == c_id[0]) nd = c_nd[0];
                    else if (node == c_id[1]) nd = c_nd[1];
                    else nd = sp.nodes[node];
                    if ((nd.kind & 0xFFu) == AH_NODE_DESCENDANTS) {
                        const uint32_t kept = sp.leaf_kept ? sp.leaf_kept[node] : nd.b;
                        if (kept) {
                            if (nl == kLeaves) {
                                failed = true;
                            } else {
                                if (j == 0) {
                                    s_leaf[o][nl] = ((uint64_t)key_word << 32) | node;
                                    s_leaf_n[o][nl] = kept;
                                    s_leaf_a[o][nl] = nd.a;
                                }
                                nl++;
                            }
                        }
                    } else if ((nd.kind & 0xFFu) != 0) {
                        c_id[0] = nd.a;
                        c_id[1] = nd.b;
                        c_nd[0] = sp.nodes[nd.a];
                        c_nd[1] = sp.nodes[nd.b];
                        float margin = 0.0f;
                        if (nd.kind & 0x100u) margin = descent_margin<true>(nv, nd.c, qvec, qh, j);
                        if (hn + 2 > kHeap) {
                            failed = true;
                        } else {
                            const float dist = key_to_dist(key_word);
                            const float pl = rust_min(-margin, dist), pr = rust_min(margin, dist);
                            if (j == 0) {
                                heap[hn] = ((uint64_t)orderable_key(pl) << 32) | nd.a;
                                heap[hn + 1] = ((uint64_t)orderable_key(pr) << 32) | nd.b;
                            }
                            hn += 2;
                        }
                    }
                    queue_argmax();
                }
            }
            // the leaves this octet popped since the last round: to the other blocks (one word each, nobody waits for the
            // stores) and to this block's own list of known leaves
            if (j == 0) {
                for (uint32_t i = published; i < nl; i++) {
                    const unsigned long long w = (s_leaf[o][i] & 0xFFFFFFFF00000000ull) | s_leaf_n[o][i];
                    dev_store64(&ctl->leaf[g * 8 + o][i], w);
                    dev_store64(&ctl->info[g * 8 + o][i], (s_leaf[o][i] << 32) | s_leaf_a[o][i]);
                    const uint32_t at = atomicAdd(&s_mw[MW_N_KNOWN], 1u);
                    if (at < kMultiKnown) __hip_atomic_store(&s_known[at], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else failed = true;
                }
            }
            published = nl;
            failed = __shfl((int)failed, 0, 8) != 0;
            const bool has = bi != kNone;
            uint32_t top = has ? (uint32_t)(best >> 32) : 0u, flags = (has ? 1u : 0u) | (failed ? 2u : 0u);
            for (uint32_t d = 32; d >= 8; d >>= 1) {
                top = max(top, (uint32_t)__shfl_xor((int)top, d, 64));
                flags |= (uint32_t)__shfl_xor((int)flags, d, 64);
            }
            if (flags & 2u) {
                if (wl == 0) {
                    atomicOr(&ctl->failed, 1u);
                    lds_store(&s_mw[MW_LOCAL_FAILED], 1u);
                }
                break;
            }
            if (!(flags & 1u)) break;                        // nothing queued in this block any more
            if (lds_load(&s_mw[MW_REMOTE_FAILED])) break;    // another block gave up: so does the query
            const uint32_t stop_x = lds_load(&s_mw[MW_STOP_X]);
            if (stop_x != 0u && top < stop_x) break;         // everything this block could still add lies behind the cut
        }
        if (wl == 0) lds_store(&s_mw[MW_FINISH], 1u);
        stamp(1);
    } else if (wave == 1) {
        // ---- the exchange: lane l watches the lists l and l + 64 of the other blocks -------------------------------
        uint32_t cursor[2] = {0u, 0u};
        const uint32_t n_lists = G * 8;
        for (;;) {
            if (lds_load(&s_mw[MW_FINISH])) break;
            unsigned long long w[2] = {0ull, 0ull};
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
                const uint32_t list = wl + 64 * u;
                if (list < n_lists && (list >> 3) != g && cursor[u] < kLeaves) w[u] = dev_load64(&ctl->leaf[list][cursor[u]]);
            }
            uint32_t remote_failed = 0;
            if (wl == 0) remote_failed = __hip_atomic_load(&ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (uint32_t u = 0; u < 2; u++)
                if (w[u] != 0ull) {
                    cursor[u]++;
                    const uint32_t at = atomicAdd(&s_mw[MW_N_KNOWN], 1u);
                    if (at < kMultiKnown) __hip_atomic_store(&s_known[at], w[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else remote_failed = 1;  // (more leaves than the list holds: the long way)
                }
            if (__any(remote_failed != 0) && wl == 0) {
                lds_store(&s_mw[MW_REMOTE_FAILED], 1u);
                if (remote_failed) atomicOr(&ctl->failed, 1u);
            }
            // x = the largest key at which the known leaves hold search_k ids (0: not yet).  Bit by bit from the top — "do the
            // leaves with key >= x | bit still hold search_k ids?" — on the lane's share of the list in registers: 32 sums of
            // n / 64 terms.  (The first version compared every leaf with every other, n^2 / 64 LDS reads per lane: at a few
            // hundred known leaves one pass took tens of microseconds, the descent popped on meanwhile, the list grew, the pass
            // got slower — until a queue's 32-leaf list overflowed and a query that opens 170 leaves took the long way every
            // other call.)
            // (Up to 128 leaves — the usual two or three dozen — the all-pairs pass is the quicker one: half a microsecond.)
            const uint32_t n_known = min(lds_load(&s_mw[MW_N_KNOWN]), kMultiKnown);
            uint32_t x = 0;
            if (n_known <= 128u) {
                for (uint32_t i = wl; i < n_known; i += 64) {
                    const unsigned long long wi = lds_load64(&s_known[i]);
                    const uint32_t ki = (uint32_t)(wi >> 32);
                    if (wi == 0ull || ki <= x) continue;
                    uint32_t sum = 0;
                    for (uint32_t e = 0; e < n_known; e++) {
                        const unsigned long long we = lds_load64(&s_known[e]);
                        sum += (uint32_t)(we >> 32) >= ki ? (uint32_t)we : 0u;
                    }
                    if (sum >= sp.search_k) x = ki;
                }
                for (uint32_t d = 32; d > 0; d >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, d, 64));
            } else {
                constexpr uint32_t kMineKnown = kMultiKnown / 64;
                uint32_t k_key[kMineKnown], k_cnt[kMineKnown];
#pragma unroll
                for (uint32_t r = 0; r < kMineKnown; r++) {
                    const uint32_t i = wl + 64u * r;
                    const unsigned long long wi = i < n_known ? lds_load64(&s_known[i]) : 0ull;
                    k_key[r] = (uint32_t)(wi >> 32);
                    k_cnt[r] = (uint32_t)wi;  // (0 for a slot whose word has not landed yet: it counts for nothing)
                }
                for (uint32_t bit = 0x80000000u; bit != 0u; bit >>= 1) {
                    const uint32_t t = x | bit;
                    uint32_t sum = 0;
#pragma unroll
                    for (uint32_t r = 0; r < kMineKnown; r++) sum += k_key[r] >= t ? k_cnt[r] : 0u;
                    for (uint32_t d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
                    if (sum >= sp.search_k) x = t;
                }
            }
            if (wl == 0 && x > lds_load(&s_mw[MW_STOP_X])) lds_store(&s_mw[MW_STOP_X], x);
        }
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence();  // this block's slots before its count
        s_mw[MW_LAST] = atomicAdd(&ctl->done, 1u) + 1u == G ? 1u : 0u;
    }
    stamp(2);
    __syncthreads();
    if (!s_mw[MW_LAST]) return;
    __threadfence();
    // ---- the last block of the query: every list, wiped as it is read --------------------------------------------
    {
        const uint32_t n_slots = G * 8 * kLeaves;
        for (uint32_t s0 = tid; s0 < n_slots; s0 += 4 * kThreads) {
            unsigned long long w[4], inf[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t sl = s0 + u * kThreads;
                w[u] = sl < n_slots ? dev_load64(&ctl->leaf[0][0] + sl) : 0ull;
                inf[u] = sl < n_slots ? dev_load64(&ctl->info[0][0] + sl) : 0ull;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t sl = s0 + u * kThreads;
                if (w[u] == 0ull) continue;
                dev_store64(&ctl->leaf[0][0] + sl, 0ull);
                dev_store64(&ctl->info[0][0] + sl, 0ull);
                const uint32_t at = atomicAdd(&s_mw[MW_N_MERGED], 1u);
                if (at < kCap) {
                    const uint32_t list = sl / kLeaves, i = sl % kLeaves;
                    s_sorted[at] = (w[u] & 0xFFFFFFFF00000000ull) | (list << 16) | (0xFFFFu - i);
                    s_sorted_n[at] = (uint32_t)w[u];
                    s_sorted_node[at] = (uint32_t)(inf[u] >> 32);
                    s_sorted_first[at] = (uint32_t)inf[u];
                }
            }
        }
    }
    __syncthreads();
    stamp(3);
    const uint32_t n_merged = s_mw[MW_N_MERGED];
    const bool any_failed = n_merged > kCap || __hip_atomic_load(&ctl->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    __syncthreads();
    if (tid == 0) {  // (every block has left: the control words are this block's to clear)
        __hip_atomic_store(&ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->failed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (any_failed) {
        if (tid == 0) {
            nns_count[q] = 0;
            overflow[q] = 1;
            if (single.units) single.n_units[q] = 0;  // (the tile launch behind this kernel runs whatever happened here)
            if (sink.err) atomicOr(sink.err, 16u);  // nobody behind this kernel: the submission takes the long way
        }
        return;
    }
    const uint32_t n_settled = n_merged;
    uint32_t taken = 0, ids_taken = 0, any_tie = 0;
    if (n_settled <= 64u) {
        // The usual case (a query opens two or three dozen leaves): ordered, summed and cut by ONE wave — a leaf per lane, ranks
        // by counting, the prefix by shuffles — and one block barrier, where the general path below takes a dozen.
        if (wave == 0) {
            const bool in = wl < n_settled;
            const uint64_t key = in ? s_sorted[wl] : 0ull;
            const uint32_t kn = in ? s_sorted_n[wl] : 0u, knode = in ? s_sorted_node[wl] : 0u, kfirst = in ? s_sorted_first[wl] : 0u;
            uint32_t rank = 0;
            for (uint32_t t = 0; t < n_settled; t++) rank += s_sorted[t] > key ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (in) {
                s_sorted[rank] = key;
                s_sorted_n[rank] = kn;
                s_sorted_node[rank] = knode;
                s_sorted_first[rank] = kfirst;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t nn = in ? s_sorted_n[wl] : 0u;  // (in sorted order now)
            uint32_t incl = nn;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if (wl >= d) incl += v;
            }
            const uint32_t before = incl - nn;
            const bool take = in && before < sp.search_k;
            if (take) s_pos[wl] = before;
            bool tie = false;
            if (take && before + nn >= sp.search_k) {  // the leaf that reaches search_k: equal keys of another list around it -> sequential queue
                const uint64_t mine = s_sorted[wl];
                const uint32_t kw = (uint32_t)(mine >> 32), lst = (uint32_t)mine >> 16;
                for (uint32_t t = wl + 1; t < n_settled && (uint32_t)(s_sorted[t] >> 32) == kw; t++)
                    if (((uint32_t)s_sorted[t] >> 16) != lst) tie = true;
                for (uint32_t t = wl; t-- > 0 && (uint32_t)(s_sorted[t] >> 32) == kw;)
                    if (((uint32_t)s_sorted[t] >> 16) != lst) tie = true;
            }
            uint32_t it = take ? before + nn : 0u;
            for (uint32_t d = 32; d > 0; d >>= 1) it = max(it, (uint32_t)__shfl_xor((int)it, d, 64));
            const uint32_t n_take = (uint32_t)__popcll(__ballot(take));
            const bool tie_any = __ballot(tie) != 0ull;
            if (wl == 0) {
                s_mw[8] = n_take;
                s_mw[9] = it;
                s_mw[10] = tie_any ? 1u : 0u;
            }
        }
        __syncthreads();
        taken = s_mw[8];
        ids_taken = s_mw[9];
        any_tie = s_mw[10];
    } else {
    uint32_t p2 = kThreads;
    while (p2 < n_settled) p2 <<= 1;
    // descending by key: a leaf's place is the number of larger keys (unique: list and pop index are their low word)
    {
        constexpr uint32_t kMine = kCap / kThreads;
        uint64_t my_key[kMine];
        uint32_t my_n[kMine], my_node[kMine], my_first[kMine], my_rank[kMine];
#pragma unroll
        for (uint32_t r = 0; r < kMine; r++) {
            const uint32_t e = tid + r * kThreads;
            my_rank[r] = 0;
            if (e < n_settled) {
                my_key[r] = s_sorted[e];
                my_n[r] = s_sorted_n[e];
                my_node[r] = s_sorted_node[e];
                my_first[r] = s_sorted_first[e];
                for (uint32_t t = 0; t < n_settled; t++) my_rank[r] += s_sorted[t] > my_key[r] ? 1u : 0u;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < kMine; r++)
            if (tid + r * kThreads < n_settled) {
                s_sorted[my_rank[r]] = my_key[r];
                s_sorted_n[my_rank[r]] = my_n[r];
                s_sorted_node[my_rank[r]] = my_node[r];
                s_sorted_first[my_rank[r]] = my_first[r];
            }
        for (uint32_t t = n_settled + tid; t < p2; t += kThreads) {
            s_sorted[t] = 0;
            s_sorted_n[t] = 0;
            s_sorted_node[t] = 0;
            s_sorted_first[t] = 0;
        }
    }
    __syncthreads();
    // `if nns.len() >= search_k { break }` before every pop: leaf i is taken iff the leaves before it hold < search_k ids
    const uint32_t per = p2 / kThreads;
    uint32_t local = 0;
    for (uint32_t i = 0; i < per; i++) local += s_sorted_n[tid * per + i];
    uint32_t incl = local;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (wl >= d) incl += v;
    }
    if (wl == 63) s_red[32 + wave] = incl;
    __syncthreads();
    uint32_t before = incl - local;
    for (uint32_t w = 0; w < wave; w++) before += s_red[32 + w];
    bool tie = false;
    for (uint32_t i = 0; i < per; i++) {
        const uint32_t e = tid * per + i;
        if (e < n_settled && before < sp.search_k) {
            s_pos[e] = before;
            taken++;
            ids_taken = before + s_sorted_n[e];
            if (ids_taken >= sp.search_k) {  // the leaf that reaches search_k: equal keys of another list around it -> sequential queue
                const uint32_t kw = (uint32_t)(s_sorted[e] >> 32), lst = (uint32_t)s_sorted[e] >> 16;
                for (uint32_t t = e + 1; t < n_settled && (uint32_t)(s_sorted[t] >> 32) == kw; t++)
                    if (((uint32_t)s_sorted[t] >> 16) != lst) tie = true;
                for (uint32_t t = e; t-- > 0 && (uint32_t)(s_sorted[t] >> 32) == kw;)
                    if (((uint32_t)s_sorted[t] >> 16) != lst) tie = true;
            }
        }
        before += s_sorted_n[e];
    }
    taken = block_sum(taken);
    ids_taken = block_max(ids_taken);
    any_tie = block_max(tie ? 1u : 0u);
    __syncthreads();
    }
    if (any_tie || ids_taken > sp.nns_stride) {
        if (tid == 0) {
            nns_count[q] = 0;
            overflow[q] = 1;
            if (single.units) single.n_units[q] = 0;
            if (sink.err) atomicOr(sink.err, 16u);
        }
        return;
    }
    stamp(4);
    if (!sp.filter_bits) {
        // the taken leaves' ids are one flat range [0, ids_taken) of the query's candidate buffer (k_descend_block)
        const bool ids_by_tiles = single.units && single.ids_by_tiles;
        TileUnit *const my_units = single.units ? single.units + (size_t)q * single.per_query : nullptr;
        Visit *const my_sorted = single.units ? single.sorted + (size_t)q * single.per_query : nullptr;
        for (uint32_t e = tid; e < taken; e += kThreads) {
            if (single.units) {
                my_units[e] = TileUnit{s_sorted_node[e], e, 1u, ids_by_tiles ? s_sorted_first[e] + 1u : 0u};
                my_sorted[e] = Visit{s_sorted_node[e], q, s_pos[e], s_sorted_n[e]};
            } else {
                record_visit(sink, s_sorted_node[e], q, s_pos[e], s_sorted_n[e]);
            }
        }
        constexpr uint32_t kFly = 20;
        uint32_t e = 0;
        {
            uint32_t hi = taken;  // the last e with s_pos[e] <= tid
            while (hi - e > 1) {
                const uint32_t mid = (e + hi) >> 1;
                if (s_pos[mid] <= tid) e = mid;
                else hi = mid;
            }
        }
        uint32_t next_first = e + 1 < taken ? s_pos[e + 1] : 0xFFFFFFFFu, base = taken ? s_sorted_first[e] - s_pos[e] : 0u;
        for (uint32_t p0 = tid; p0 < (ids_by_tiles ? 0u : ids_taken); p0 += kFly * kThreads) {
            uint32_t id[kFly];
#pragma unroll
            for (uint32_t u = 0; u < kFly; u++) {
                const uint32_t p = p0 + u * kThreads;
                while (p >= next_first) {
                    e++;
                    next_first = e + 1 < taken ? s_pos[e + 1] : 0xFFFFFFFFu;
                    base = s_sorted_first[e] - s_pos[e];
                }
                id[u] = p < ids_taken ? sp.desc[base + p] : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t u = 0; u < kFly; u++)
                if (p0 + u * kThreads < ids_taken) my_nns[p0 + u * kThreads] = id[u];
        }
    } else {
        for (uint32_t e = tid >> 3; e < taken; e += kThreads / 8) {
            const uint32_t node = s_sorted_node[e], pos = s_pos[e];
            const DNode nd = sp.nodes[node];
            copy_filtered(sp, sp.desc + nd.a, nd.b, my_nns + pos, j);
            if (j == 0) {
                if (single.units) {
                    single.units[(size_t)q * single.per_query + e] = TileUnit{node, e, 1u, 0u};
                    single.sorted[(size_t)q * single.per_query + e] = Visit{node, q, pos, s_sorted_n[e]};
                } else {
                    record_visit(sink, node, q, pos, s_sorted_n[e]);
                }
            }
        }
    }
    stamp(5);
    if (tid == 0) {
        if (trace_on) {
            ctl->trace[g][6] = n_merged;
            ctl->trace[g][7] = taken;
        }
        nns_count[q] = ids_taken;
        overflow[q] = 0;
        if (sp.stats) {
            atomicAdd(&sp.stats[SS_BLOCK], 1u);
            atomicAdd(&sp.stats[SS_MULTI], 1u);
        }
        if (single.units) {
            single.n_units[q] = taken;
            if (sink.total && nq == 1u) *sink.total = taken;
            if (sp.stats && taken) {
                atomicAdd(&sp.stats[SS_VISITS], taken);
                atomicAdd(&sp.stats[SS_UNITS_4], taken);
            }
        }
    }
}

// nns.sort_unstable(); nns.dedup()  (src/reader.rs:378-379): one block per query, LDS bitonic sort.
__global__ __launch_bounds__(256) void k_sort_dedup_lds(uint32_t *__restrict__ nns, uint32_t stride,
                                                        uint32_t *__restrict__ counts) {
    extern __shared__ uint32_t s_ids[];
    __shared__ uint32_t s_scan[256];
    const uint32_t q = blockIdx.x;
    const uint32_t n = counts[q];
    uint32_t *ids = nns + (uint64_t)q * stride;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    if (np2 < 2) np2 = 2;
    for (uint32_t t = threadIdx.x; t < np2; t += blockDim.x) s_ids[t] = t < n ? ids[t] : 0xFFFFFFFFu;
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t str = size >> 1; str > 0; str >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (np2 >> 1); t += blockDim.x) {
                const uint32_t lo = 2 * t - (t & (str - 1)), hi = lo + str;
                const bool up = (lo & size) == 0;
                const uint32_t a = s_ids[lo], b = s_ids[hi];
                if ((a > b) == up) {
                    s_ids[lo] = b;
                    s_ids[hi] = a;
                }
            }
        }
    }
    __syncthreads();
    // dedup: thread t owns the contiguous slice [t*per, (t+1)*per)
    const uint32_t per = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += (i == 0 || s_ids[i] != s_ids[i - 1]) ? 1u : 0u;
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < blockDim.x; t++) {
            const uint32_t v = s_scan[t];
            s_scan[t] = run;
            run += v;
        }
        counts[q] = run;
    }
    __syncthreads();
    uint32_t w = s_scan[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++)
        if (i == 0 || s_ids[i] != s_ids[i - 1]) ids[w++] = s_ids[i];
}

// The same sort + dedup when the id space is small enough for one bit per id in LDS: set the bit of every candidate,
// then walk the words in order -- the ids come out ascending and unique, O(n + ids/32) instead of O(n log^2 n).
// 16 waves per block (one block owns the LDS of a CU): wave v walks the words [v, v+1) * n_words/16, 64 consecutive
// words per step, a wave prefix sum of the popcounts gives every lane its place, so one store instruction of the wave
// lands in a few adjacent lines of the output.
// An id beyond `id_limit` cannot be an item of the dataset: bit 0 of *err, the same error the re-rank would raise.
static constexpr uint32_t kBitmapMaxWords = 39 * 1024;  // 156 KiB of the 160 KiB of LDS: 1 277 952 ids
__global__ __launch_bounds__(1024) void k_dedup_bitmap_lds(uint32_t *__restrict__ nns, uint32_t stride,
                                                           uint32_t *__restrict__ counts, uint32_t n_words,
                                                           uint32_t id_limit, uint32_t *__restrict__ err) {
    extern __shared__ uint32_t s_bits[];  // n_words, a multiple of 1024
    __shared__ uint32_t s_wave[16];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = counts[q];
    uint32_t *ids = nns + (uint64_t)q * stride;
    for (uint32_t t = tid; t < n_words / 4; t += 1024) reinterpret_cast<uint4 *>(s_bits)[t] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    bool bad = false;
    for (uint32_t t0 = tid; t0 < n; t0 += 4 * 1024) {
        uint32_t id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) id[u] = t0 + u * 1024 < n ? ids[t0 + u * 1024] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t0 + u * 1024 >= n) continue;
            if (id[u] >= id_limit)
                bad = true;
            else
                atomicOr(&s_bits[id[u] >> 5], 1u << (id[u] & 31));
        }
    }
    if (bad) atomicOr(err, 1u);
    __syncthreads();
    const uint32_t per_wave = n_words / 16, first = wave * per_wave;
    uint32_t mine = 0;
    for (uint32_t i = lane; i < per_wave; i += 64) mine += __popc(s_bits[first + i]);
    for (uint32_t d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t v = 0; v < 16; v++) {
        const uint32_t c = s_wave[v];
        base += v < wave ? c : 0u;
        total += c;
    }
    if (tid == 0) counts[q] = total;
    for (uint32_t i = 0; i < per_wave; i += 64) {
        uint32_t bits = s_bits[first + i + lane];
        const uint32_t c = __popc(bits);
        uint32_t incl = c;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        uint32_t w = base + incl - c;
        const uint32_t id0 = (first + i + lane) << 5;
        while (bits) {
            ids[w++] = id0 + (uint32_t)__ffs(bits) - 1u;
            bits &= bits - 1;
        }
        base += __shfl(incl, 63, 64);
    }
}

// ---- leaf-tile re-rank of ah_search_batch -----------------------------------------------------------------------
// The candidates of a query are whole leaves, and queries of one submission meet in the same leaves.  Instead of
// sorting every query's list and inverting the (query, candidate) pairs by row, the descent records its leaf visits
// (VisitSink); the visits are counting-sorted by node and cut into units of <= 16 visits of ONE node, and a block takes
// a slab of the leaf's rows against the unit's queries: an octet holds R rows x Q queries of accumulators (a step of 32
// dimensions issues R + Q loads for R*Q pairs; the row-run kernel of batch.hip: 5 loads for 4 pairs) and the octets of
// a wave that work on the same rows share one load instruction, so a row line is fetched once for up to 16 queries
// (measured, 1000 queries x 11k candidates over 1M x 1536: 9.5 GB of HBM reads per submission; one block per 4 queries
// of a leaf read 15.9 GB -- concurrent misses of one line in L2 are not merged).  Every pair is still reduced by one
// octet in the reference's order: the distances are bit-identical to the other re-rank kernels.
// Distances land at the pair's position in the query's UNSORTED candidate list; duplicates (an item met in leaves of
// several trees) are flagged by k_flag_duplicates, and k_search_select orders by (OrderedFloat(distance), id) -- the
// order of the reference's (distance, position-in-the-sorted-list) keys.  What that cannot reproduce (a non-finite
// distance: reader.rs:611-621 looks at positions; a selection that does not fit the small sort) raises a bit of *err
// and the submission is redone by the sort + row-major path.

// Two exclusive scans over the per-node visit counters in one pass (3 launches): `cursor` = first slot of the node's
// visits in the sorted list, `ustart` = first work unit of the node (a unit = <= 16 visits of one node).  The last
// launch also writes the units.
static constexpr uint32_t kLeafScanItems = 2048;  // 256 threads x 8
__device__ __forceinline__ uint2 block_exclusive_scan2(uint2 local, uint2 *s_wave, uint2 &block_total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint2 incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ux = __shfl_up(incl.x, d), uy = __shfl_up(incl.y, d);
        if ((int)lane >= d) {
            incl.x += ux;
            incl.y += uy;
        }
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint2 before = make_uint2(incl.x - local.x, incl.y - local.y);
    block_total = make_uint2(0, 0);
    for (uint32_t w = 0; w < 4; w++) {
        if (w < wave) {
            before.x += s_wave[w].x;
            before.y += s_wave[w].y;
        }
        block_total.x += s_wave[w].x;
        block_total.y += s_wave[w].y;
    }
    __syncthreads();
    return before;
}
__global__ __launch_bounds__(256) void k_leaf_scan_block(const uint32_t *__restrict__ count, uint32_t n,
                                                         uint32_t *__restrict__ cursor, uint32_t *__restrict__ ustart,
                                                         uint2 *__restrict__ sums) {
    __shared__ uint2 s_wave[4];
    const uint32_t base = blockIdx.x * kLeafScanItems + threadIdx.x * 8;
    uint32_t v[8];
    uint2 local = make_uint2(0, 0);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        v[e] = base + e < n ? count[base + e] : 0u;
        local.x += v[e];
        local.y += (v[e] + kUnitVisits - 1) / kUnitVisits;
    }
    uint2 total;
    uint2 before = block_exclusive_scan2(local, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (base + e < n) {
            cursor[base + e] = before.x;
            ustart[base + e] = before.y;
        }
        before.x += v[e];
        before.y += (v[e] + kUnitVisits - 1) / kUnitVisits;
    }
}
__global__ __launch_bounds__(256) void k_leaf_scan_sums(uint2 *__restrict__ sums, uint32_t n_sums, uint32_t *__restrict__ n_units) {
    __shared__ uint2 s_wave[4];
    __shared__ uint2 s_carry;
    if (threadIdx.x == 0) s_carry = make_uint2(0, 0);
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_sums; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint2 v = i < n_sums ? sums[i] : make_uint2(0, 0);
        uint2 total;
        const uint2 before = block_exclusive_scan2(v, s_wave, total);
        const uint2 carry = s_carry;
        if (i < n_sums) sums[i] = make_uint2(carry.x + before.x, carry.y + before.y);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = make_uint2(carry.x + total.x, carry.y + total.y);
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_units = s_carry.y;
}
__global__ __launch_bounds__(256) void k_leaf_scan_add(const uint32_t *__restrict__ count, uint32_t n,
                                                       uint32_t *__restrict__ cursor, const uint32_t *__restrict__ ustart,
                                                       const uint2 *__restrict__ sums, TileUnit *__restrict__ units,
                                                       uint32_t *__restrict__ stats) {
    const uint2 add = sums[blockIdx.x];
    const uint32_t base = blockIdx.x * kLeafScanItems;
    uint32_t u16 = 0, u8 = 0, u4 = 0, visits = 0;  // units by the tile variant that will serve them (k_leaf_tiles)
    for (uint32_t e = threadIdx.x; e < kLeafScanItems; e += 256) {
        const uint32_t node = base + e;
        if (node >= n) break;
        const uint32_t c = count[node], first = cursor[node] + add.x;
        cursor[node] = first;
        TileUnit *dst = units + ustart[node] + add.y;
        visits += c;
        for (uint32_t i = 0; i * kUnitVisits < c; i++) {
            const uint32_t nv = min(kUnitVisits, c - i * kUnitVisits);
            dst[i] = TileUnit{node, first + i * kUnitVisits, nv, 0u};
            u16 += nv > 8 ? 1u : 0u;
            u8 += nv > 4 && nv <= 8 ? 1u : 0u;
            u4 += nv <= 4 ? 1u : 0u;
        }
    }
    if (stats) {
        if (u16) atomicAdd(&stats[SS_UNITS_16], u16);
        if (u8) atomicAdd(&stats[SS_UNITS_8], u8);
        if (u4) atomicAdd(&stats[SS_UNITS_4], u4);
        if (visits) atomicAdd(&stats[SS_VISITS], visits);
    }
}
// visits -> their node's run of the sorted list
__global__ __launch_bounds__(256) void k_visit_scatter(const Visit *__restrict__ visits, const uint32_t *__restrict__ total,
                                                       uint32_t cap, uint32_t *__restrict__ cursor, Visit *__restrict__ sorted) {
    const uint32_t n = min(*total, cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Visit v = visits[i];
        sorted[atomicAdd(&cursor[v.node], 1u)] = v;
    }
}

// The same for a SMALL submission (arroy's own API is one query per call, src/reader.rs:46-75): the three scans above walk a
// counter per node of the index (and a fourth launch zeroes them) to place what for one query is a dozen visits.  One block
// sorts the visits by node in LDS instead — runs of one node are the node's visits, every 16 of a run a unit.  More than
// kSmallVisits visits: bit 5 of *err, the submission takes the long way like any other overflow of the visit list.
static constexpr uint32_t kSmallVisits = 2048;
__global__ __launch_bounds__(256) void k_units_small(const Visit *__restrict__ visits, const uint32_t *__restrict__ total,
                                                     uint32_t cap, Visit *__restrict__ sorted, TileUnit *__restrict__ units,
                                                     uint32_t *__restrict__ n_units, uint32_t *__restrict__ stats, QueriesH16 h16,
                                                     uint32_t *__restrict__ items = nullptr, uint32_t items_cap = 0) {
    // items (round 6, a few queries a call): [count][unit << 16 | slab] — the (unit, slab of its leaf) pairs k_leaf_tiles16<true>
    // has work for, so that its launch is one block per pair instead of a 2-D grid of 32 nq x (largest leaf's slabs) blocks of
    // which a fifth find work (items[0] = 0xFFFFFFFF: more pairs than the list holds, the launch walks its grid as before)
    __shared__ uint64_t s_key[kSmallVisits];  // node << 32 | index into visits
    __shared__ uint32_t s_wave[4], s_wave_items[4];
    if (blockIdx.x != 0) {
        if (threadIdx.x < 64) query_h16(blockIdx.x - 1, threadIdx.x, h16.qvecs, h16.qstride, h16.dims, h16.hpitch, h16.q16, h16.qstats);
        return;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = *total;
    if (n > cap || n > kSmallVisits) {  // block-uniform
        if (tid == 0) {
            atomicOr(&stats[SS_ERR], 32u);
            *n_units = 0;
            if (items) items[0] = 0u;
        }
        return;
    }
    uint32_t p2 = 2;
    while (p2 < n) p2 <<= 1;
    for (uint32_t i = tid; i < p2; i += 256) s_key[i] = i < n ? ((uint64_t)visits[i].node << 32) | i : ~0ull;
    for (uint32_t size = 2; size <= p2; size <<= 1)
        for (uint32_t str = size >> 1; str > 0; str >>= 1) {
            __syncthreads();
            for (uint32_t t = tid; t < (p2 >> 1); t += 256) {
                const uint32_t a_i = 2 * t - (t & (str - 1)), b_i = a_i + str;
                const bool up = (a_i & size) == 0;  // ascending
                const uint64_t x = s_key[a_i], y = s_key[b_i];
                if ((x > y) == up) {
                    s_key[a_i] = y;
                    s_key[b_i] = x;
                }
            }
        }
    __syncthreads();
    auto lower_bound = [&](uint64_t key) {  // first position of the sorted keys that is >= key
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_key[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    // thread t owns the positions [t per, (t + 1) per): the units come out in node order like the scans' would
    const uint32_t per = (n + 255u) / 256u;
    uint32_t mine = 0;
    for (uint32_t p = tid * per; p < min(n, (tid + 1) * per); p++) {
        const uint64_t key = s_key[p];
        sorted[p] = visits[(uint32_t)key];
        mine += ((p - lower_bound(key & 0xFFFFFFFF00000000ull)) % kUnitVisits) == 0 ? 1u : 0u;
    }
    uint32_t incl = mine;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t u = incl - mine, all = 0;
    for (uint32_t w = 0; w < 4; w++) {
        u += w < wave ? s_wave[w] : 0u;
        all += s_wave[w];
    }
    uint32_t u16 = 0, u8 = 0, u4 = 0;
    constexpr uint32_t kMine = kSmallVisits / 256;  // units a thread can own
    uint32_t my_unit[kMine], my_slabs[kMine], n_mine = 0, my_items = 0;
    for (uint32_t p = tid * per; p < min(n, (tid + 1) * per); p++) {
        const uint64_t key = s_key[p], node_key = key & 0xFFFFFFFF00000000ull;
        if (((p - lower_bound(node_key)) % kUnitVisits) != 0) continue;
        const uint32_t nv = min(kUnitVisits, lower_bound(node_key + (1ull << 32)) - p);
        const uint32_t slab = tile_slab_rows(nv, true), n_leaf = visits[(uint32_t)key].n;
        my_unit[n_mine] = u;
        my_slabs[n_mine] = (n_leaf + slab - 1u) / slab;
        my_items += my_slabs[n_mine];
        n_mine++;
        units[u++] = TileUnit{(uint32_t)(key >> 32), p, nv, 0u};
        u16 += nv > 8 ? 1u : 0u;
        u8 += nv > 4 && nv <= 8 ? 1u : 0u;
        u4 += nv <= 4 ? 1u : 0u;
    }
    if (items) {  // block-uniform
        uint32_t incl_i = my_items;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl_i, d, 64);
            if (lane >= d) incl_i += v;
        }
        if (lane == 63) s_wave_items[wave] = incl_i;
        __syncthreads();
        uint32_t at = incl_i - my_items, all_items = 0;
        for (uint32_t w = 0; w < 4; w++) {
            at += w < wave ? s_wave_items[w] : 0u;
            all_items += s_wave_items[w];
        }
        const bool fits = all_items < items_cap && all < 65536u;
        if (fits)
            for (uint32_t i = 0; i < n_mine; i++)
                for (uint32_t sl = 0; sl < my_slabs[i]; sl++) items[1u + at++] = (my_unit[i] << 16) | min(sl, 0xFFFFu);
        if (tid == 0) items[0] = fits ? all_items : 0xFFFFFFFFu;
    }
    if (u16) atomicAdd(&stats[SS_UNITS_16], u16);
    if (u8) atomicAdd(&stats[SS_UNITS_8], u8);
    if (u4) atomicAdd(&stats[SS_UNITS_4], u4);
    if (tid == 0) {
        if (n) atomicAdd(&stats[SS_VISITS], n);
        *n_units = all;
    }
}

// nns.dedup() without the sort: one bit per id in LDS, the second and later occurrences of an id become 0xFFFFFFFF
// (never an item id here: the id space fits the bitmap).  unique[q] = ids left.
__global__ __launch_bounds__(1024) void k_flag_duplicates(uint32_t *__restrict__ nns, uint32_t stride,
                                                          const uint32_t *__restrict__ counts, uint32_t n_words,
                                                          uint32_t id_limit, uint32_t *__restrict__ unique,
                                                          uint32_t *__restrict__ err) {
    extern __shared__ uint32_t s_bits[];  // n_words, a multiple of 1024
    __shared__ uint32_t s_unique;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = counts[q];
    uint32_t *ids = nns + (uint64_t)q * stride;
    for (uint32_t t = tid; t < n_words / 4; t += 1024) reinterpret_cast<uint4 *>(s_bits)[t] = make_uint4(0, 0, 0, 0);
    if (tid == 0) s_unique = 0;
    __syncthreads();
    uint32_t mine = 0;
    bool bad = false;
    for (uint32_t t0 = tid; t0 < n; t0 += 4 * 1024) {
        uint32_t id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) id[u] = t0 + u * 1024 < n ? ids[t0 + u * 1024] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t0 + u * 1024 >= n) continue;
            if (id[u] >= id_limit) {
                bad = true;
                ids[t0 + u * 1024] = 0xFFFFFFFFu;
                continue;
            }
            const uint32_t bit = 1u << (id[u] & 31);
            if (atomicOr(&s_bits[id[u] >> 5], bit) & bit) ids[t0 + u * 1024] = 0xFFFFFFFFu;
            else mine++;
        }
    }
    if (bad) atomicOr(err, 1u);
    for (uint32_t d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((tid & 63u) == 0 && mine) atomicAdd(&s_unique, mine);
    __syncthreads();
    if (tid == 0) unique[q] = s_unique;
}

// One block (4 waves) on rows [row_begin, row_end) of a leaf against the <= 16 visits of a unit.  The 8 octets of a wave
// are QO query groups x 8/QO row groups; an octet holds R rows x Q queries of accumulators.  The octets of a wave that
// share a row group issue the same row addresses in the same load instruction, so a row line leaves L2 once per wave for
// up to QO*Q = 16 queries.  This version keeps its operands in registers; it serves the units of <= 4 visits (QO = 1),
// which are bound by the HBM reads of their rows whatever the inner loop does.
// The same for id spaces too big for a bitmap (10M items: 1.25 MB of bits): an open-addressing set of the query's
// candidates in LDS, 32768 slots for <= 24576 candidates (a submission with longer lists takes the sorted path).  The
// first insertion of an id wins its slot; a later one finds it there and is flagged.  0xFFFFFFFF marks an empty slot, so the
// path is not taken when that id is stored.
static constexpr uint32_t kHashSlots = 32768, kHashMaxCandidates = 24576;
__global__ __launch_bounds__(1024) void k_flag_duplicates_hash(uint32_t *__restrict__ nns, uint32_t stride,
                                                               const uint32_t *__restrict__ counts,
                                                               uint32_t *__restrict__ unique, uint32_t *__restrict__ err) {
    extern __shared__ uint32_t s_set[];  // kHashSlots
    __shared__ uint32_t s_unique;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = counts[q];
    uint32_t *ids = nns + (uint64_t)q * stride;
    for (uint32_t t = tid; t < kHashSlots / 4; t += 1024) reinterpret_cast<uint4 *>(s_set)[t] = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (tid == 0) s_unique = 0;
    __syncthreads();
    if (n > kHashMaxCandidates) {  // block-uniform; cannot happen for the strides this path is chosen for
        if (tid == 0) atomicOr(err, 8u);
        return;
    }
    uint32_t mine = 0;
    for (uint32_t t = tid; t < n; t += 1024) {
        const uint32_t id = ids[t];
        uint32_t h = (id * 2654435761u) >> 17;  // Fibonacci hashing: the top 15 bits
        for (;;) {
            const uint32_t old = atomicCAS(&s_set[h], 0xFFFFFFFFu, id);
            if (old == 0xFFFFFFFFu) {
                mine++;
                break;
            }
            if (old == id) {
                ids[t] = 0xFFFFFFFFu;
                break;
            }
            h = (h + 1) & (kHashSlots - 1);
        }
    }
    for (uint32_t d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((tid & 63u) == 0 && mine) atomicAdd(&s_unique, mine);
    __syncthreads();
    if (tid == 0) unique[q] = s_unique;
}

template <int METRIC, int R, int Q, int QO>
__device__ __forceinline__ void leaf_tile(const DataView &dv, const uint32_t *__restrict__ leaf_ids, uint32_t row_begin,
                                          uint32_t n_rows, const Visit *__restrict__ vis, uint32_t n_vis,
                                          const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                          const float *__restrict__ qhdrs, float *__restrict__ dist, uint32_t stride,
                                          uint32_t *err) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    constexpr uint32_t RO = 8 / QO;  // row groups per wave
    const uint32_t j = threadIdx.x & 7u, ow = (threadIdx.x >> 3) & 7u, wave = threadIdx.x >> 6;
    const uint32_t q_oct = ow % QO, row_oct = wave * RO + ow / QO;
    const uint32_t blocks = dv.dims >> 5;
    const float4 *q4[Q];
    uint32_t qi[Q];
    float *out[Q];
#pragma unroll
    for (int t = 0; t < Q; t++) {
        const Visit v = vis[min(q_oct * Q + (uint32_t)t, n_vis - 1)];
        qi[t] = v.q;
        q4[t] = reinterpret_cast<const float4 *>(qvecs + (uint64_t)v.q * qstride) + j;
        out[t] = dist + (uint64_t)v.q * stride + v.pos;
    }
    for (uint32_t r0 = row_begin + row_oct * R; r0 < n_rows; r0 += 4 * RO * R) {
        const float4 *r4[R];
        uint64_t row[R];
#pragma unroll
        for (int u = 0; u < R; u++) {
            row[u] = row_of_id(dv, leaf_ids[min(r0 + u, n_rows - 1)]);
            r4[u] = reinterpret_cast<const float4 *>(dv.rows_f32 + (row[u] == ~0ull ? 0ull : row[u]) * dv.pitch) + j;
        }
        float4 acc[R][Q];
#pragma unroll
        for (int u = 0; u < R; u++)
#pragma unroll
            for (int t = 0; t < Q; t++) acc[u][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < blocks; k++) {
            float4 x[R], y[Q];
#pragma unroll
            for (int u = 0; u < R; u++) x[u] = r4[u][k * 8];
#pragma unroll
            for (int t = 0; t < Q; t++) y[t] = q4[t][k * 8];
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int t = 0; t < Q; t++) fma_step<OP>(acc[u][t], y[t], x[u]);
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            if (r0 + u >= n_rows) continue;
            const float *rp = reinterpret_cast<const float *>(r4[u] - j);
#pragma unroll
            for (int t = 0; t < Q; t++) {
                if (q_oct * Q + (uint32_t)t >= n_vis) continue;
                float r = octet_finish(acc[u][t]);
                r = scalar_tail<OP>(r, reinterpret_cast<const float *>(q4[t] - j), rp, blocks << 5, dv.dims);
                if (j == 0) {
                    float d = r;
                    if (METRIC == AH_COSINE) d = cosine_from_dot(r, qhdrs[2 * (uint64_t)qi[t]], dv.headers[row[u] == ~0ull ? 0 : row[u]]);
                    if (METRIC == AH_DOT_PRODUCT) d = -r;
                    if (row[u] == ~0ull) {
                        atomicOr(err, 1u);
                        d = __uint_as_float(0x7FC00000u);
                    }
                    out[t][r0 + u] = d;
                }
            }
        }
    }
}

// The same tile for units of 5..16 visits, with both operands streamed through a wave-private ring in LDS by the DMA path
// (global_load_lds_dwordx4: lane l of an instruction fetches 16 bytes and they land lane-linearly, 1 KiB per instruction).
// One step of 32 dimensions of a wave = 3 KiB: QO = 4: its 8 rows (1 instruction) + the 16 queries (2); QO = 2: 16 rows
// (2) + 8 queries (1).  DEPTH - 1 steps are in flight while one is consumed (`s_waitcnt vmcnt` counts the
// instructions of a step), so a wave keeps 15 KiB of loads outstanding without holding a register for them, and what
// reaches the registers comes from LDS (ds_read_b128), not through the texture path a second time.  The LDS reads are
// inline assembly: the compiler would otherwise wait for EVERY outstanding DMA before an LDS read it cannot tell apart.
// Measured (1000 queries x 11 k candidates, 1M x 1536): k_leaf_tiles 2.21 -> 1.82 ms; the register version with 8 x 4 or
// 4 x 8 accumulators per octet instead (fewer loads per pair, fewer waves): 2.9 ms -- the loop lives on loads in flight.
// Slots: row w (0..8*4/QO-1) at 128 w; query qo*4+t at 128 (t*QO + qo), so the two octets of a 16-lane group read
// adjacent 128-byte segments (all 64 banks once).
static constexpr uint32_t kRingBytesPerWave = 18 * 1024;
typedef float ring_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ring_f4 lds_read_f4(uint32_t addr) {
    ring_f4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// QO = 4 / 2 query groups per wave: 16 / 8 queries against 8 / 16 rows per step, DEPTH steps in the ring.  Slots: row
// w = ro*4 + u at 128 w; query qo*4 + t at 128 (t*QO + qo) behind the rows -- the two octets of a 16-lane group then read
// either the same 128 bytes or adjacent ones.  (The ring for units of <= 4 visits as well, 32 rows + 4 queries per step:
// 364 k instead of 370 k queries/s, and no better on queries that share nothing -- not kept.)
template <int METRIC, int QO, int DEPTH>
__device__ __forceinline__ void leaf_tile_ring(const DataView &dv, const uint32_t *__restrict__ leaf_ids, uint32_t row_begin,
                                               uint32_t n_rows, const Visit *__restrict__ vis, uint32_t n_vis,
                                               const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                               const float *__restrict__ qhdrs, float *__restrict__ dist, uint32_t stride,
                                               uint32_t *err, uint8_t *ring_all) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    constexpr int R = 4, Q = 4;
    constexpr uint32_t RO = 8 / QO;                        // row groups per wave
    constexpr uint32_t kRowInstr = RO * R / 8;             // DMA instructions for the rows of a step
    constexpr uint32_t kQueryInstr = QO * Q / 8;
    constexpr uint32_t kStepInstr = kRowInstr + kQueryInstr, kStepBytes = kStepInstr * 1024;
    static_assert(DEPTH * kStepBytes <= kRingBytesPerWave, "the ring of a wave");
    const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, ow = lane >> 3, wave = threadIdx.x >> 6;
    const uint32_t q_oct = ow % QO, ro = ow / QO;
    const uint32_t blocks = dv.dims >> 5;
    uint8_t *ring = ring_all + wave * kRingBytesPerWave;
    const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)ring;
    // the queries: what this lane fetches for the ring, and what this octet finishes
    const uint8_t *q_src[kQueryInstr];
#pragma unroll
    for (uint32_t i = 0; i < kQueryInstr; i++) {
        const uint32_t slot = i * 8 + ow, t = slot / QO, qo = slot % QO;
        q_src[i] = qvecs + (uint64_t)vis[min(qo * Q + t, n_vis - 1)].q * qstride + j * 16;
    }
    uint32_t qi[Q];
    float *out[Q];
#pragma unroll
    for (int t = 0; t < Q; t++) {
        const Visit v = vis[min(q_oct * Q + (uint32_t)t, n_vis - 1)];
        qi[t] = v.q;
        out[t] = dist + (uint64_t)v.q * stride + v.pos;
    }
    const uint32_t x_addr = ring_addr + ro * R * 128 + j * 16;
    const uint32_t y_addr = ring_addr + kRowInstr * 1024 + q_oct * 128 + j * 16;
    for (uint32_t base = row_begin + wave * RO * R; base < n_rows; base += 4 * RO * R) {
        const uint8_t *r_src[kRowInstr];
#pragma unroll
        for (uint32_t i = 0; i < kRowInstr; i++) {
            const uint64_t row = row_of_id(dv, leaf_ids[min(base + i * 8 + ow, n_rows - 1)]);
            r_src[i] = reinterpret_cast<const uint8_t *>(dv.rows_f32 + (row == ~0ull ? 0ull : row) * dv.pitch) + j * 16;
        }
        auto issue = [&](uint32_t k) {
            uint8_t *slot = ring + (k % DEPTH) * kStepBytes;
#pragma unroll
            for (uint32_t i = 0; i < kRowInstr; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(r_src[i] + (uint64_t)k * 128),
                                                 (__attribute__((address_space(3))) void *)(slot + i * 1024), 16, 0, 0);
#pragma unroll
            for (uint32_t i = 0; i < kQueryInstr; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(q_src[i] + (uint64_t)k * 128),
                                                 (__attribute__((address_space(3))) void *)(slot + (kRowInstr + i) * 1024), 16, 0, 0);
        };
        for (uint32_t k = 0; k + 1 < DEPTH && k < blocks; k++) issue(k);
        float4 acc[R][Q];
#pragma unroll
        for (int u = 0; u < R; u++)
#pragma unroll
            for (int t = 0; t < Q; t++) acc[u][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < blocks; k++) {
            if (k + DEPTH - 1 < blocks) {
                issue(k + DEPTH - 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStepInstr * (DEPTH - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const uint32_t so = (k % DEPTH) * kStepBytes;
            ring_f4 x[R], y[Q];
#pragma unroll
            for (int u = 0; u < R; u++) x[u] = lds_read_f4(x_addr + so + u * 128);
#pragma unroll
            for (int t = 0; t < Q; t++) y[t] = lds_read_f4(y_addr + so + t * QO * 128);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int t = 0; t < Q; t++)
                    fma_step<OP>(acc[u][t], make_float4(y[t].x, y[t].y, y[t].z, y[t].w), make_float4(x[u].x, x[u].y, x[u].z, x[u].w));
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t r = base + ro * R + u;
            if (r >= n_rows) continue;
            const uint64_t row = row_of_id(dv, leaf_ids[r]);
            const float *rp = dv.rows_f32 + (row == ~0ull ? 0ull : row) * dv.pitch;
#pragma unroll
            for (int t = 0; t < Q; t++) {
                if (q_oct * Q + (uint32_t)t >= n_vis) continue;
                float red = octet_finish(acc[u][t]);
                red = scalar_tail<OP>(red, reinterpret_cast<const float *>(qvecs + (uint64_t)qi[t] * qstride), rp, blocks << 5, dv.dims);
                if (j == 0) {
                    float d = red;
                    if (METRIC == AH_COSINE) d = cosine_from_dot(red, qhdrs[2 * (uint64_t)qi[t]], dv.headers[row == ~0ull ? 0 : row]);
                    if (METRIC == AH_DOT_PRODUCT) d = -red;
                    if (row == ~0ull) {
                        atomicOr(err, 1u);
                        d = __uint_as_float(0x7FC00000u);
                    }
                    out[t][r] = d;
                }
            }
        }
    }
}

// blockIdx.x walks the units (persistent), blockIdx.y is the slab of rows of the unit's leaf: 128 rows when the unit has
// more than 8 visits (4 rounds of 32 rows x 16 queries), 256 rows otherwise.
#ifndef AH_TILES16_KF
#define AH_TILES16_KF 1  // k-steps of a leaf tile requested together in the big submissions' variants (A/B: see DESIGN.md)
#endif
template <int METRIC>
__global__ __launch_bounds__(256) void k_leaf_tiles(DataView dv, const uint32_t *__restrict__ nns,
                                                    const Visit *__restrict__ sorted,
                                                    const TileUnit *__restrict__ units, const uint32_t *__restrict__ n_units_p,
                                                    const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                    const float *__restrict__ qhdrs, float *__restrict__ dist, uint32_t stride,
                                                    uint32_t *err) {
    extern __shared__ uint4 s_ring4[];
    uint8_t *ring = reinterpret_cast<uint8_t *>(s_ring4);
    const uint32_t n_units = *n_units_p;
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const TileUnit unit = units[u];
        const Visit *vis = sorted + unit.first;
        // the leaf's ids (those the candidate filter kept) as the descent wrote them for the unit's first visit
        const uint32_t n_leaf = vis[0].n;
        const uint32_t n_vis = unit.n_vis, slab = n_vis > 8 ? kTileSlab : 2 * kTileSlab;
        const uint32_t row_begin = blockIdx.y * slab;
        if (row_begin >= n_leaf) continue;
        const uint32_t row_end = min(n_leaf, row_begin + slab);
        const uint32_t *leaf_ids = nns + (uint64_t)vis[0].q * stride + vis[0].pos;
#define AH_TILE(R, Q, QO) \
    leaf_tile<METRIC, R, Q, QO>(dv, leaf_ids, row_begin, row_end, vis, n_vis, qvecs, qstride, qhdrs, dist, stride, err)
#define AH_RING(QO, DEPTH) \
    leaf_tile_ring<METRIC, QO, DEPTH>(dv, leaf_ids, row_begin, row_end, vis, n_vis, qvecs, qstride, qhdrs, dist, stride, err, ring)
        if (n_vis > 8) AH_RING(4, 6);
        else if (n_vis > 4) AH_RING(2, 6);
        else if (n_vis > 2) AH_TILE(4, 4, 1);
        else if (n_vis == 2) AH_TILE(4, 2, 1);
        else AH_TILE(8, 1, 1);
#undef AH_TILE
#undef AH_RING
    }
}

// ---- certified top-k screen of the re-rank (Cosine, DotProduct) ---------------------------------------------------
// `Reader::nns_by_leaf` needs the `count` smallest (distance, id) of ~10 000 candidates (src/reader.rs:381-399): the VALUE of
// a distance matters only for the ~1 % that can reach the top.  So the candidates are first evaluated on the binary16 shadow
// of the rows (half the bytes: the re-rank is bound by the HBM reads of its rows) with the screen dot product
// s = <q~, x~> and the RIGOROUS bound E >= |s - r| of screen_device.h on its distance to the reference's f32 dot product r
// (the six 2-norms are measured when the copies are made, gamma covers both accumulations).  The metric's epilogue f
// (cosine.rs:43-59, dot_product.rs:52-56) is non-increasing in r and every f32 operation in it is monotone, so
//     L = f(s + E) <= d_ref <= f(s - E) = U          for every candidate.
// Let T be the k-th smallest U over the (de-duplicated) candidates: at least k candidates have d_ref <= T, so a candidate with
// L > T cannot be among the k smallest (distance, id) — whatever the ties.  Only the survivors (L <= T: the top k and the few
// candidates within 2 E of them) are evaluated in the reference's f32 arithmetic, from their f32 rows, and ordered by
// (OrderedFloat(d), id): the same bits as the unscreened path, by construction and by test (every search test runs both).
// A non-finite screen value, or more survivors than the selection holds, sends the submission to the exact paths.
struct PairSeg {   // candidate list of one query of ah_rerank_batch: ids[off, off + n), k = min(count, n)  (api.hip: HostSeg)
    uint64_t off;
    uint32_t n, k;
};
struct PairTile {  // kPairTile candidates of one list, starting at `first`  (api.hip: HostTile)
    uint32_t query, first;
};
struct ScreenSearch {
    const uint16_t *rows16;     // n x hpitch halves (ah_dataset::d_rows_h16)
    float4 max_stats;           // component-wise maximum over the rows of {|x~|, |x - x~|, |x|}, rounded up: the bound built
                                // from it holds for every row and needs no per-candidate load (inf for a dataset with rows
                                // too small to be measured: nothing is screened then)
    float *aux;                 // Cosine: the stored norm of every candidate's row, next to its screen value
    uint32_t hpitch;
    float gamma_s, gamma_r;
    const uint16_t *q16;        // nq x hpitch halves
    const float4 *qstats;       // per query {|q~|, |q - q~|, |q|}, rounded up
    // Round 6 — the int8 copy of the rows as the FIRST stage (rows8 != nullptr): the screen values come from 1 byte per
    // dimension instead of 2, with an error bound per candidate, E = A_q x s_r (the query's part x the row's scale, forest.hip:
    // k_shadow_rows8); the interval argument of the selection is the same, only more candidates survive into the f32 pass
    // (~2.5 % instead of ~1.4 % at 1536-d) — 1.7 KB per candidate instead of 3.2.
    const int8_t *rows8;        // n x pitch8 (ah_dataset::d_rows_i8)
    const float *row_scale8;    // n (ah_dataset::d_scale8_rows; 0 for an all-zero row, inf for a row that must not be screened)
    const float *dim_scale;     // pitch8 powers of two (the columns' scales)
    uint32_t pitch8;
    float4 max8;                // dataset-wide maxima of {|q|, |y / s_r - q|, |x| / s_r} over the rows
    const int8_t *q8;           // nq x 2 x pitch8: the queries' two int8 digits in the column-scaled space
    const float4 *q8stats;      // per query {s_q, A_q, 0, 0}
    float *aux8;                // per candidate, next to its screen value: the row's scale s_r
};

// queries (f32 leaves at qvecs) -> two int8 digits of q' = q o d (d: the columns' power-of-two scales, so <q, x> = <q', y> with
// the rows' y = x / d) and the query's share of the error bound (forest.hip: k_forest_shadow_normals8 / screen8_decides — the
// query plays the normal's part): A_q = |q' - q~'| max|q8| + |q'| max|y/s - q8| + 2e-6 |q~'| max|q8| + gamma_r |q| max|x|/s,
// so that |s_r S - r_ref| <= A_q s_r for every row.  One wave per query.
__device__ __forceinline__ void query_i8(uint32_t q, uint32_t lane, const uint8_t *__restrict__ qvecs, uint64_t qstride, uint32_t dims,
                                         const ScreenSearch &ss, float gamma_r) {
    const float *v = reinterpret_cast<const float *>(qvecs + (uint64_t)q * qstride);
    int8_t *out_hi = const_cast<int8_t *>(ss.q8) + (uint64_t)q * 2 * ss.pitch8, *out_lo = out_hi + ss.pitch8;
    uint32_t mbits = 0;
    for (uint32_t i = lane; i < dims; i += 64) mbits = max(mbits, __float_as_uint(v[i] * ss.dim_scale[i]) & 0x7FFFFFFFu);
    for (int off = 32; off > 0; off >>= 1) mbits = max(mbits, (uint32_t)__shfl_xor((int)mbits, off));
    const float m = __uint_as_float(mbits);
    const bool ok = mbits >= kTinyBits && mbits < 0x7F800000u;  // finite, not (nearly) zero
    const float scale = ok ? m / 127.0f : 0.0f, inv_scale = ok ? 127.0f / m : 0.0f;
    float sa = 0.f, sb = 0.f, sc = 0.f, s0 = 0.f;
    for (uint32_t i = lane; i < ss.pitch8; i += 64) {
        const float x0 = i < dims ? v[i] : 0.0f;
        const float x = i < dims ? x0 * ss.dim_scale[i] : 0.0f;  // exact: a power of two
        const float t = x * inv_scale;
        const int qh = ok ? quantize8(x, inv_scale) : 0;
        const int ql = ok ? (int)fminf(fmaxf(rintf((t - (float)qh) * 256.0f), -127.0f), 127.0f) : 0;
        const float y = ((float)qh + (float)ql * 0.00390625f) * scale, d = x - y;  // the digits sum exactly (16 bits)
        sa += y * y;
        sb += d * d;
        sc += x * x;
        s0 += x0 * x0;
        out_hi[i] = (int8_t)qh;
        out_lo[i] = (int8_t)ql;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_xor(sa, off);
        sb += __shfl_xor(sb, off);
        sc += __shfl_xor(sc, off);
        s0 += __shfl_xor(s0, off);
    }
    if (lane == 0) {
        const float up = 1.0f + (float)(ss.pitch8 + 64u) * 1.2e-7f;
        const float inf = __uint_as_float(0x7F800000u);
        const float an = sqrtf(sa) * up, cn = sqrtf(sc) * up, cn0 = sqrtf(s0) * up;
        // a query that is all zero is exact (every digit 0, every product 0); one too small or not finite is never screened
        const float bn = ok ? sqrtf(sb) * up + 128.0f * scale * 6.0e-8f * sqrtf((float)ss.pitch8) : (mbits == 0u ? 0.0f : inf);
        const float a = (bn * ss.max8.x + cn * ss.max8.y + 2.0e-6f * (an * ss.max8.x) + gamma_r * (cn0 * ss.max8.z)) * 1.000001f * 1.002f;
        const_cast<float4 *>(ss.q8stats)[q] = make_float4(scale, a, 0.0f, 0.0f);
    }
}
__global__ __launch_bounds__(64) void k_queries_i8(const uint8_t *__restrict__ qvecs, uint64_t qstride, uint32_t dims, ScreenSearch ss) {
    query_i8(blockIdx.x, threadIdx.x, qvecs, qstride, dims, ss, ss.gamma_r);
}

// queries (f32 leaves at qvecs) -> binary16 copies + norms: one wave per query
// The leaf tile of k_leaf_tiles on the binary16 copies: R rows x Q queries of screen dot products per octet, any summation
// order (gamma_s covers it).  The value lands where the f32 tile would put the distance.
template <int R, int Q, int QO, int KF = 1>
__device__ __forceinline__ void leaf_tile16(const ScreenSearch &ss, const DataView &dv, const uint32_t *__restrict__ leaf_ids,
                                            uint32_t row_begin, uint32_t n_rows, const Visit *__restrict__ vis, uint32_t n_vis,
                                            float *__restrict__ dist, uint32_t stride, uint32_t *err, uint32_t *__restrict__ copy_ids_to = nullptr,
                                            const Visit *only_visit = nullptr) {
    constexpr uint32_t RO = 8 / QO;
    const uint32_t j = threadIdx.x & 7u, ow = (threadIdx.x >> 3) & 7u, wave = threadIdx.x >> 6;
    const uint32_t q_oct = ow % QO, row_oct = wave * RO + ow / QO;
    const uint32_t steps = ss.hpitch >> 6;
    const uint4 *q4[Q];
    uint64_t out[Q];  // index of the pair of (query t, first row of the leaf) in the candidate buffers
#pragma unroll
    for (int t = 0; t < Q; t++) {
        // (only_visit: the unit's one visit, already in the caller's registers — a trip to memory less on a small submission's chain)
        const Visit v = only_visit ? *only_visit : vis[min(q_oct * Q + (uint32_t)t, n_vis - 1)];
        q4[t] = reinterpret_cast<const uint4 *>(ss.q16 + (uint64_t)v.q * ss.hpitch) + j;
        out[t] = (uint64_t)v.q * stride + v.pos;
    }
    for (uint32_t r0 = row_begin + row_oct * R; r0 < n_rows; r0 += 4 * RO * R) {
        const uint4 *r4[R];
        bool missing[R];
        float xn[R];
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t id_u = leaf_ids[min(r0 + u, n_rows - 1)];
            // (a single query whose descent left the ids in the blob: SingleQueryOut::ids_by_tiles)
            if (copy_ids_to && q_oct == 0 && j == 0 && r0 + u < n_rows) copy_ids_to[r0 + u] = id_u;
            const uint64_t row = row_of_id(dv, id_u);
            missing[u] = row == ~0ull;
            r4[u] = reinterpret_cast<const uint4 *>(ss.rows16 + (missing[u] ? 0ull : row) * ss.hpitch) + j;
            xn[u] = ss.aux && !missing[u] ? dv.headers[row] : 0.0f;  // Cosine: the row's stored norm (cosine.rs:21-24)
        }
        float acc[R][Q];
#pragma unroll
        for (int u = 0; u < R; u++)
#pragma unroll
            for (int t = 0; t < Q; t++) acc[u][t] = 0.f;
        uint32_t k = 0;
        if constexpr (KF > 1) {
            // the small submissions' variant: KF steps of every row and query requested before the first use — with a dozen
            // leaves in the whole launch there is no other wave to switch to, and a step per trip to memory was 24 trips
            for (; k + KF <= steps; k += KF) {
                uint4 x[KF][R], y[KF][Q];
#pragma unroll
                for (int f = 0; f < KF; f++) {
#pragma unroll
                    for (int u = 0; u < R; u++) x[f][u] = r4[u][(k + f) * 8];
#pragma unroll
                    for (int t = 0; t < Q; t++) y[f][t] = q4[t][(k + f) * 8];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < KF; f++)
#pragma unroll
                    for (int u = 0; u < R; u++)
#pragma unroll
                        for (int t = 0; t < Q; t++) acc[u][t] = screen_dot8(x[f][u], y[f][t], acc[u][t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; k < steps; k++) {
            uint4 x[R], y[Q];
#pragma unroll
            for (int u = 0; u < R; u++) x[u] = r4[u][k * 8];
#pragma unroll
            for (int t = 0; t < Q; t++) y[t] = q4[t][k * 8];
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int t = 0; t < Q; t++) acc[u][t] = screen_dot8(x[u], y[t], acc[u][t]);
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
#pragma unroll
            for (int t = 0; t < Q; t++) {
                const float sdot = octet_sum(acc[u][t]);
                if (r0 + u >= n_rows || q_oct * Q + (uint32_t)t >= n_vis || j != 0) continue;
                if (missing[u]) atomicOr(err, 1u);
                dist[out[t] + r0 + u] = missing[u] ? __uint_as_float(0x7FC00000u) : sdot;
                if (ss.aux) ss.aux[out[t] + r0 + u] = xn[u];
                if (ss.aux8) ss.aux8[out[t] + r0 + u] = -1.0f;  // (a submission that mixes the stages: this value is a binary16 one)
            }
        }
    }
}
template <bool SMALL>
__global__ __launch_bounds__(256) void k_leaf_tiles16(DataView dv, ScreenSearch ss, const uint32_t *__restrict__ nns,
                                                      const Visit *__restrict__ sorted, const TileUnit *__restrict__ units,
                                                      const uint32_t *__restrict__ n_units_p, float *__restrict__ dist,
                                                      uint32_t stride, uint32_t *err, uint32_t min_vis = 0,
                                                      const uint32_t *__restrict__ blob = nullptr, uint32_t speculate = 0,
                                                      uint32_t *__restrict__ trace = nullptr, uint32_t flat = 0,
                                                      const uint32_t *__restrict__ items = nullptr, uint32_t per_query = 0) {
    const uint64_t t_start = trace ? wall_clock64() : 0ull;
    auto stamp = [&](uint32_t slot) {  // (AH_SEARCH_MULTI_TRACE: the latest block's time at each point, 10 ns ticks)
        if (trace && threadIdx.x == 0) atomicMax(&trace[slot], (uint32_t)(wall_clock64() - t_start));
    };
    if constexpr (SMALL) {
        if (flat) {
            // ONE query whose descent wrote the units itself (unit e = visit e = one leaf): block b is item b of the list of
            // (leaf, slab of 64 rows) pairs.  The 2-D grid of the general path dispatches 32 x (largest leaf / 64) = 768 blocks of
            // this 324-register kernel (one per CU at a time) to find ~156 with work: three rounds of dispatch, the last useful
            // block starting late — 21.8 us of kernel for 8 us of slab.  Here every wave reads the first 128 units and visits (one
            // trip, with the count), scans their slab counts and picks its item: the useful blocks are the FIRST blocks.
            // (a few queries a call: blockIdx.y is the query, its lists at [y * per_query, ...), its count at n_units_p[y])
            sorted += (size_t)blockIdx.y * per_query;
            units += (size_t)blockIdx.y * per_query;
            const uint32_t lane = threadIdx.x & 63u;
            const uint4 va = reinterpret_cast<const uint4 *>(sorted)[lane], vb = reinterpret_cast<const uint4 *>(sorted)[lane + 64u];
            const uint4 ua = reinterpret_cast<const uint4 *>(units)[lane], ub = reinterpret_cast<const uint4 *>(units)[lane + 64u];
            const uint32_t n_units = n_units_p[blockIdx.y];
            auto slab_of_leaf = [&](const Visit &v0, uint32_t pad, uint32_t e, uint32_t slab_index) {
                const uint32_t row_begin = slab_index * kTileSmallSlab, row_end = min(v0.n, row_begin + kTileSmallSlab);
                const uint32_t *leaf_ids = nns + (uint64_t)v0.q * stride + v0.pos;
                const bool from_blob = blob && pad != 0u;
                leaf_tile16<2, 1, 1, 24>(ss, dv, from_blob ? blob + (pad - 1u) : leaf_ids, row_begin, row_end, sorted + e, 1u, dist, stride, err,
                                         from_blob ? const_cast<uint32_t *>(leaf_ids) : nullptr, &v0);
            };
            if (n_units > 128u) {
                // (block-uniform, rare: a query that opens more leaves than one scan covers — a large search_k on an index of a
                // hundred trees.  Every block walks the visit list itself and takes the items of its number modulo the grid.)
                uint32_t item = 0;
                for (uint32_t e = 0; e < n_units; e++) {
                    const Visit v0 = sorted[e];
                    const uint32_t slabs = (v0.n + kTileSmallSlab - 1u) / kTileSmallSlab;
                    for (uint32_t sl = 0; sl < slabs; sl++, item++)
                        if (item % gridDim.x == blockIdx.x) slab_of_leaf(v0, units[e].pad, e, sl);
                }
                return;
            }
            const bool in0 = lane < n_units, in1 = lane + 64u < n_units;
            uint32_t p0 = in0 ? (va.w + kTileSmallSlab - 1u) / kTileSmallSlab : 0u, p1 = in1 ? (vb.w + kTileSmallSlab - 1u) / kTileSmallSlab : 0u;
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t a = __shfl_up(p0, d, 64), c = __shfl_up(p1, d, 64);
                if (lane >= d) {
                    p0 += a;
                    p1 += c;
                }
            }
            p1 += __shfl(p0, 63, 64);
            const uint32_t b = blockIdx.x;
            const uint32_t e = (uint32_t)__popcll(__ballot(in0 && p0 <= b)) + (uint32_t)__popcll(__ballot(in1 && p1 <= b));
            if (e >= n_units) return;  // block-uniform: beyond the last item
            const uint32_t before0 = __shfl(p0, (int)((e + 63u) & 63u), 64), before1 = __shfl(p1, (int)((e + 63u) & 63u), 64);
            const uint32_t before = e == 0u ? 0u : (e - 1u < 64u ? before0 : before1);
            const int src = (int)(e & 63u);
            const bool hi = e >= 64u;
            const Visit v0{(uint32_t)__shfl((int)(hi ? vb.x : va.x), src, 64), (uint32_t)__shfl((int)(hi ? vb.y : va.y), src, 64),
                           (uint32_t)__shfl((int)(hi ? vb.z : va.z), src, 64), (uint32_t)__shfl((int)(hi ? vb.w : va.w), src, 64)};
            const uint32_t pad = (uint32_t)__shfl((int)(hi ? ub.w : ua.w), src, 64);
            stamp(0);
            slab_of_leaf(v0, pad, e, b - before);
            stamp(1);
            return;
        }
    }
    // SMALL + speculate (the host: both tables hold at least gridDim.x entries): the block's first unit and the visit with the
    // unit's number are requested together with the unit count — a single query's units are its visits in order, and the chain
    // count -> unit -> visit -> ids -> rows of dependent trips to memory is most of such a launch's time
    TileUnit unit_first{};
    Visit visit_first{};
    if (SMALL && speculate) {
        unit_first = units[blockIdx.x];
        visit_first = sorted[blockIdx.x];
    }
    // items (a few queries a call, k_units_small): the (unit, slab) pairs with work, one block each — or, items[0] = 0xFFFFFFFF or
    // items == nullptr, the 2-D grid: blockIdx.x walks the units, blockIdx.y is the slab
    uint32_t n_items = SMALL && items ? items[0] : 0xFFFFFFFFu;
    const bool listed = n_items != 0xFFFFFFFFu;
    const uint32_t n_units = *n_units_p;
    const uint32_t n_work = listed ? n_items : n_units;
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        uint32_t u = w, slab_index = blockIdx.y;
        if (listed) {
            const uint32_t it = items[1u + w];
            u = it >> 16;
            slab_index = it & 0xFFFFu;
        }
        const bool spec = SMALL && speculate && !listed && u == blockIdx.x;
        const TileUnit unit = spec ? unit_first : units[u];
        if (unit.n_vis < min_vis) continue;  // (the units of few visits went to k_leaf_tiles8)
        const Visit *vis = sorted + unit.first;
        const Visit v0 = (spec && unit.first == u) ? visit_first : vis[0];
        const uint32_t n_leaf = v0.n;
        if (trace && n_leaf == 0xFFFFFFF0u) trace[7] = 1;  // (trace only: the unit and its visit have arrived at the stamp)
        stamp(0);
        // SMALL (its own kernel: the registers of the in-flight variant would cost the big submissions their occupancy): a small
        // submission — the leaves of one or two queries in slabs of kTileSmallSlab rows, two rows per octet with the whole row
        // in flight; the launch's grid.y counts those slabs
        const bool fly = SMALL && unit.n_vis <= 2;
        const uint32_t n_vis = unit.n_vis, slab = tile_slab_rows(n_vis, SMALL);
        // (an item names its slab; without the list the launch's grid.y counts the slabs — and a launch sized for the list that
        // finds it overflowed, grid.y = 1, walks them)
        for (; slab_index * slab < n_leaf; slab_index += listed ? 0xFFFFu : gridDim.y) {
        const uint32_t row_begin = slab_index * slab;
        const uint32_t row_end = min(n_leaf, row_begin + slab);
        const uint32_t *leaf_ids = nns + (uint64_t)v0.q * stride + v0.pos;
#define AH_TILE16(R, Q, QO) leaf_tile16<R, Q, QO, AH_TILES16_KF>(ss, dv, leaf_ids, row_begin, row_end, vis, n_vis, dist, stride, err)
        if constexpr (SMALL) {
            if (fly && n_vis == 1) {
                // (unit.pad != 0: the ids are still in the blob, this launch copies them — SingleQueryOut::ids_by_tiles)
                const bool from_blob = blob && unit.pad != 0u;
                leaf_tile16<2, 1, 1, 24>(ss, dv, from_blob ? blob + (unit.pad - 1u) : leaf_ids, row_begin, row_end, vis, n_vis, dist, stride, err,
                                         from_blob ? const_cast<uint32_t *>(leaf_ids) : nullptr, &v0);
                stamp(1);
                continue;
            }
            if (fly) {
                leaf_tile16<2, 2, 1, 12>(ss, dv, leaf_ids, row_begin, row_end, vis, n_vis, dist, stride, err);
                continue;
            }
        }
        if (n_vis > 8) AH_TILE16(4, 4, 4);
        else if (n_vis > 4) AH_TILE16(4, 4, 2);
        else if (n_vis > 2) AH_TILE16(4, 4, 1);
        else if (n_vis == 2) AH_TILE16(4, 2, 1);
        else AH_TILE16(8, 1, 1);
#undef AH_TILE16
        }
    }
}

// The leaf tile on the INT8 copy of the rows (round 6): R rows x Q queries per octet, both int8 digits of every query against the
// row's bytes (v_dot4_i32_i8, exact integers); the value s_q s_r (<hi, q8> + <lo, q8> / 256) lands where the binary16 tile would
// put its screen value, the row's scale next to it (aux8: the candidate's own error bound is A_q s_r).
template <int R, int Q, int QO>
__device__ __forceinline__ void leaf_tile8(const ScreenSearch &ss, const DataView &dv, const uint32_t *__restrict__ leaf_ids,
                                           uint32_t row_begin, uint32_t n_rows, const Visit *__restrict__ vis, uint32_t n_vis,
                                           float *__restrict__ dist, uint32_t stride, uint32_t *err) {
    constexpr uint32_t RO = 8 / QO;
    const uint32_t j = threadIdx.x & 7u, ow = (threadIdx.x >> 3) & 7u, wave = threadIdx.x >> 6;
    const uint32_t q_oct = ow % QO, row_oct = wave * RO + ow / QO;
    const uint32_t steps = ss.pitch8 >> 7, lo_off = ss.pitch8 >> 4;  // (uint4 units)
    const uint4 *q4[Q];
    uint64_t out[Q];
    float s_q[Q];
#pragma unroll
    for (int t = 0; t < Q; t++) {
        const Visit v = vis[min(q_oct * Q + (uint32_t)t, n_vis - 1)];
        q4[t] = reinterpret_cast<const uint4 *>(ss.q8 + (uint64_t)v.q * 2 * ss.pitch8) + j;
        out[t] = (uint64_t)v.q * stride + v.pos;
        s_q[t] = ss.q8stats[v.q].x;
    }
    for (uint32_t r0 = row_begin + row_oct * R; r0 < n_rows; r0 += 4 * RO * R) {
        const uint4 *r4[R];
        bool missing[R];
        float xn[R], sr[R];
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint64_t row = row_of_id(dv, leaf_ids[min(r0 + u, n_rows - 1)]);
            missing[u] = row == ~0ull;
            r4[u] = reinterpret_cast<const uint4 *>(ss.rows8 + (missing[u] ? 0ull : row) * ss.pitch8) + j;
            xn[u] = ss.aux && !missing[u] ? dv.headers[row] : 0.0f;
            sr[u] = missing[u] ? 0.0f : ss.row_scale8[row];
        }
        int ah[R][Q], al[R][Q];
#pragma unroll
        for (int u = 0; u < R; u++)
#pragma unroll
            for (int t = 0; t < Q; t++) ah[u][t] = al[u][t] = 0;
        for (uint32_t k = 0; k < steps; k++) {
            uint4 x[R], yh[Q], yl[Q];
#pragma unroll
            for (int u = 0; u < R; u++) x[u] = r4[u][k * 8];
#pragma unroll
            for (int t = 0; t < Q; t++) {
                yh[t] = q4[t][k * 8];
                yl[t] = q4[t][lo_off + k * 8];
            }
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int t = 0; t < Q; t++) {
                    ah[u][t] = dot16_i8(yh[t], x[u], ah[u][t]);
                    al[u][t] = dot16_i8(yl[t], x[u], al[u][t]);
                }
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
#pragma unroll
            for (int t = 0; t < Q; t++) {
                const float v = octet_sum((float)ah[u][t] + (float)al[u][t] * 0.00390625f);
                if (r0 + u >= n_rows || q_oct * Q + (uint32_t)t >= n_vis || j != 0) continue;
                if (missing[u]) atomicOr(err, 1u);
                dist[out[t] + r0 + u] = missing[u] ? __uint_as_float(0x7FC00000u) : (v * s_q[t]) * sr[u];
                ss.aux8[out[t] + r0 + u] = sr[u];
                if (ss.aux) ss.aux[out[t] + r0 + u] = xn[u];
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_leaf_tiles8(DataView dv, ScreenSearch ss, const uint32_t *__restrict__ nns,
                                                     const Visit *__restrict__ sorted, const TileUnit *__restrict__ units,
                                                     const uint32_t *__restrict__ n_units_p, float *__restrict__ dist,
                                                     uint32_t stride, uint32_t *err, uint32_t max_vis) {
    const uint32_t n_units = *n_units_p;
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const TileUnit unit = units[u];
        // Only the units of few visits: a row read for many queries is no longer the bound of its tile — the dot products are,
        // and they cost the same on either copy — while the int8 bound doubles the survivors of every query (measured, round 6:
        // 1000 queries from 64 base items 533 k -> 501 k queries/s with every unit on int8; 1000 unrelated ones 344 k -> 406 k)
        if (unit.n_vis > max_vis) continue;
        const Visit *vis = sorted + unit.first;
        const uint32_t n_leaf = vis[0].n;
        const uint32_t n_vis = unit.n_vis, slab = n_vis > 8 ? kTileSlab : 2 * kTileSlab;
        const uint32_t row_begin = blockIdx.y * slab;
        if (row_begin >= n_leaf) continue;
        const uint32_t row_end = min(n_leaf, row_begin + slab);
        const uint32_t *leaf_ids = nns + (uint64_t)vis[0].q * stride + vis[0].pos;
#define AH_TILE8(R, Q, QO) leaf_tile8<R, Q, QO>(ss, dv, leaf_ids, row_begin, row_end, vis, n_vis, dist, stride, err)
        if (n_vis > 8) AH_TILE8(4, 4, 4);
        else if (n_vis > 4) AH_TILE8(4, 4, 2);
        else if (n_vis > 2) AH_TILE8(4, 4, 1);
        else if (n_vis == 2) AH_TILE8(4, 2, 1);
        else AH_TILE8(8, 1, 1);
#undef AH_TILE8
    }
}

// Screen values of the candidate lists of ah_rerank_batch: one block per tile of a list, the query's binary16 copy in LDS, one
// octet per candidate gathers its binary16 row (2 x dims bytes instead of the 4 x dims of k_batch_distances_f32).
__global__ __launch_bounds__(256) void k_pairs_screen16(DataView dv, ScreenSearch ss, const PairSeg *__restrict__ segs,
                                                        const PairTile *__restrict__ tiles, uint32_t tile_candidates,
                                                        const uint32_t *__restrict__ ids, float *__restrict__ dist, uint32_t *err) {
    extern __shared__ uint4 s_q4[];  // hpitch / 8
    const PairTile tl = tiles[blockIdx.x];
    const PairSeg seg = segs[tl.query];
    const uint4 *g_q4 = reinterpret_cast<const uint4 *>(ss.q16 + (uint64_t)tl.query * ss.hpitch);
    for (uint32_t i = threadIdx.x; i < (ss.hpitch >> 3); i += blockDim.x) s_q4[i] = g_q4[i];
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u, steps = ss.hpitch >> 6;
    const uint32_t end = min(seg.n, tl.first + tile_candidates);
    for (uint32_t c = tl.first + (threadIdx.x >> 3); c < end; c += blockDim.x >> 3) {
        const uint64_t row = row_of_id(dv, ids[seg.off + c]);
        const bool missing = row == ~0ull;
        const uint4 *r4 = reinterpret_cast<const uint4 *>(ss.rows16 + (missing ? 0ull : row) * ss.hpitch) + j;
        float a0 = 0.f, a1 = 0.f;
        uint32_t k = 0;
        for (; k + 8 <= steps; k += 8) {
            uint4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = ld_stream_u4(r4 + (k + u) * 8);
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0 = screen_dot8(s_q4[(k + u) * 8 + j], x[u], a0);
                a1 = screen_dot8(s_q4[(k + u + 1) * 8 + j], x[u + 1], a1);
            }
        }
        for (; k < steps; k++) a0 = screen_dot8(s_q4[k * 8 + j], ld_stream_u4(r4 + k * 8), a0);
        const float sdot = octet_sum(a0 + a1);
        if (j == 0) {
            if (missing) atomicOr(err, 1u);
            dist[seg.off + c] = missing ? __uint_as_float(0x7FC00000u) : sdot;
            if (ss.aux) ss.aux[seg.off + c] = missing ? 0.0f : dv.headers[row];
        }
    }
}

// The same on the int8 copy of the rows (round 6): the query's two int8 digits in LDS, one octet per candidate gathers
// pitch8 bytes — the row's scale travels with the value (aux8) for the candidate's own error bound.
__global__ __launch_bounds__(256) void k_pairs_screen8(DataView dv, ScreenSearch ss, const PairSeg *__restrict__ segs,
                                                       const PairTile *__restrict__ tiles, uint32_t tile_candidates,
                                                       const uint32_t *__restrict__ ids, float *__restrict__ dist, uint32_t *err) {
    extern __shared__ uint4 s_q4[];  // 2 x pitch8 / 16: hi digits, lo digits
    const PairTile tl = tiles[blockIdx.x];
    const PairSeg seg = segs[tl.query];
    const uint4 *g_q4 = reinterpret_cast<const uint4 *>(ss.q8 + (uint64_t)tl.query * 2 * ss.pitch8);
    for (uint32_t i = threadIdx.x; i < (ss.pitch8 >> 3); i += blockDim.x) s_q4[i] = g_q4[i];
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u, steps = ss.pitch8 >> 7;
    const uint4 *hi4 = s_q4 + j, *lo4 = s_q4 + (ss.pitch8 >> 4) + j;
    const float s_q = ss.q8stats[tl.query].x;
    const uint32_t end = min(seg.n, tl.first + tile_candidates);
    for (uint32_t c = tl.first + (threadIdx.x >> 3); c < end; c += blockDim.x >> 3) {
        const uint64_t row = row_of_id(dv, ids[seg.off + c]);
        const bool missing = row == ~0ull;
        const uint4 *r4 = reinterpret_cast<const uint4 *>(ss.rows8 + (missing ? 0ull : row) * ss.pitch8) + j;
        int h0 = 0, h1 = 0, l0 = 0, l1 = 0;
        uint32_t k = 0;
        for (; k + 6 <= steps; k += 6) {
            uint4 x[6];
#pragma unroll
            for (int u = 0; u < 6; u++) x[u] = ld_stream_u4(r4 + (k + u) * 8);
#pragma unroll
            for (int u = 0; u < 6; u += 2) {
                h0 = dot16_i8(hi4[(k + u) * 8], x[u], h0);
                l0 = dot16_i8(lo4[(k + u) * 8], x[u], l0);
                h1 = dot16_i8(hi4[(k + u + 1) * 8], x[u + 1], h1);
                l1 = dot16_i8(lo4[(k + u + 1) * 8], x[u + 1], l1);
            }
        }
        for (; k < steps; k++) {
            const uint4 x = ld_stream_u4(r4 + k * 8);
            h0 = dot16_i8(hi4[k * 8], x, h0);
            l0 = dot16_i8(lo4[k * 8], x, l0);
        }
        const float v = octet_sum((float)(h0 + h1) + (float)(l0 + l1) * 0.00390625f);
        if (j == 0) {
            if (missing) atomicOr(err, 1u);
            const float s_r = missing ? 0.0f : ss.row_scale8[row];
            dist[seg.off + c] = missing ? __uint_as_float(0x7FC00000u) : (v * s_q) * s_r;  // (inf scale x 0 digits = NaN: not screened)
            ss.aux8[seg.off + c] = s_r;
            if (ss.aux) ss.aux[seg.off + c] = missing ? 0.0f : dv.headers[row];
        }
    }
}

// bounds of the reference distance of one candidate from its screen value (see above)
// E >= |s - r_ref| for every row of the dataset: screen accumulation, the two quantisations (Cauchy-Schwarz on measured norms,
// the rows' from their dataset-wide maxima: the bound is monotone in every one), the reference's own rounding
__device__ __forceinline__ float screened_error(const float4 rs, const float4 qs, float gamma_s, float gamma_r) {
    return (gamma_s * (qs.x * rs.x) + qs.y * rs.x + qs.z * rs.y + gamma_r * (qs.z * rs.z)) * 1.002f;
}
template <int METRIC>
__device__ __forceinline__ void screened_bounds(float sdot, float e_query, float qn, float xn, float &lo, float &hi) {
    const float e = e_query + 2.4e-7f * fabsf(sdot) + 1e-30f;  // + the rounding of s -+ E below
    const float r_lo = sdot - e, r_hi = sdot + e;
    if (METRIC == AH_DOT_PRODUCT) {  // d = -r (dot_product.rs:52-56)
        lo = -r_hi;
        hi = -r_lo;
    } else {                          // cosine.rs:43-59: non-increasing in r, every f32 step monotone
        lo = cosine_from_dot(r_hi, qn, xn);
        hi = cosine_from_dot(r_lo, qn, xn);
    }
}

// Selection of k_search_select with the screen in front: `dist_all` holds the SCREEN dot products of the candidates.
// err bit 2: a non-finite screen value or distance; bit 3: more survivors than kSelectCap.
template <int METRIC, bool FLAG>
__device__ __forceinline__ void search_select_screened_body(DataView dv, ScreenSearch ss, const uint32_t *__restrict__ nns,
                                                                 const float *__restrict__ dist_all, uint32_t stride,
                                                                 const uint32_t *__restrict__ counts,
                                                                 const uint32_t *__restrict__ unique, uint32_t k_out,
                                                                 const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                                 const float *__restrict__ qhdrs,
                                                                 uint32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                                 uint32_t *err, const PairSeg *__restrict__ segs,
                                                                 uint32_t flag_words, uint32_t id_limit,
                                                                 uint32_t *__restrict__ unique_out, uint32_t *__restrict__ trace = nullptr) {
    const uint64_t t_start = trace ? wall_clock64() : 0ull;
    auto stamp = [&](uint32_t slot) {  // (AH_SEARCH_MULTI_TRACE: where a single query's selection spends its time)
        if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[slot] = (uint32_t)(wall_clock64() - t_start);
    };
    // FLAG (a small submission of ah_search_batch, n <= 16 384): nns.dedup() done here, on the ids this block holds in registers
    // anyway — k_flag_duplicates' bitmap (flag_words words of dynamic LDS behind the query leaf) without its launch, its trip to
    // the candidate buffer and back, and the `unique` it would have left is written to unique_out
    constexpr uint32_t kBins = 2048, kCap = 1024, kThreads = 1024;
    extern __shared__ float4 s_qf4[];  // the query leaf in f32 (row pitch): the survivors' exact distances read it 143 times
    __shared__ uint32_t s_hist[kBins];
    __shared__ uint64_t s_key[kCap];
    __shared__ uint32_t s_pos[kCap];
    __shared__ float s_val[kCap];
    __shared__ uint32_t s_min, s_max, s_wave[kThreads / 64], s_bin, s_n, s_bad;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    // candidates of query q: a slot of `stride` entries (ah_search_batch), or a segment of the caller's lists (ah_rerank_batch)
    const uint64_t first = segs ? segs[q].off : (uint64_t)q * stride;
    const uint32_t n = segs ? segs[q].n : counts[q];
    uint32_t kk = FLAG ? 0u : (segs ? min(k_out, segs[q].k) : min(k_out, unique[q]));
    const uint32_t *ids = nns + first;
    const float *sd = dist_all + first;
    uint32_t *s_bits = reinterpret_cast<uint32_t *>(s_qf4 + (dv.pitch >> 2));  // FLAG: one bit per item id
    if constexpr (!FLAG) {
        for (uint32_t t = kk + tid; t < k_out; t += kThreads) {
            out_ids[(uint64_t)q * k_out + t] = 0xFFFFFFFFu;
            out_dist[(uint64_t)q * k_out + t] = __uint_as_float(0xFFFFFFFFu);
        }
        if (kk == 0) return;
    } else {
        for (uint32_t t = tid; t < flag_words / 4; t += kThreads) reinterpret_cast<uint4 *>(s_bits)[t] = make_uint4(0, 0, 0, 0);
    }
    for (uint32_t b = tid; b < kBins; b += kThreads) s_hist[b] = 0;
    {
        const float4 *g_q4 = reinterpret_cast<const float4 *>(qvecs + (uint64_t)q * qstride);
        for (uint32_t i = tid; i < (dv.pitch >> 2); i += kThreads) s_qf4[i] = g_q4[i];
    }
    if (tid == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0u;
        s_n = 0u;
        s_bad = 0u;
        s_bin = 0u;  // (FLAG: the count of distinct ids until the histogram's scan takes the word over)
    }
    __syncthreads();
    stamp(0);
    const float qn = qhdrs[2 * (uint64_t)q];
    // int8 first stage: the error bound is the query's A_q times the candidate's row scale; binary16: one number per query
    // (a negative "row scale": the value came from the binary16 rows — a submission may mix the two, see k_leaf_tiles8)
    const bool s8 = ss.aux8 != nullptr;
    const float e_query = ss.qstats ? screened_error(ss.max_stats, ss.qstats[q], ss.gamma_s, ss.gamma_r) : 0.0f;
    const float a8 = s8 ? ss.q8stats[q].y : 0.0f;
    const float *xns = ss.aux + (METRIC == AH_COSINE ? first : 0ull);
    const float *srs = ss.aux8 + (s8 ? first : 0ull);
    // Keys of a candidate: orderable(U) and orderable(L) from its screen value (the error bound is one number per query).  The
    // three passes below see every candidate; its keys are computed ONCE, into registers (thread t owns the candidates
    // t + 1024 r, r < kOwn, all their loads in flight together): three loops of dependent loads were 30 trips to memory for a
    // single query's 10 000 candidates.  What a list holds beyond 16 384 candidates is read again in every pass.
    constexpr uint32_t kOwn = 16;
    if (FLAG && n > kOwn * kThreads) {  // (the host does not choose FLAG for such a stride)
        if (tid == 0) atomicOr(err, 8u);
        return;
    }
    auto keys_of = [&](uint32_t id, float sdot, float xn, float sr, uint32_t &ukey, uint32_t &lkey) -> bool {
        if (!segs && id == 0xFFFFFFFFu) return false;  // (a flagged duplicate of the search; a caller's list may hold that id)
        if (!(fabsf(sdot) <= 3.0e38f)) {  // NaN or inf (a missing item, an overflow in binary16): not this path's business
            s_bad = 1u;
            return false;
        }
        float lo, hi;
        screened_bounds<METRIC>(sdot, (s8 && !(sr < 0.0f)) ? a8 * sr + 1.3e-7f * fabsf(sdot) : e_query, qn, xn, lo, hi);
        ukey = orderable_key(hi);
        lkey = orderable_key(lo);
        return true;
    };
    uint32_t own_u[kOwn], own_l[kOwn], own_id[kOwn], own_ok = 0;
    {
        float v_sd[kOwn], v_xn[kOwn], v_sr[kOwn];
#pragma unroll
        for (uint32_t r = 0; r < kOwn; r++) {
            const uint32_t g = tid + r * kThreads;
            own_id[r] = g < n ? ids[g] : 0xFFFFFFFFu;
            v_sd[r] = g < n ? sd[g] : 0.0f;
            v_xn[r] = (METRIC == AH_COSINE && g < n) ? xns[g] : 0.0f;
            v_sr[r] = (s8 && g < n) ? srs[g] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (trace && own_id[0] == 0x12345678u && v_sd[kOwn - 1] == 1.5f) trace[15] = 1;  // (trace only: the loads have arrived at the stamp)
        stamp(1);
        if constexpr (FLAG) {  // the second and later occurrences of an id drop out (which one stays does not matter: same id, same row)
            uint32_t mine = 0;
            bool bad_id = false;
#pragma unroll
            for (uint32_t r = 0; r < kOwn; r++) {
                if (tid + r * kThreads >= n) continue;
                if (own_id[r] >= id_limit) {
                    bad_id = true;
                    own_id[r] = 0xFFFFFFFFu;
                    continue;
                }
                const uint32_t bit = 1u << (own_id[r] & 31);
                if (atomicOr(&s_bits[own_id[r] >> 5], bit) & bit) own_id[r] = 0xFFFFFFFFu;
                else mine++;
            }
            if (bad_id) atomicOr(err, 1u);
            for (uint32_t d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
            if ((tid & 63u) == 0 && mine) atomicAdd(&s_bin, mine);
            __syncthreads();
            const uint32_t n_unique = s_bin;
            kk = min(k_out, n_unique);
            if (tid == 0) unique_out[q] = n_unique;
            for (uint32_t t = kk + tid; t < k_out; t += kThreads) {
                out_ids[(uint64_t)q * k_out + t] = 0xFFFFFFFFu;
                out_dist[(uint64_t)q * k_out + t] = __uint_as_float(0xFFFFFFFFu);
            }
            if (kk == 0) return;  // block-uniform
        }
        stamp(2);
#pragma unroll
        for (uint32_t r = 0; r < kOwn; r++) {
            own_u[r] = own_l[r] = 0;
            if (tid + r * kThreads < n && keys_of(own_id[r], v_sd[r], v_xn[r], v_sr[r], own_u[r], own_l[r])) own_ok |= 1u << r;
        }
    }
#define AH_SELECT_REST(G, ID, UK, LK, BODY)                                                     \
    for (uint32_t G = tid + kOwn * kThreads; G < n; G += kThreads) {                            \
        uint32_t UK, LK;                                                                        \
        const uint32_t ID = ids[G];                                                             \
        if (keys_of(ID, sd[G], METRIC == AH_COSINE ? xns[G] : 0.0f, s8 ? srs[G] : 0.0f, UK, LK)) { BODY; } \
    }
    uint32_t lo_k = 0xFFFFFFFFu, hi_k = 0u;
#pragma unroll
    for (uint32_t r = 0; r < kOwn; r++)
        if (own_ok >> r & 1u) {
            lo_k = min(lo_k, own_u[r]);
            hi_k = max(hi_k, own_u[r]);
        }
    AH_SELECT_REST(g, id, uk, lk, (lo_k = min(lo_k, uk), hi_k = max(hi_k, uk)))
    for (int off = 32; off > 0; off >>= 1) {
        lo_k = min(lo_k, (uint32_t)__shfl_xor((int)lo_k, off));
        hi_k = max(hi_k, (uint32_t)__shfl_xor((int)hi_k, off));
    }
    if ((tid & 63u) == 0) {
        atomicMin(&s_min, lo_k);
        atomicMax(&s_max, hi_k);
    }
    __syncthreads();
    if (s_bad || s_max == 0xFFFFFFFFu) {  // (a NaN bound: a NaN header or norm)
        if (tid == 0) atomicOr(err, 4u);
        return;
    }
    stamp(3);
    const uint32_t w_min = s_min;
    const uint64_t span = (uint64_t)(s_max - w_min) + 1ull;
    const bool direct = span <= kBins;
    const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)kBins << 32) / span);
    auto bin_of = [&](uint32_t w) -> uint32_t { return direct ? w - w_min : (uint32_t)(((uint64_t)(w - w_min) * scale) >> 32); };
#pragma unroll
    for (uint32_t r = 0; r < kOwn; r++)
        if (own_ok >> r & 1u) atomicAdd(&s_hist[bin_of(own_u[r])], 1u);
    AH_SELECT_REST(g, id, uk, lk, atomicAdd(&s_hist[bin_of(uk)], 1u))
    __syncthreads();
    {  // the bin that holds the k-th smallest U
        constexpr uint32_t kPer = kBins / kThreads;
        uint32_t c[kPer], mine = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            c[u] = s_hist[tid * kPer + u];
            mine += c[u];
        }
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int)(tid & 63u) >= off) incl += up;
        }
        if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += s_wave[w];
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            if (before < kk && before + c[u] >= kk) s_bin = tid * kPer + u;
            before += c[u];
        }
    }
    __syncthreads();
    stamp(4);
    const uint32_t bin_k = s_bin;
    // T = the largest key of that bin (>= the k-th smallest U: a valid, slightly generous threshold).  In closed form:
    // bin_of(w) <= bin_k  <=>  (w - w_min) * scale < (bin_k + 1) << 32  <=>  w - w_min <= ceil(((bin_k + 1) << 32) / scale) - 1
    uint32_t t_key;
    if (direct) {
        t_key = w_min + bin_k;
    } else {
        const uint64_t lim = ((uint64_t)bin_k + 1ull) << 32;
        const uint64_t d = (lim + scale - 1ull) / scale;
        t_key = (uint32_t)min((uint64_t)w_min + d - 1ull, (uint64_t)s_max);
    }
    // the survivors: their ids (s_pos), in the order the atomics hand out
#pragma unroll
    for (uint32_t r = 0; r < kOwn; r++)
        if ((own_ok >> r & 1u) && own_l[r] <= t_key) {
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < kCap) s_pos[at] = own_id[r];
        }
    AH_SELECT_REST(g, id, uk, lk, if (lk <= t_key) {
        const uint32_t at = atomicAdd(&s_n, 1u);
        if (at < kCap) s_pos[at] = id;
    })
#undef AH_SELECT_REST
    __syncthreads();
    stamp(5);
    const uint32_t n_sel = s_n;
    if (n_sel > kCap) {
        if (tid == 0) atomicOr(err, 8u);
        return;
    }
    if (tid == 0) {
        atomicAdd(&err[SS_SCREENED], 1u);
        atomicAdd(&err[SS_SURVIVORS], n_sel);
    }
    // the survivors in the reference's arithmetic: one octet per candidate (the f32 row against the f32 query leaf; the
    // reference's chains and tree, sixteen lines of the row requested before the first use)
    const uint32_t j = tid & 7u;
    bool bad = false;
    for (uint32_t e = tid >> 3; e < n_sel; e += kThreads >> 3) {
        const uint32_t id = s_pos[e];
        const uint64_t row = row_of_id(dv, id);
        const float xh = METRIC == AH_COSINE ? dv.headers[row] : 0.0f;
        const float r = octet_reduce_stream<OP_DOT, 16, true>(s_qf4, dv.rows_f32 + row * dv.pitch, dv.dims, j);
        const float d = METRIC == AH_COSINE ? cosine_from_dot(r, qn, xh) : -r;
        if (j == 0) {
            const uint32_t w = orderable_key(d);
            if (w > 0xFF7FFFFFu) bad = true;  // +inf / NaN: src/reader.rs:611-621 looks at positions
            s_key[e] = ((uint64_t)w << 32) | id;
            s_val[e] = d;
        }
    }
    if (bad) atomicOr(err, 4u);
    __syncthreads();
    stamp(6);
    // ascending (OrderedFloat(distance), id): a survivor's place is the number of smaller keys (unique: the id is their low
    // word) — n_sel broadcast reads per thread and one barrier instead of a sorting network's 36 - 55
    for (uint32_t e = tid; e < n_sel; e += kThreads) {
        const uint64_t mine = s_key[e];
        uint32_t rank = 0;
        uint32_t i = 0;
        for (; i + 8 <= n_sel; i += 8) {  // (eight broadcast reads in flight: the loop is LDS latency, not work)
            uint64_t other[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) other[u] = s_key[i + u];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) rank += (other[u] < mine || (other[u] == mine && i + u < e)) ? 1u : 0u;
        }
        for (; i < n_sel; i++) {  // (a caller's list may repeat an id: equal keys keep their order)
            const uint64_t other = s_key[i];
            rank += (other < mine || (other == mine && i < e)) ? 1u : 0u;
        }
        if (rank < kk) {
            out_ids[(uint64_t)q * k_out + rank] = (uint32_t)mine;
            out_dist[(uint64_t)q * k_out + rank] = normalized_distance(dv.metric, s_val[e], dv.dims);
        }
    }
    stamp(7);
    if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[9] = n_sel;
}

// host_status != nullptr (a small submission; out_ids / out_dist / unique_out are then the caller's pinned buffers): the block
// that finishes last copies the status words there as well, and the call needs no copy back — only the wait for this kernel.
template <int METRIC, bool FLAG = false>
__global__ __launch_bounds__(1024) void k_search_select_screened(DataView dv, ScreenSearch ss, const uint32_t *__restrict__ nns,
                                                                 const float *__restrict__ dist_all, uint32_t stride,
                                                                 const uint32_t *__restrict__ counts,
                                                                 const uint32_t *__restrict__ unique, uint32_t k_out,
                                                                 const uint8_t *__restrict__ qvecs, uint64_t qstride,
                                                                 const float *__restrict__ qhdrs,
                                                                 uint32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                                 uint32_t *err, const PairSeg *__restrict__ segs,
                                                                 uint32_t flag_words = 0, uint32_t id_limit = 0,
                                                                 uint32_t *__restrict__ unique_out = nullptr,
                                                                 uint32_t *__restrict__ host_status = nullptr,
                                                                 uint32_t *__restrict__ trace = nullptr) {
    const uint64_t t_start = trace ? wall_clock64() : 0ull;
    search_select_screened_body<METRIC, FLAG>(dv, ss, nns, dist_all, stride, counts, unique, k_out, qvecs, qstride, qhdrs, out_ids, out_dist,
                                              err, segs, flag_words, id_limit, unique_out, trace);
    if (host_status) {  // (every return of the body is block-uniform)
        // The results went to the caller's pinned buffers: system-scope fences, and word 0 of the status LAST — the host may be
        // polling that word instead of waiting for the stream (AH_SEARCH_SPIN_WAIT), and what it reads after it must be there.
        __threadfence_system();
        __syncthreads();
        // (one query: this block is the last one by construction — no counter, no second fence)
        // The first wave of the last block: lane w carries status word w (sixteen loads and stores side by side, not one after the
        // other), word 0 after a fence of its own, then the wipe for the next submission (Context::clean_status).
        if (threadIdx.x < 64u) {
            uint32_t last = 0;
            if (threadIdx.x == 0) last = (gridDim.x == 1u || atomicAdd(&err[SS_DONE], 1u) + 1u == gridDim.x) ? 1u : 0u;
            last = (uint32_t)__shfl((int)last, 0, 64);
            if (last) {  // wave-uniform
                if (gridDim.x != 1u) __threadfence_system();
                const uint32_t w = threadIdx.x;
                const uint32_t word = w < SS_WORDS ? __hip_atomic_load(&err[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                if (w >= 1u && w < SS_WORDS) host_status[w] = word;
                __threadfence_system();
                if (w == 0u) __hip_atomic_store(&host_status[0], word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (w < SS_WORDS) __hip_atomic_store(&err[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (trace && w == 0u) trace[8] = (uint32_t)(wall_clock64() - t_start);
            }
        }
    }
}

// The k smallest (OrderedFloat(distance), id) of one query's unflagged candidates, ascending, as (id, normalized
// distance); the slots beyond min(k, unique) are padded with 0xFFFFFFFF / NaN like k_batch_topk_emit does.  Selection as
// in k_batch_topk_select (batch.hip): 2048 linear bins over the distance words, the bin of the k-th key by a scan, a
// bitonic sort of the <= 1024 keys up to that bin.  err bit 2: a non-finite distance; bit 3: the selection does not fit.
static constexpr uint32_t kSelectBins = 2048, kSelectCap = 1024, kSelectThreads = 1024;
__global__ __launch_bounds__(kSelectThreads) void k_search_select(DataView dv, const uint32_t *__restrict__ nns,
                                                       const float *__restrict__ dist_all, uint32_t stride,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ unique, uint32_t k_out,
                                                       uint32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                                                       uint32_t *err) {
    __shared__ uint32_t s_hist[kSelectBins];
    __shared__ uint64_t s_key[kSelectCap];
    __shared__ uint32_t s_pos[kSelectCap];
    __shared__ uint32_t s_min, s_max, s_wave[kSelectThreads / 64], s_bin, s_count, s_n;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = counts[q], kk = min(k_out, unique[q]);
    const uint32_t *ids = nns + (uint64_t)q * stride;
    const float *dist = dist_all + (uint64_t)q * stride;
    for (uint32_t t = kk + tid; t < k_out; t += kSelectThreads) {
        out_ids[(uint64_t)q * k_out + t] = 0xFFFFFFFFu;
        out_dist[(uint64_t)q * k_out + t] = __uint_as_float(0xFFFFFFFFu);
    }
    if (kk == 0) return;
    for (uint32_t b = tid; b < kSelectBins; b += kSelectThreads) s_hist[b] = 0;
    if (tid == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0u;
        s_n = 0u;
    }
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t g = tid; g < n; g += kSelectThreads) {
        if (ids[g] == 0xFFFFFFFFu) continue;
        const uint32_t w = orderable_key(dist[g]);
        lo = min(lo, w);
        hi = max(hi, w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if ((tid & 63u) == 0) {
        atomicMin(&s_min, lo);
        atomicMax(&s_max, hi);
    }
    __syncthreads();
    const uint32_t w_min = s_min;
    if (s_max > 0xFF7FFFFFu) {  // +inf / NaN: the reference's rule looks at positions in the sorted list
        if (tid == 0) atomicOr(err, 4u);
        return;
    }
    const uint64_t span = (uint64_t)(s_max - w_min) + 1ull;
    const bool direct = span <= kSelectBins;
    const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)kSelectBins << 32) / span);
    auto bin_of = [&](uint32_t w) -> uint32_t {
        return direct ? w - w_min : (uint32_t)(((uint64_t)(w - w_min) * scale) >> 32);
    };
    for (uint32_t g = tid; g < n; g += kSelectThreads) {
        if (ids[g] == 0xFFFFFFFFu) continue;
        atomicAdd(&s_hist[bin_of(orderable_key(dist[g]))], 1u);
    }
    __syncthreads();
    {  // the bin of the k-th smallest key: thread t owns kPer consecutive bins
        constexpr uint32_t kPer = kSelectBins / kSelectThreads;
        uint32_t c[kPer], mine = 0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            c[u] = s_hist[tid * kPer + u];
            mine += c[u];
        }
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int)(tid & 63u) >= off) incl += up;
        }
        if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += s_wave[w];
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            if (before < kk && before + c[u] >= kk) {  // exactly one bin qualifies (kk <= unflagged candidates)
                s_bin = tid * kPer + u;
                s_count = before + c[u];
            }
            before += c[u];
        }
    }
    __syncthreads();
    const uint32_t n_sel = s_count, bin_k = s_bin;
    if (n_sel > kSelectCap) {
        if (tid == 0) atomicOr(err, 8u);
        return;
    }
    for (uint32_t g = tid; g < n; g += kSelectThreads) {
        const uint32_t id = ids[g];
        if (id == 0xFFFFFFFFu) continue;
        const uint32_t w = orderable_key(dist[g]);
        if (bin_of(w) <= bin_k) {
            const uint32_t at = atomicAdd(&s_n, 1u);
            s_key[at] = ((uint64_t)w << 32) | id;
            s_pos[at] = g;
        }
    }
    __syncthreads();
    uint32_t p2 = 64;
    while (p2 < n_sel) p2 <<= 1;
    for (uint32_t t = n_sel + tid; t < p2; t += kSelectThreads) s_key[t] = ~0ull;
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t str = size >> 1; str > 0; str >>= 1) {
            __syncthreads();
            for (uint32_t t = tid; t < (p2 >> 1); t += kSelectThreads) {
                const uint32_t a_i = 2 * t - (t & (str - 1)), b_i = a_i + str;
                const bool up = (a_i & size) == 0;
                const uint64_t x = s_key[a_i], y = s_key[b_i];
                if ((x > y) == up) {
                    s_key[a_i] = y;
                    s_key[b_i] = x;
                    const uint32_t px = s_pos[a_i];
                    s_pos[a_i] = s_pos[b_i];
                    s_pos[b_i] = px;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t t = tid; t < kk; t += kSelectThreads) {
        out_ids[(uint64_t)q * k_out + t] = (uint32_t)s_key[t];
        out_dist[(uint64_t)q * k_out + t] = normalized_distance(dv.metric, dist[s_pos[t]], dv.dims);
    }
}

// Larger candidate sets: bitonic steps in global memory, all queries of the batch per launch.
__global__ void k_pad_ids(uint32_t *nns, uint32_t stride, const uint32_t *counts, uint32_t np2) {
    const uint32_t q = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < np2 && t >= counts[q]) nns[(uint64_t)q * stride + t] = 0xFFFFFFFFu;
}
__global__ void k_bitonic_ids(uint32_t *nns, uint32_t stride, uint32_t np2, uint32_t size, uint32_t str) {
    const uint32_t q = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (np2 >> 1)) return;
    uint32_t *ids = nns + (uint64_t)q * stride;
    const uint32_t lo = 2 * t - (t & (str - 1)), hi = lo + str;
    const bool up = (lo & size) == 0;
    const uint32_t a = ids[lo], b = ids[hi];
    if ((a > b) == up) {
        ids[lo] = b;
        ids[hi] = a;
    }
}
// in-place dedup of a sorted list by one block per query (serial over chunks of 256, stable)
__global__ __launch_bounds__(256) void k_dedup_sorted(uint32_t *nns, uint32_t stride, uint32_t *counts) {
    __shared__ uint32_t s_flag[256];
    __shared__ uint32_t s_base;
    const uint32_t q = blockIdx.x;
    const uint32_t n = counts[q];
    uint32_t *ids = nns + (uint64_t)q * stride;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t c = 0; c < n; c += 256) {
        const uint32_t i = c + threadIdx.x;
        uint32_t v = 0, keep = 0;
        if (i < n) {
            v = ids[i];
            keep = (i == 0 || ids[i - 1] != v) ? 1u : 0u;
        }
        __syncthreads();  // every read of this chunk (and of ids[c-1]) happens before any write below
        s_flag[threadIdx.x] = keep;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t t = 0; t < threadIdx.x; t++) before += s_flag[t];
        const uint32_t base = s_base;
        // writes go to positions <= i, and ids[c-1 .. c+255] were already read by this chunk's threads
        if (keep) ids[base + before] = v;
        __syncthreads();
        if (threadIdx.x == 255) s_base = base + before + keep;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[q] = s_base;
}

__global__ void k_load_items_as_queries(DataView dv, const uint32_t *__restrict__ rows, uint8_t *qvecs, uint64_t qstride,
                                        float *qhdrs) {
    const uint32_t q = blockIdx.x;
    const uint64_t row = rows[q];
    const size_t words32 = metric_is_bq_dev(dv.metric) ? (size_t)dv.pitch * 2 : dv.pitch;
    const uint32_t *src = metric_is_bq_dev(dv.metric) ? reinterpret_cast<const uint32_t *>(dv.rows_bq + row * dv.pitch)
                                                      : reinterpret_cast<const uint32_t *>(dv.rows_f32 + row * dv.pitch);
    uint32_t *dst = reinterpret_cast<uint32_t *>(qvecs + q * qstride);
    for (uint32_t i = threadIdx.x; i < words32; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) {
        const uint32_t hf = dv.metric == AH_DOT_PRODUCT ? 2u : 1u;
        qhdrs[2 * q] = dv.headers[row * hf];
        qhdrs[2 * q + 1] = hf == 2 ? dv.headers[row * hf + 1] : 0.0f;
    }
}

// insert_items_in_descendants_from_frozen_reader (src/writer.rs:1398-1459) for every (tree, new item) pair at once:
// each pair walks from the root to the Descendants node the item lands in.  One octet per pair.
__global__ __launch_bounds__(256) void k_route_items(DataView dv, DataView nv, const DNode *__restrict__ nodes,
                                                     const uint32_t *__restrict__ roots, uint32_t n_trees,
                                                     const uint32_t *__restrict__ ids, uint64_t n,
                                                     const uint64_t *__restrict__ seeds, uint32_t *__restrict__ out_leaf,
                                                     uint32_t *err) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool live = pair < n * n_trees;
    const uint32_t t = live ? (uint32_t)(pair / n) : 0u;
    const uint64_t i = live ? pair % n : 0;
    const uint32_t id = live ? ids[i] : 0u;
    const uint64_t row = live ? row_of_id(dv, id) : ~0ull;
    bool active = live && row != ~0ull;
    if (live && row == ~0ull && j == 0) {
        atomicOr(err, 1u);
        out_leaf[pair] = 0xFFFFFFFFu;
    }
    const bool bq = metric_is_bq_dev(dv.metric);
    const void *leaf_vec = nullptr;
    LeafHdr lh = {0.0f, 0.0f};
    if (active) {
        leaf_vec = bq ? static_cast<const void *>(dv.rows_bq + row * dv.pitch)
                      : static_cast<const void *>(dv.rows_f32 + row * dv.pitch);
        const uint32_t hf = dv.metric == AH_DOT_PRODUCT ? 2u : 1u;
        lh.h0 = dv.headers[row * hf];
    }
    uint32_t node = live ? roots[t] : 0u;
    while (__any(active)) {
        if (!active) continue;
        const DNode nd = nodes[node];
        if ((nd.kind & 0xFFu) == AH_NODE_DESCENDANTS) {
            if (j == 0) out_leaf[pair] = node;
            active = false;
            continue;
        }
        uint32_t right;
        if (nd.kind & 0x100u) right = side_of_margin(descent_margin(nv, nd.c, leaf_vec, lh, j));  // D::side
        else right = ah_route_side_is_left(seeds[t], node, id) ^ 1u;                               // Side::random
        node = right ? nd.b : nd.a;
    }
}

__global__ void k_filter_bitmap(const uint32_t *__restrict__ ids, uint64_t n, uint32_t *bits) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride)
        atomicOr(&bits[ids[g] >> 5], 1u << (ids[g] & 31));
}

// normal records [vector (row_bytes)][header (16)] -> row matrix + header array of the normals view
// Only the `valid_words32` words of the stored vector (ah_vector_size bytes) are read from the record: caller views are
// compact ([header][vector], stride hs + vs), so the device pitch beyond them is zero-filled, never copied.
__global__ void k_unpack_normals(const uint8_t *__restrict__ recs, const uint64_t *__restrict__ offsets, uint32_t n,
                                 uint64_t vec_off, uint64_t hdr_off, uint32_t valid_words32, uint32_t row_words32,
                                 uint32_t hf, uint32_t *__restrict__ rows, float *__restrict__ headers) {
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(recs + offsets[r] + vec_off);
    for (uint32_t i = threadIdx.x; i < row_words32; i += blockDim.x)
        rows[(uint64_t)r * row_words32 + i] = i < valid_words32 ? src[i] : 0u;
    if (threadIdx.x < hf)
        headers[(uint64_t)r * hf + threadIdx.x] = reinterpret_cast<const float *>(recs + offsets[r] + hdr_off)[threadIdx.x];
}

// The certified top-k screen for the candidate lists of ah_rerank_batch (api.hip): screen values of every (query, candidate)
// pair from the binary16 rows, then per query the survivors in f32 and their order.  The queries are already prepared
// (d_qvecs / d_qhdrs).  d_q16 / d_qstats / d_aux: scratch of nq x hpitch halves, nq float4, and (Cosine) one float per candidate.
// err bits as k_search_select_screened: the caller redoes the submission on the exact path when 4 or 8 is raised.
int launch_rerank_screened(ah_dataset *ds, uint32_t nq, const uint8_t *d_qvecs, uint64_t qstride, const float *d_qhdrs,
                           const void *d_segs, const void *d_tiles, uint32_t tile_first, uint32_t n_tiles, uint32_t tile_candidates,
                           const uint32_t *d_ids, float *d_dist, float *d_aux, uint16_t *d_q16, float4 *d_qstats, uint32_t k_out,
                           uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_err, hipStream_t s, bool first, bool select,
                           int8_t *d_q8, float4 *d_q8stats, float *d_aux8) {
    const DataView dv = ds->view();
    ScreenSearch ss{};
    ss.rows16 = ds->d_rows_h16;
    ss.max_stats = make_float4(ds->screen_max[0], ds->screen_max[1], ds->screen_max[2], 0.0f);
    ss.aux = ds->metric == AH_COSINE ? d_aux : nullptr;
    ss.hpitch = ds->hpitch;
    ss.gamma_s = (float)(4.0 * (2.0 * (ds->hpitch / 16) + 8.0) * 5.9604645e-8);
    ss.gamma_r = (float)(4.0 * ((double)(ds->dims / 32) + 6.0 + 62.0) * 5.9604645e-8);
    ss.q16 = d_q16;
    ss.qstats = d_qstats;
    if (d_q8) {  // the int8 copy of the rows first (the caller has made sure it exists: ensure_screen8_search)
        ss.rows8 = ds->d_rows_i8;
        ss.row_scale8 = ds->d_scale8_rows;
        ss.dim_scale = ds->d_dim_scale;
        ss.pitch8 = ds->pitch8;
        ss.max8 = make_float4(ds->screen8_max[0], ds->screen8_max[1], ds->screen8_max[2], 0.0f);
        ss.q8 = d_q8;
        ss.q8stats = d_q8stats;
        ss.aux8 = d_aux8;
        ss.qstats = nullptr;  // (every candidate of a list is screened on the int8 rows: no binary16 copy of the queries is made)
        if (first) hipLaunchKernelGGL(k_queries_i8, dim3(nq), dim3(64), 0, s, d_qvecs, qstride, ds->dims, ss);
        const size_t sh8 = (size_t)ds->pitch8 * 2;
        if (sh8 > 48 * 1024)
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pairs_screen8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh8));
        if (n_tiles)
            hipLaunchKernelGGL(k_pairs_screen8, dim3(n_tiles), dim3(256), sh8, s, dv, ss, reinterpret_cast<const PairSeg *>(d_segs),
                               reinterpret_cast<const PairTile *>(d_tiles) + tile_first, tile_candidates, d_ids, d_dist, d_err);
    } else {
        if (first) hipLaunchKernelGGL(k_queries_h16, dim3(nq), dim3(64), 0, s, d_qvecs, qstride, ds->dims, ds->hpitch, d_q16, d_qstats);
        const size_t sh = (size_t)ds->hpitch * 2;
        if (sh > 48 * 1024)
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pairs_screen16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        if (n_tiles)
            hipLaunchKernelGGL(k_pairs_screen16, dim3(n_tiles), dim3(256), sh, s, dv, ss, reinterpret_cast<const PairSeg *>(d_segs),
                               reinterpret_cast<const PairTile *>(d_tiles) + tile_first, tile_candidates, d_ids, d_dist, d_err);
    }
    if (!select) {
        AH_HIP(hipGetLastError());
        return AH_OK;
    }
    const size_t sel_lds = (size_t)ds->pitch * 4;
    if (sel_lds > 32 * 1024) {
        AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_select_screened<AH_COSINE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
        AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_select_screened<AH_DOT_PRODUCT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
    }
    if (ds->metric == AH_COSINE)
        hipLaunchKernelGGL((k_search_select_screened<AH_COSINE>), dim3(nq), dim3(1024), sel_lds, s, dv, ss, d_ids, d_dist, 0u,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, k_out, d_qvecs, qstride, d_qhdrs, d_out_ids,
                           d_out_dist, d_err, reinterpret_cast<const PairSeg *>(d_segs));
    else
        hipLaunchKernelGGL((k_search_select_screened<AH_DOT_PRODUCT>), dim3(nq), dim3(1024), sel_lds, s, dv, ss, d_ids, d_dist, 0u,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, k_out, d_qvecs, qstride, d_qhdrs, d_out_ids,
                           d_out_dist, d_err, reinterpret_cast<const PairSeg *>(d_segs));
    AH_HIP(hipGetLastError());
    return AH_OK;
}

}  // namespace ah

using namespace ah;

struct ah_index {
    ah_dataset *ds = nullptr;
    DataView nv{};  // the normals as a row matrix
    DNode *d_nodes = nullptr;
    uint32_t *d_roots = nullptr, *d_desc = nullptr;
    void *d_nrows = nullptr;
    float *d_nhdrs = nullptr;
    uint32_t n_trees = 0, n_nodes = 0, n_normals = 0, max_desc = 0;
    uint32_t n_leaves = 0;  // Descendants nodes (desc_len / n_leaves: the mean leaf, what the small-submission gate estimates with)
    uint64_t desc_len = 0;
    std::mutex stats_mu;       // ah_search_batch may run on any number of threads
    ah_search_stats stats{};
    std::atomic<uint32_t> search8_fails{0};  // sub-batches whose int8 stage left too many survivors ...
    std::atomic<bool> search8_off{false};    // ... eight of them: the index's tile re-rank starts on the binary16 rows from now on
};

extern "C" {

int ah_index_create_from_view(ah_dataset *ds, const ah_forest_view *view, ah_index **out);

int ah_index_destroy(ah_index *ix) {
    AH_GUARDED("ah_index_destroy")
    if (!ix) return AH_OK;
    NoFailScope no_fail;
    if (ix->ds) (void)hipSetDevice(ix->ds->device);
    (void)hipDeviceSynchronize();
    if (ix->d_nodes) (void)dev_free(ix->d_nodes);
    if (ix->d_roots) (void)dev_free(ix->d_roots);
    if (ix->d_desc) (void)dev_free(ix->d_desc);
    if (ix->d_nrows) (void)dev_free(ix->d_nrows);
    if (ix->d_nhdrs) (void)dev_free(ix->d_nhdrs);
    delete ix;
    return AH_OK;
    AH_GUARDED_END
}

// Mirror a forest in HBM next to its dataset.  The forest handle may be destroyed afterwards.
int ah_index_create(ah_dataset *ds, const ah_forest *forest, ah_index **out) {
    AH_GUARDED("ah_index_create")
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(forest, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    ah_forest_view v;
    AH_TRY(ah_forest_view_get(forest, &v));
    return ah_index_create_from_view(ds, &v, out);
    AH_GUARDED_END
}

// Same from caller-owned arrays (e.g. tree nodes decoded from LMDB by `Reader::open`); nothing is retained.
int ah_index_create_from_view(ah_dataset *ds, const ah_forest_view *view, ah_index **out) {
    AH_GUARDED("ah_index_create_from_view")
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(ds && view, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE(ds->finalized, AH_ERR_NOT_FINALIZED, "dataset not finalized");
    const ah_forest_view v = *view;
    AH_REQUIRE(v.n_nodes == 0 || v.nodes, AH_ERR_INVALID_ARGUMENT, "nodes is NULL");
    AH_REQUIRE(v.n_trees == 0 || v.roots, AH_ERR_INVALID_ARGUMENT, "roots is NULL");
    AH_REQUIRE(v.descendants_len == 0 || v.descendants, AH_ERR_INVALID_ARGUMENT, "descendants is NULL");
    AH_REQUIRE(v.descendants_len < 0xFFFFFFFFull && v.n_nodes < 0xFFFFFFFFull, AH_ERR_INVALID_ARGUMENT,
               "forest too large for 32-bit node / descendant offsets");
    {
        // record geometry: the vector and the header must lie inside a record, on 4-byte boundaries
        const uint64_t hs = ah_header_size(ds->metric), vs = ah_vector_size(ds->metric, ds->dims);
        bool any_normal = false;
        for (uint64_t i = 0; i < v.n_nodes && !any_normal; i++) any_normal = v.nodes[i].kind == AH_NODE_SPLIT && v.nodes[i].has_normal;
        if (any_normal) {
            AH_REQUIRE(v.normals, AH_ERR_INVALID_ARGUMENT, "normals is NULL");
            AH_REQUIRE(v.normal_vector_offset + vs <= v.normal_stride && v.normal_header_offset + hs <= v.normal_stride,
                       AH_ERR_INVALID_ARGUMENT, "normal vector / header do not fit the record stride %llu",
                       (unsigned long long)v.normal_stride);
            AH_REQUIRE((v.normal_vector_offset & 3) == 0 && (v.normal_header_offset & 3) == 0, AH_ERR_INVALID_ARGUMENT,
                       "normal vector / header offsets must be multiples of 4");
        }
    }
    for (uint64_t i = 0; i < v.n_nodes; i++) {
        const ah_node &nd = v.nodes[i];
        if (nd.kind == AH_NODE_SPLIT) {
            AH_REQUIRE(nd.left < v.n_nodes && nd.right < v.n_nodes, AH_ERR_INVALID_ARGUMENT, "node %llu: child out of range",
                       (unsigned long long)i);
            AH_REQUIRE(!nd.has_normal || ((nd.offset & 3) == 0 && nd.offset + v.normal_stride <= v.normals_len),
                       AH_ERR_INVALID_ARGUMENT, "node %llu: normal record out of range or misaligned", (unsigned long long)i);
        } else {
            AH_REQUIRE(nd.kind == AH_NODE_DESCENDANTS && nd.offset + nd.count <= v.descendants_len, AH_ERR_INVALID_ARGUMENT,
                       "node %llu: bad kind or descendants out of range", (unsigned long long)i);
        }
    }
    for (uint32_t t = 0; t < v.n_trees; t++)
        AH_REQUIRE(v.roots[t] < v.n_nodes, AH_ERR_INVALID_ARGUMENT, "root %u out of range", t);
    {
        // The view must be a forest: every node reachable from at most one root / parent (no cycle, no shared sub-tree).
        // The descent relies on it: a cycle would never terminate, and a Descendants node popped twice would overflow
        // the per-query candidate buffer, which is sized for every Descendants node being collected at most once.
        std::vector<uint8_t> seen(v.n_nodes, 0);
        std::vector<uint32_t> stack;
        uint64_t total_desc = 0;
        for (uint32_t t = 0; t < v.n_trees; t++) {
            stack.push_back(v.roots[t]);
            while (!stack.empty()) {
                const uint32_t i = stack.back();
                stack.pop_back();
                AH_REQUIRE(!seen[i], AH_ERR_INVALID_ARGUMENT,
                           "node %u is reachable twice (cycle or shared sub-tree): the view is not a forest", i);
                seen[i] = 1;
                const ah_node &nd = v.nodes[i];
                if (nd.kind == AH_NODE_SPLIT) {
                    stack.push_back(nd.left);
                    stack.push_back(nd.right);
                } else {
                    total_desc += nd.count;
                }
            }
        }
        AH_REQUIRE(total_desc <= v.descendants_len, AH_ERR_INVALID_ARGUMENT,
                   "the Descendants nodes hold %llu ids but the blob has %llu: ranges overlap",
                   (unsigned long long)total_desc, (unsigned long long)v.descendants_len);
    }
    AH_HIP(hipSetDevice(ds->device));
    ah_index *ix = new (std::nothrow) ah_index();
    AH_REQUIRE(ix, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
    struct IndexGuard {  // every way out before the last line (error codes AND exceptions) destroys the half-made index
        ah_index *p;
        ~IndexGuard() {
            if (p) (void)ah_index_destroy(p);
        }
    } guard{ix};
    ix->ds = ds;
    ix->n_trees = v.n_trees;
    ix->n_nodes = (uint32_t)v.n_nodes;
    ix->desc_len = v.descendants_len;
    std::vector<DNode> nodes(v.n_nodes);
    std::vector<uint64_t> offsets;
    for (uint64_t i = 0; i < v.n_nodes; i++) {
        const ah_node &nd = v.nodes[i];
        DNode d{};
        d.kind = nd.kind;
        if (nd.kind == AH_NODE_SPLIT) {
            d.a = nd.left;
            d.b = nd.right;
            if (nd.has_normal) {
                d.kind |= 0x100u;
                d.c = (uint32_t)offsets.size();
                offsets.push_back(nd.offset);
            }
        } else {
            d.a = (uint32_t)nd.offset;
            d.b = nd.count;
            ix->max_desc = std::max(ix->max_desc, nd.count);
            ix->n_leaves++;
        }
        nodes[i] = d;
    }
    ix->n_normals = (uint32_t)offsets.size();
    const bool bq = metric_is_bq(ds->metric);
    const uint32_t hf = header_floats(ds->metric);
    const size_t row_bytes = ds->row_bytes();
    int st = AH_OK;
    auto fail = [&](int code) { return code; };  // (the guard destroys the index)
#define AH_IX(expr)                                                                        \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                      \
            return fail(_e == hipErrorOutOfMemory ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE); \
        }                                                                                  \
    } while (0)
    AH_IX(dev_malloc((void **)&ix->d_nodes, std::max<size_t>(1, nodes.size()) * sizeof(DNode)));
    AH_IX(dev_malloc((void **)&ix->d_roots, std::max<size_t>(1, v.n_trees) * 4));
    AH_IX(dev_malloc((void **)&ix->d_desc, std::max<uint64_t>(1, v.descendants_len) * 4));
    AH_IX(dev_malloc(&ix->d_nrows, std::max<size_t>(1, ix->n_normals) * row_bytes));
    AH_IX(dev_malloc((void **)&ix->d_nhdrs, std::max<size_t>(1, ix->n_normals) * hf * 4));
    if (!nodes.empty()) AH_IX(hipMemcpy(ix->d_nodes, nodes.data(), nodes.size() * sizeof(DNode), hipMemcpyHostToDevice));
    if (v.n_trees) AH_IX(hipMemcpy(ix->d_roots, v.roots, v.n_trees * 4, hipMemcpyHostToDevice));
    if (v.descendants_len)
        AH_IX(hipMemcpy(ix->d_desc, v.descendants, v.descendants_len * 4, hipMemcpyHostToDevice));
    if (ix->n_normals) {
        DevMem recs, offs;
        AH_IX(dev_malloc(&recs.p, v.normals_len));
        AH_IX(dev_malloc(&offs.p, offsets.size() * 8));
        AH_IX(hipMemcpy(recs.p, v.normals, v.normals_len, hipMemcpyHostToDevice));
        AH_IX(hipMemcpy(offs.p, offsets.data(), offsets.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_unpack_normals, dim3(ix->n_normals), dim3(256), 0, 0, recs.as<uint8_t>(), offs.as<uint64_t>(),
                           ix->n_normals, v.normal_vector_offset, v.normal_header_offset,
                           (uint32_t)(ah_vector_size(ds->metric, ds->dims) / 4), (uint32_t)(row_bytes / 4), hf,
                           reinterpret_cast<uint32_t *>(ix->d_nrows), ix->d_nhdrs);
        AH_IX(hipDeviceSynchronize());
    }
#undef AH_IX
    (void)st;
    // the binary16 shadow of the rows for the certified top-k screen of the re-rank (made once per dataset; the forest build
    // of an f32 dataset has usually made it already).  No memory for it = no screen, not an error.
    if (tun(TUN_SEARCH_SCREEN) != 0 && (ds->metric == AH_COSINE || ds->metric == AH_DOT_PRODUCT) && ds->dims >= 32 && ds->n) {
        ContextLease lease(ds);
        if (lease.c && ensure_screen(ds, lease.c->stream, false) && tun(TUN_SEARCH_SCREEN8) != 0)
            (void)ensure_screen8_search(ds, lease.c->stream);  // ... and the int8 copy in front of it (round 6), likewise
    }
    ix->nv.metric = ds->metric;
    ix->nv.dims = ds->dims;
    ix->nv.pitch = ds->pitch;
    ix->nv.words = ds->words;
    ix->nv.n = ix->n_normals;
    ix->nv.rows_f32 = bq ? nullptr : reinterpret_cast<const float *>(ix->d_nrows);
    ix->nv.rows_bq = bq ? reinterpret_cast<const uint64_t *>(ix->d_nrows) : nullptr;
    ix->nv.headers = ix->d_nhdrs;
    ix->nv.ids = nullptr;
    ix->nv.lut = nullptr;
    ix->nv.lut_len = 0;
    ix->nv.identity_ids = 1;
    guard.p = nullptr;
    *out = ix;
    return AH_OK;
    AH_GUARDED_END
}

struct HostSeg2 {
    uint64_t off;
    uint32_t n, k;
};
struct HostTile2 {
    uint32_t query, first;
};

// Counters of one sub-batch -> the index's ah_search_stats.
struct ChunkStats {
    ah_search_stats s{};
    void device_words(const uint32_t *w) {
        s.descent_wave_small += w[SS_WAVE_SMALL];
        s.descent_wave_big += w[SS_WAVE_BIG];
        s.descent_octet_lds += w[SS_OCTET_LDS];
        s.descent_octet_global += w[SS_OCTET_GLOBAL];
        s.descent_block += w[SS_BLOCK];
        s.descent_multi += w[SS_MULTI];
        s.tile_units_16 += w[SS_UNITS_16];
        s.tile_units_8 += w[SS_UNITS_8];
        s.tile_units_4 += w[SS_UNITS_4];
        s.tile_visits += w[SS_VISITS];
        s.rerank_screened += w[SS_SCREENED];
        s.screen_survivors += w[SS_SURVIVORS];
    }
    void commit(ah_index *ix) {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        uint64_t *dst = reinterpret_cast<uint64_t *>(&ix->stats);
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&s);
        for (size_t i = 0; i < sizeof(ah_search_stats) / 8; i++) dst[i] += src[i];
    }
};

static int search_chunk(ah_index *ix, Context *ctx, const float *queries, const uint32_t *query_rows, size_t nq,
                        size_t count, uint32_t search_k, uint32_t nns_stride, const uint32_t *d_filter_bits,
                        uint64_t filter_len_bits, double filter_share, bool wave_descent, const uint32_t *d_leaf_kept,
                        uint32_t *out_ids, float *out_dists, uint32_t *out_counts, bool allow8 = true) {
    ah_dataset *ds = ix->ds;
    hipStream_t s = ctx->stream;
    const size_t qstride = (ds->row_bytes() + 255) & ~(size_t)255;
    const size_t k = count;
    // device scratch (one carve for the whole chunk)
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const uint32_t max_tiles_bound = (uint32_t)(nq * ((size_t)nns_stride / batch_tile_candidates() + 1));
    // count <= 2048: the batched tournament top-k; beyond: the single-query kernels, one query after the other (the
    // reference accepts any count, `Reader::nns(count)`; large counts are rare and not a throughput path)
    const bool big_k = !batch_supported((uint32_t)std::min<size_t>(k, 0xFFFFFFFFu));
    const size_t kstride = big_k ? (topk_scratch_bytes(nns_stride, std::min<size_t>(k, nns_stride)) + 15) / 16 : batch_key_stride(nns_stride);
    const uint32_t heap_cap = ix->n_nodes + ix->n_trees + 2;
    size_t dev_bytes = pad(nq * (size_t)ds->dims * 4) + pad(nq * 4) + nq * qstride + pad(nq * 8) + pad(nq * (size_t)nns_stride * 4) * 2 +
                       pad(nq * 4) * 3 + pad(nq * sizeof(HostSeg2)) + pad((size_t)max_tiles_bound * sizeof(HostTile2)) +
                       2 * pad(nq * kstride * 8) + pad(nq * k * 4) * 2 + pad(SS_WORDS * 4) + pad(nq * 4) + 4096;
    // counters of the row-major re-rank, reserved when the candidate lists could be long enough for it
    const size_t inv_bytes = batch_invert_wanted(ds->view(), (uint64_t)nq * nns_stride) ? batch_invert_counter_bytes(ds->n, (uint64_t)nq * nns_stride) : 0;
    dev_bytes += pad(inv_bytes);
    // leaf-tile re-rank (see k_leaf_tiles): which submissions take it, and its scratch
    const uint32_t max_id = ds->identity_ids ? (uint32_t)(ds->n - 1) : ds->last_id;
    const uint32_t bitmap_words = (uint32_t)(((uint64_t)max_id / 32 + 1 + 1023) / 1024 * 1024);
    const bool bitmap_fits = tun(TUN_SEARCH_BITMAP) != 0 && bitmap_words <= kBitmapMaxWords;
    // nns.dedup() of the tiles: the LDS bitmap, or a hash set of the candidates when the id space is too big for it
    const bool hash_fits = nns_stride <= kHashMaxCandidates && max_id != 0xFFFFFFFFu;
    const bool tiles = tun(TUN_SEARCH_TILES) != 0 && (bitmap_fits || hash_fits) && !big_k && ds->dims >= 32 &&
                       ix->max_desc <= 65535u * kTileSlab &&
                       (ds->metric == AH_EUCLIDEAN || ds->metric == AH_COSINE || ds->metric == AH_DOT_PRODUCT);
    const uint32_t visit_cap = (uint32_t)std::min<uint64_t>((uint64_t)nq * nns_stride, 2u << 20);
    const uint32_t n_leaf_sums = (ix->n_nodes + kLeafScanItems - 1) / kLeafScanItems;
    // certified top-k screen of the tile re-rank: the binary16 shadow of the rows must exist (ah_index_create makes it)
    const bool screened = tiles && tun(TUN_SEARCH_SCREEN) != 0 && (ds->metric == AH_COSINE || ds->metric == AH_DOT_PRODUCT) &&
                          ds->screen_ready.load(std::memory_order_acquire);  // (published: common.h)
    if (screened) dev_bytes += pad(nq * (size_t)ds->hpitch * 2) + pad(nq * sizeof(float4)) + pad(nq * (size_t)nns_stride * 4);
    // ... and on the int8 copy before that (round 6): the submissions of at least AH_SEARCH_SCREEN8_MIN_QUERIES queries (65: past
    // the small ones' unit builder).  From 9 queries a call the leaf tiles are bound by the bytes of their rows (64 queries: 294 of
    // the call's 508 us) and the int8 rows do cut them to 249 — but the wider band of survivors costs the selection 65 -> 95 us
    // and the queries' digits a launch: 511 us against 502, measured, so the default leaves those calls on binary16; see ScreenSearch
    const bool screened8 = screened && allow8 && tun(TUN_SEARCH_SCREEN8) != 0 && !ix->search8_off.load(std::memory_order_relaxed) &&
                           (long long)nq > std::max(tun(TUN_SEARCH_SMALL_TILES_MAX_QUERIES), tun(TUN_SEARCH_SCREEN8_MIN_QUERIES) - 1) &&
                           ds->screen8_ready.load(std::memory_order_acquire);
    if (screened8) dev_bytes += pad(nq * (size_t)ds->pitch8 * 2) + pad(nq * sizeof(float4)) + pad(nq * (size_t)nns_stride * 4);
    // (a few queries a call: the list of (unit, slab) pairs k_units_small leaves for the tile launch)
    const size_t items_cap = tiles && (long long)nq <= tun(TUN_SEARCH_SMALL_UNITS_MAX_QUERIES) && tun(TUN_SEARCH_ITEM_LIST) != 0
                                 ? nq * ((size_t)nns_stride / kTileSmallSlab + 1) + kSmallVisits + 1
                                 : 0;
    if (tiles)
        dev_bytes += pad((size_t)visit_cap * sizeof(Visit)) * 2 + pad((size_t)visit_cap * sizeof(TileUnit)) +
                     pad((size_t)ix->n_nodes * 4 + 8) + 2 * pad((size_t)ix->n_nodes * 4) + pad((size_t)n_leaf_sums * 8) + pad(nq * 4) +
                     pad(items_cap * 4) + pad(nq * 4);
    void *const clean_status = ctx->clean_status;  // (ensure_device forgets it: see Context)
    AH_TRY(ctx->ensure_device(dev_bytes));
    const size_t pin_bytes = pad(nq * (size_t)ds->dims * 4) + pad(nq * 4) * 4 + pad(nq * sizeof(HostSeg2)) +
                             pad((size_t)max_tiles_bound * sizeof(HostTile2)) + pad(nq * k * 4) * 2 + 4096;
    AH_TRY(ctx->ensure_pinned(pin_bytes));
    uint8_t *dbase = reinterpret_cast<uint8_t *>(ctx->d_scratch);
    size_t doff = 0;
    auto dtake = [&](size_t bytes) {
        void *p = dbase + doff;
        doff += pad(bytes);
        return p;
    };
    uint8_t *pbase = reinterpret_cast<uint8_t *>(ctx->h_pinned);
    size_t poff = 0;
    auto ptake = [&](size_t bytes) {
        void *p = pbase + poff;
        poff += pad(bytes);
        return p;
    };
    float *d_qf32 = (float *)dtake(nq * (size_t)ds->dims * 4);
    uint32_t *d_qrows = (uint32_t *)dtake(nq * 4);
    uint8_t *d_qvecs = (uint8_t *)dtake(nq * qstride);
    float *d_qhdrs = (float *)dtake(nq * 8);
    uint32_t *d_nns = (uint32_t *)dtake(nq * (size_t)nns_stride * 4);
    float *d_dist = (float *)dtake(nq * (size_t)nns_stride * 4);
    uint32_t *d_counts = (uint32_t *)dtake(nq * 4);
    uint32_t *d_overflow = (uint32_t *)dtake(nq * 4);
    uint32_t *d_list = (uint32_t *)dtake(nq * 4);
    HostSeg2 *d_segs = (HostSeg2 *)dtake(nq * sizeof(HostSeg2));
    HostTile2 *d_tiles = (HostTile2 *)dtake((size_t)max_tiles_bound * sizeof(HostTile2));
    uint64_t *d_ka = (uint64_t *)dtake(nq * kstride * 8);
    uint64_t *d_kb = (uint64_t *)dtake(nq * kstride * 8);
    uint32_t *d_oi = (uint32_t *)dtake(nq * k * 4);
    float *d_od = (float *)dtake(nq * k * 4);
    uint32_t *d_err = (uint32_t *)dtake(SS_WORDS * 4);  // [error bits][SearchStatSlot counters]
    // (d_oi, d_od, d_err, d_unique lie back to back, and so do their pinned mirrors below: the tile path reads all four
    // back with ONE copy — a call of one query is a dozen launches and copies of ~5 us each)
    uint32_t *d_unique = (uint32_t *)dtake(nq * 4);
    uint32_t *d_inv = inv_bytes ? (uint32_t *)dtake(inv_bytes) : nullptr;
    Visit *d_visits = nullptr, *d_sorted = nullptr;
    TileUnit *d_units = nullptr;
    uint32_t *d_leaf_count = nullptr, *d_cursor = nullptr, *d_ustart = nullptr, *d_items = nullptr, *d_unit_counts = nullptr;
    uint2 *d_leaf_sums = nullptr;
    if (tiles) {
        d_visits = (Visit *)dtake((size_t)visit_cap * sizeof(Visit));
        d_sorted = (Visit *)dtake((size_t)visit_cap * sizeof(Visit));
        d_units = (TileUnit *)dtake((size_t)visit_cap * sizeof(TileUnit));
        d_leaf_count = (uint32_t *)dtake((size_t)ix->n_nodes * 4 + 8);  // + the number of visits, + the number of units
        d_cursor = (uint32_t *)dtake((size_t)ix->n_nodes * 4);
        d_ustart = (uint32_t *)dtake((size_t)ix->n_nodes * 4);
        d_leaf_sums = (uint2 *)dtake((size_t)n_leaf_sums * 8);
        if (items_cap) d_items = (uint32_t *)dtake(items_cap * 4);
        d_unit_counts = (uint32_t *)dtake(nq * 4);
    }
    ScreenSearch ss{};
    if (screened) {
        ss.rows16 = ds->d_rows_h16;
        ss.max_stats = make_float4(ds->screen_max[0], ds->screen_max[1], ds->screen_max[2], 0.0f);
        ss.aux = ds->metric == AH_COSINE ? (float *)dtake(nq * (size_t)nns_stride * 4) : nullptr;
        ss.hpitch = ds->hpitch;
        // accumulation-error factors as in the forest build (forest.hip: build_batch), 4x over the standard model
        ss.gamma_s = (float)(4.0 * (2.0 * (ds->hpitch / 16) + 8.0) * 5.9604645e-8);
        ss.gamma_r = (float)(4.0 * ((double)(ds->dims / 32) + 6.0 + 62.0) * 5.9604645e-8);
        ss.q16 = (const uint16_t *)dtake(nq * (size_t)ds->hpitch * 2);
        ss.qstats = (const float4 *)dtake(nq * sizeof(float4));
    }
    if (screened8) {
        ss.rows8 = ds->d_rows_i8;
        ss.row_scale8 = ds->d_scale8_rows;
        ss.dim_scale = ds->d_dim_scale;
        ss.pitch8 = ds->pitch8;
        ss.max8 = make_float4(ds->screen8_max[0], ds->screen8_max[1], ds->screen8_max[2], 0.0f);
        ss.q8 = (const int8_t *)dtake(nq * (size_t)ds->pitch8 * 2);
        ss.q8stats = (const float4 *)dtake(nq * sizeof(float4));
        ss.aux8 = (float *)dtake(nq * (size_t)nns_stride * 4);
    }
    float *h_q = (float *)ptake(nq * (size_t)ds->dims * 4);
    uint32_t *h_qrows = (uint32_t *)ptake(nq * 4);
    uint32_t *h_overflow = (uint32_t *)ptake(nq * 4);
    uint32_t *h_list = (uint32_t *)ptake(nq * 4);
    HostSeg2 *h_segs = (HostSeg2 *)ptake(nq * sizeof(HostSeg2));
    HostTile2 *h_tiles = (HostTile2 *)ptake((size_t)max_tiles_bound * sizeof(HostTile2));
    uint32_t *h_oi = (uint32_t *)ptake(nq * k * 4);
    float *h_od = (float *)ptake(nq * k * 4);
    uint32_t *h_err = (uint32_t *)ptake(SS_WORDS * 4);
    uint32_t *h_counts = (uint32_t *)ptake(nq * 4);
    ChunkStats cs;
    cs.s.chunks = 1;
    cs.s.queries = nq;
    if (d_filter_bits) cs.s.filtered_queries = nq;

    const DataView dv = ds->view();
    // 1. query leaves (src/reader.rs:46-51 by_item, :64-75 by_vector)
    // (a small submission: the kernel that prepares the query leaves reads the pinned staging buffer itself — 6 KB over the link
    // cost less than the copy engine's launch)
    const bool zero_copy_queries = queries && (long long)nq <= tun(TUN_SEARCH_SMALL_UNITS_MAX_QUERIES);
    // The small-submission kernels have hard capacities and, on the leaf-tile path, nothing behind them: k_descend_block holds 32
    // leaves per octet (trees t = octet mod 32), k_units_small kSmallVisits visits per call; past either the whole chunk is
    // redone the long way — correct, and more than twice the latency (round-5 advice: 128-d data at search_k = 10 000 opens
    // ~100 leaves per query, a 3-tree index more than 32 per tree).  So: an estimate of the leaves one query opens — search_k
    // items at the index's mean leaf size (what a filter keeps of it), a quarter more for leaves smaller than the mean — decides
    // on the host whether a call starts there at all (at most 8 estimated leaves per octet, 0.8 x kSmallVisits visits per call).  AH_SEARCH_SMALL_GATE=0: as before (the overflow tests force the
    // fall-backs with it).
    bool block_fits = true, small_units_fit = true;
    if (tun(TUN_SEARCH_SMALL_GATE) != 0 && ix->n_leaves) {
        const double mean_leaf = std::max(1.0, (double)ix->desc_len / ix->n_leaves * (d_filter_bits ? std::max(filter_share, 1e-3) : 1.0));
        const double est_leaves = 1.25 * (double)search_k / mean_leaf + 2.0;
        // (measured, round 6: 64 queries x ~18 estimated leaves per octet overflowed the block kernel — the best-first order does
        // not spread a query's leaves evenly over the trees; the benchmarked shapes sit near 1 per octet)
        block_fits = est_leaves / std::max(1u, std::min(ix->n_trees, 32u)) <= 8.0;
        small_units_fit = (double)nq * est_leaves <= 0.8 * kSmallVisits;
    }
    const bool block_ok = block_fits && (long long)nq <= tun(TUN_SEARCH_BLOCK_MAX_QUERIES);
    if (queries) {
        memcpy(h_q, queries, nq * (size_t)ds->dims * 4);
        if (!zero_copy_queries) AH_HIP(hipMemcpyAsync(d_qf32, h_q, nq * (size_t)ds->dims * 4, hipMemcpyHostToDevice, s));
    } else {
        memcpy(h_qrows, query_rows, nq * 4);
        AH_HIP(hipMemcpyAsync(d_qrows, h_qrows, nq * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_load_items_as_queries, dim3((unsigned)nq), dim3(64), 0, s, dv, d_qrows, d_qvecs, qstride, d_qhdrs);
    }
    // (the selection kernel of the previous small submission wiped its status block after copying it out: same place, same stream
    // — no memset node in front of this call's first kernel)
    if (!(clean_status == (void *)d_err && tun(TUN_SEARCH_STATUS_WIPE) != 0)) AH_HIP(hipMemsetAsync(d_err, 0, SS_WORDS * 4, s));
    // (by_vector leaves are prepared by the batch launcher below; the descent needs them first)
    SearchParams sp{};
    sp.stats = d_err;
    sp.nodes = ix->d_nodes;
    sp.roots = ix->d_roots;
    sp.n_trees = ix->n_trees;
    sp.desc = ix->d_desc;
    sp.filter_bits = d_filter_bits;
    sp.filter_len_bits = filter_len_bits;
    sp.search_k = search_k;
    sp.nns_stride = nns_stride;
    // (... and when the block descent is the first kernel to want the leaves, it prepares them itself)
    // (a small submission on the leaf-tile path makes the binary16 copies of its queries in the launch that places its visits)
    const bool small_units = tiles && small_units_fit && (long long)nq <= tun(TUN_SEARCH_SMALL_UNITS_MAX_QUERIES);
    // (small_units: otherwise k_queries_h16 below wants the leaves BEFORE the descent — until round 6 a call with the block
    // descent on and k_units_small off screened against copies of unprepared leaves and was saved by the fall-back only)
    const bool fuse_prepare = zero_copy_queries && tiles && small_units && wave_descent && !metric_is_bq_dev(ds->metric) &&
                              (!d_filter_bits || filter_share >= 0.35) && block_ok && tun(TUN_SEARCH_FUSED_PREPARE) != 0;
    if (queries && !fuse_prepare)
        AH_TRY(launch_prepare_queries_only(dv, zero_copy_queries ? h_q : d_qf32, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, s));
    if (screened && !small_units)
        hipLaunchKernelGGL(k_queries_h16, dim3((unsigned)nq), dim3(64), 0, s, d_qvecs, qstride, ds->dims, ds->hpitch,
                           const_cast<uint16_t *>(ss.q16), const_cast<float4 *>(ss.qstats));
    // 2. descent: one wave per query, then one octet per query for what that left, queue in LDS
    // (ah_search_batch decides: a filter that keeps under 5 % of the items makes a query pop more nodes than the queues of
    // a wave hold; under a filter the wave descent reads what the filter keeps of every leaf, computed once per submission)
    if (wave_descent && d_filter_bits) sp.leaf_kept = d_leaf_kept;
    uint32_t multi_launched = 0;  // blocks per query of k_descend_multi, when that was the descent
    bool passes_done = false;  // set by launch_wave: the block descent was the whole descent
    bool units_done = false;   // ... and it wrote the leaf tiles' work units as well (one query)
    bool items_made = false;   // k_units_small left the list of (unit, slab) pairs for the tile launch
    uint32_t units_per_query_used = 0;  // > 0: every query's descent wrote its own units (SingleQueryOut::per_query)
    auto launch_wave = [&](const VisitSink &sink) -> int {
        // a query pops about 1 / (kept share) as many nodes under a filter: start with the big queues (one query per CU
        // at a time) only then; otherwise they take what the small ones (four per CU) could not hold
        const bool small_first = !d_filter_bits || filter_share >= 0.35;
        // few queries: a block of 32 octets per query (one tree per octet: a third of the chain of dependent pops) while the
        // device has the room — arroy's own API is one query per call (src/reader.rs:46-75)
        static std::atomic<bool> lds_opt_in[64];  // once per device: the kernels that want more than 64 KiB of LDS
        // (+ qstride bytes each: the query leaf's copy in LDS; a leaf of more than 32 KiB is refused above)
        const size_t block_lds = block_descend_lds_bytes<32, 128, 32>() + qstride;
        const void *wave_big = reinterpret_cast<const void *>(k_descend_wave<1024, 128>);
        const void *block_fn = reinterpret_cast<const void *>(k_descend_block<32, 128, 32, 2>);
        const void *multi_fn = reinterpret_cast<const void *>(k_descend_multi<128, 1>);
        if (!lds_opt_in[ds->device & 63].load(std::memory_order_acquire)) {
            AH_HIP(hipFuncSetAttribute(multi_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(multi_descend_lds_bytes<128>() + (32u << 10))));
            AH_HIP(hipFuncSetAttribute(wave_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wave_lds_bytes(1024, 128) + (32u << 10))));
            AH_HIP(hipFuncSetAttribute(block_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(block_descend_lds_bytes<32, 128, 32>() + (32u << 10))));
            lds_opt_in[ds->device & 63].store(true, std::memory_order_release);
        }
        if (small_first && block_ok) {
            // On the leaf-tile path nothing runs behind the block kernel: what it cannot hold (rare: its capacities, equal keys
            // of two octets across the cut) raises bit 4 of *err and the submission is redone the long way, instead of every
            // small call paying two more launches for passes that find nothing to do.
            const bool last_pass = sink.visits != nullptr;
            // (one query: its units and its binary16 copy come out of the descent itself, see SingleQueryOut)
            // A handful of queries, more trees than one wave has octets: the trees of a query dealt over G blocks of one descent wave
            // each (k_descend_multi) — one query on G compute units.  Only as the last pass (what it cannot hold takes the long way).
            const uint32_t per_block = (uint32_t)std::min<long long>(8, std::max<long long>(1, tun(TUN_SEARCH_MULTI_TREES_PER_BLOCK)));
            const uint32_t multi_blocks = std::min<uint32_t>(kMultiMaxBlocks, (ix->n_trees + per_block - 1) / per_block);
            const bool multi = last_pass && tun(TUN_SEARCH_MULTI) != 0 && multi_blocks >= 2 &&
                               (long long)nq <= std::min<long long>(tun(TUN_SEARCH_MULTI_MAX_QUERIES), kMultiMaxQueries);
            // ... and a few queries whose descents write their own units, query by query (the flat tile launch reads them with
            // blockIdx.y = query): no launch that sorts a hundred visits of all queries by leaf.  Queries that open the same leaf
            // read its rows once each.
            const uint32_t units_per_query = (uint32_t)(visit_cap / std::max<size_t>(nq, 1));
            const bool own_units = multi && nq > 1 && screened && units_per_query >= kMultiCap && tun(TUN_SEARCH_MULTI_OWN_UNITS) != 0 &&
                                   (long long)nq <= tun(TUN_SEARCH_SMALL_TILES_MAX_QUERIES) && tun(TUN_SEARCH_FLAT_TILES) != 0 &&
                                   std::max(1u, (ix->max_desc + kTileSmallSlab - 1) / kTileSmallSlab) <= 65535u;
            SingleQueryOut single{};
            if (last_pass && (nq == 1 || own_units) && small_units && tun(TUN_SEARCH_SINGLE_FUSED) != 0) {
                single.units = d_units;
                single.sorted = d_sorted;
                single.n_units = nq == 1 ? d_err + SS_N_UNITS : d_unit_counts;
                single.per_query = nq == 1 ? 0u : units_per_query;
                units_per_query_used = single.per_query;
                if (screened)
                    single.h16 = QueriesH16{d_qvecs, qstride, ds->dims, ds->hpitch, const_cast<uint16_t *>(ss.q16), const_cast<float4 *>(ss.qstats)};
                units_done = true;
                // (k_leaf_tiles16<true> will be the tile launch: the condition of `small_tiles` below)
                single.ids_by_tiles = screened && !d_filter_bits && tun(TUN_SEARCH_MULTI_IDS_BY_TILES) != 0 &&
                                      (long long)nq <= tun(TUN_SEARCH_SMALL_TILES_MAX_QUERIES) &&
                                      std::max(1u, (ix->max_desc + kTileSmallSlab - 1) / kTileSmallSlab) <= 65535u;
            }
            const float *raw = (last_pass && fuse_prepare) ? (const float *)h_q : (const float *)nullptr;
            if (multi) {
                AH_TRY(ctx->ensure_multi(multi_ctl_bytes()));
                hipLaunchKernelGGL((k_descend_multi<128, 1>), dim3((unsigned)nq * multi_blocks), dim3(256), multi_descend_lds_bytes<128>() + qstride,
                                   s, ix->nv, sp, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow, sink, raw, d_qhdrs, single,
                                   reinterpret_cast<MultiCtl *>(ctx->d_multi), multi_blocks, tun(TUN_SEARCH_MULTI_TRACE) != 0 ? 1u : 0u);
                multi_launched = multi_blocks;
            } else
                hipLaunchKernelGGL((k_descend_block<32, 128, 32, 2>), dim3((unsigned)nq), dim3(256), block_lds, s, ix->nv, sp, (uint32_t)nq,
                                   d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow, sink, last_pass, raw, d_qhdrs, single);
            if (last_pass) {
                passes_done = true;
                return AH_OK;
            }
        } else if (small_first)
            hipLaunchKernelGGL((k_descend_wave<256, 64>), dim3((unsigned)nq), dim3(64), wave_lds_bytes(256, 64) + qstride, s, ix->nv, sp,
                               (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow, sink, false);
        hipLaunchKernelGGL((k_descend_wave<1024, 128>), dim3((unsigned)nq), dim3(64), wave_lds_bytes(1024, 128) + qstride, s, ix->nv, sp,
                           (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow, sink, small_first);
        return AH_OK;
    };
    const size_t heap_lds = (size_t)8 * kHeapLds * 8;
    {
        static std::atomic<bool> heap_opt_in[64];  // once per device (a runtime call per search is microseconds of a 0.2 ms call)
        if (!heap_opt_in[ds->device & 63].load(std::memory_order_acquire)) {
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_descend<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)heap_lds));
            heap_opt_in[ds->device & 63].store(true, std::memory_order_release);
        }
    }
    if (tiles) {  // 2'. the leaf-tile path: descent with its visits recorded, no host round trip before the results
        // (the visit and unit counters live in spare words of the status block: one memset clears them with it)
        uint32_t *d_total = d_err + SS_VISIT_TOTAL, *d_n_units = d_err + SS_N_UNITS;
        // a small submission places its visits with one block (k_units_small) and needs no counter per node
        if (!small_units) AH_HIP(hipMemsetAsync(d_leaf_count, 0, (size_t)ix->n_nodes * 4, s));
        const VisitSink sink{d_visits, d_total, visit_cap, small_units ? nullptr : d_leaf_count, d_err};
        if (wave_descent) AH_TRY(launch_wave(sink));
        if (!passes_done)
            hipLaunchKernelGGL((k_descend<false>), dim3((unsigned)((nq + 7) / 8)), dim3(64), heap_lds, s, ix->nv, sp,
                               (const uint32_t *)nullptr, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow,
                               (uint64_t *)nullptr, 0u, sink, wave_descent);
        if (units_done) {
            // (nothing to place)
        } else if (small_units) {
            const QueriesH16 h16{d_qvecs, qstride, ds->dims, ds->hpitch, const_cast<uint16_t *>(ss.q16), const_cast<float4 *>(ss.qstats)};
            hipLaunchKernelGGL(k_units_small, dim3(1u + (screened ? (unsigned)nq : 0u)), dim3(256), 0, s, d_visits, d_total, visit_cap,
                               d_sorted, d_units, d_n_units, d_err, h16, d_items, (uint32_t)items_cap);
            items_made = d_items != nullptr;
        } else {
            hipLaunchKernelGGL(k_leaf_scan_block, dim3(n_leaf_sums), dim3(256), 0, s, d_leaf_count, ix->n_nodes, d_cursor, d_ustart,
                               d_leaf_sums);
            hipLaunchKernelGGL(k_leaf_scan_sums, dim3(1), dim3(256), 0, s, d_leaf_sums, n_leaf_sums, d_n_units);
            hipLaunchKernelGGL(k_leaf_scan_add, dim3(n_leaf_sums), dim3(256), 0, s, d_leaf_count, ix->n_nodes, d_cursor, d_ustart,
                               d_leaf_sums, d_units, d_err);
            hipLaunchKernelGGL(k_visit_scatter, dim3(256), dim3(256), 0, s, d_visits, d_total, visit_cap, d_cursor, d_sorted);
        }
        const unsigned tile_slabs = std::max(1u, (ix->max_desc + kTileSlab - 1) / kTileSlab);
#define AH_TILES(M)                                                                                                        \
    do {                                                                                                                   \
        AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_leaf_tiles<M>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)(4 * kRingBytesPerWave)));                                                         \
        hipLaunchKernelGGL((k_leaf_tiles<M>), dim3(2048, tile_slabs), dim3(256), 4 * kRingBytesPerWave, s, dv, d_nns, d_sorted, \
                           d_units, d_n_units, d_qvecs, qstride, d_qhdrs, d_dist, nns_stride, d_err);                       \
    } while (0)
        if (screened8) {
            // the leaves few queries reached on the int8 copy (a quarter of the f32 bytes, a wider band of survivors), the leaves
            // many queries share on the binary16 copy as before; the selection takes every candidate's bound from its own stage
            // (the queries' int8 digits here, behind the descent: a small submission's query leaves are prepared BY its descent)
            hipLaunchKernelGGL(k_queries_i8, dim3((unsigned)nq), dim3(64), 0, s, d_qvecs, qstride, ds->dims, ss);
            const uint32_t max_vis8 = (uint32_t)std::max<long long>(1, tun(TUN_SEARCH_SCREEN8_MAX_VISITS));
            hipLaunchKernelGGL(k_leaf_tiles8, dim3(2048, tile_slabs), dim3(256), 0, s, dv, ss, d_nns, d_sorted, d_units, d_n_units, d_dist,
                               nns_stride, d_err, max_vis8);
            hipLaunchKernelGGL((k_leaf_tiles16<false>), dim3(2048, tile_slabs), dim3(256), 0, s, dv, ss, d_nns, d_sorted, d_units,
                               d_n_units, d_dist, nns_stride, d_err, max_vis8 + 1);
        } else if (screened) {  // the candidates on the binary16 copies first: half the bytes (see k_search_select_screened)
            // a small submission: slabs of 64 rows (a dozen leaves then fill 150 CUs instead of 50) and a grid that does not
            // dispatch 24 000 blocks to find 12 units
            const unsigned small_slabs = std::max(1u, (ix->max_desc + kTileSmallSlab - 1) / kTileSmallSlab);
            const bool small_tiles = (long long)nq <= tun(TUN_SEARCH_SMALL_TILES_MAX_QUERIES) && small_slabs <= 65535u;
            // (one query whose units came out of its descent: a flat list of (leaf, slab) items, see the kernel)
            const bool flat_tiles = small_tiles && units_done && (nq == 1 || units_per_query_used) && visit_cap >= 128u &&
                                    tun(TUN_SEARCH_FLAT_TILES) != 0;
            uint32_t *tile_trace =
                multi_launched && tun(TUN_SEARCH_MULTI_TRACE) != 0 ? &reinterpret_cast<MultiCtl *>(ctx->d_multi)->tile_trace[0] : nullptr;
            if (flat_tiles)
                hipLaunchKernelGGL((k_leaf_tiles16<true>), dim3(nns_stride / kTileSmallSlab + 129u, (unsigned)nq), dim3(256), 0, s, dv, ss, d_nns,
                                   d_sorted, d_units, nq == 1 ? d_n_units : d_unit_counts, d_dist, nns_stride, d_err, 0u, ix->d_desc, 0u, tile_trace,
                                   1u, (const uint32_t *)nullptr, units_per_query_used);
            else if (small_tiles && items_made)  // one block per listed (unit, slab) pair (grid.y = 1: an overflowed list is walked)
                hipLaunchKernelGGL((k_leaf_tiles16<true>), dim3((unsigned)std::min<size_t>(items_cap, 2048)), dim3(256), 0, s, dv, ss, d_nns,
                                   d_sorted, d_units, d_n_units, d_dist, nns_stride, d_err, 0u, ix->d_desc, 0u, tile_trace, 0u, d_items);
            else if (small_tiles)
                hipLaunchKernelGGL((k_leaf_tiles16<true>), dim3(std::min<unsigned>(2048u, 32u * (unsigned)nq), small_slabs), dim3(256), 0, s,
                                   dv, ss, d_nns, d_sorted, d_units, d_n_units, d_dist, nns_stride, d_err, 0u, ix->d_desc,
                                   visit_cap >= std::min<unsigned>(2048u, 32u * (unsigned)nq) ? 1u : 0u, tile_trace);
            else
                hipLaunchKernelGGL((k_leaf_tiles16<false>), dim3(2048, tile_slabs), dim3(256), 0, s, dv, ss, d_nns, d_sorted, d_units,
                                   d_n_units, d_dist, nns_stride, d_err);
        } else {
            switch (ds->metric) {
            case AH_EUCLIDEAN: AH_TILES(AH_EUCLIDEAN); break;
            case AH_COSINE: AH_TILES(AH_COSINE); break;
            default: AH_TILES(AH_DOT_PRODUCT); break;
            }
        }
#undef AH_TILES
        if (tun(TUN_DEBUG)) {
            AH_HIP(hipStreamSynchronize(s));
            uint32_t nv = 0, np = 0;
            AH_HIP(hipMemcpy(&nv, d_total, 4, hipMemcpyDeviceToHost));
            AH_HIP(hipMemcpy(&np, d_n_units, 4, hipMemcpyDeviceToHost));
            std::vector<Visit> hv(std::min(nv, visit_cap));
            AH_HIP(hipMemcpy(hv.data(), d_visits, hv.size() * sizeof(Visit), hipMemcpyDeviceToHost));
            std::vector<uint32_t> per(ix->n_nodes, 0);
            for (auto &v : hv) per[v.node]++;
            uint32_t hist[9] = {0}, leaves = 0;
            for (uint32_t c : per)
                if (c) { leaves++; hist[std::min(c, 8u)]++; }
            fprintf(stderr, "[ah] search tiles: %u visits, %u leaves, %u units; visits per leaf 1..8+: %u %u %u %u %u %u %u %u\n", nv,
                    leaves, np, hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
        }
        // (after the tiles: they read the leaves' ids from the candidate buffers)
        // a small submission: nns.dedup() inside the selection kernel (FLAG), when the bitmap fits beside its other LDS
        const size_t sel_lds = (size_t)ds->pitch * 4;  // the query leaf in f32
        const size_t fused_lds = sel_lds + (size_t)bitmap_words * 4;
        const bool fused_flag = small_units && screened && bitmap_fits && nns_stride <= 16384 && fused_lds + (26u << 10) <= (160u << 10) &&
                                tun(TUN_SEARCH_FUSED_FLAG) != 0;
        if (fused_flag) {
            *h_err = 0xFFFFFFFFu;  // (overwritten by the selection kernel's last block: a launch that never ran reads as a failure)
            static std::atomic<uint32_t> fused_opt_in[64][2];  // [device][metric]: largest dynamic LDS opted in for
            const int mi = ds->metric == AH_COSINE ? 0 : 1;
            if (fused_opt_in[ds->device & 63][mi].load(std::memory_order_acquire) < fused_lds) {
                const void *fn = mi == 0 ? reinterpret_cast<const void *>(k_search_select_screened<AH_COSINE, true>)
                                         : reinterpret_cast<const void *>(k_search_select_screened<AH_DOT_PRODUCT, true>);
                AH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
                fused_opt_in[ds->device & 63][mi].store((uint32_t)fused_lds, std::memory_order_release);
            }
        } else if (bitmap_fits) {
            const size_t sh = (size_t)bitmap_words * 4;
            static std::atomic<uint32_t> flag_lds[64];  // largest bitmap this device's kernel has been opted in for
            if (sh > 48 * 1024 && flag_lds[ds->device & 63].load(std::memory_order_acquire) < sh) {
                AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_flag_duplicates),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
                flag_lds[ds->device & 63].store((uint32_t)sh, std::memory_order_release);
            }
            hipLaunchKernelGGL(k_flag_duplicates, dim3((unsigned)nq), dim3(1024), sh, s, d_nns, nns_stride, d_counts, bitmap_words,
                               max_id + 1, d_unique, d_err);
        } else {
            static std::atomic<bool> hash_opt_in[64];
            if (!hash_opt_in[ds->device & 63].load(std::memory_order_acquire)) {
                AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_flag_duplicates_hash),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kHashSlots * 4)));
                hash_opt_in[ds->device & 63].store(true, std::memory_order_release);
            }
            hipLaunchKernelGGL(k_flag_duplicates_hash, dim3((unsigned)nq), dim3(1024), kHashSlots * 4, s, d_nns, nns_stride, d_counts,
                               d_unique, d_err);
        }
        if (screened && sel_lds > 32 * 1024 && !fused_flag) {
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_select_screened<AH_COSINE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_select_screened<AH_DOT_PRODUCT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
        }
        uint32_t *sel_trace = multi_launched && tun(TUN_SEARCH_MULTI_TRACE) != 0 ? &reinterpret_cast<MultiCtl *>(ctx->d_multi)->sel_trace[0] : nullptr;
        if (fused_flag && ds->metric == AH_COSINE)
            hipLaunchKernelGGL((k_search_select_screened<AH_COSINE, true>), dim3((unsigned)nq), dim3(1024), fused_lds, s, dv, ss, d_nns, d_dist,
                               nns_stride, d_counts, d_unique, (uint32_t)k, d_qvecs, qstride, d_qhdrs, h_oi, h_od, d_err, (const PairSeg *)nullptr,
                               bitmap_words, max_id + 1, h_counts, h_err, sel_trace);
        else if (fused_flag)
            hipLaunchKernelGGL((k_search_select_screened<AH_DOT_PRODUCT, true>), dim3((unsigned)nq), dim3(1024), fused_lds, s, dv, ss, d_nns,
                               d_dist, nns_stride, d_counts, d_unique, (uint32_t)k, d_qvecs, qstride, d_qhdrs, h_oi, h_od, d_err,
                               (const PairSeg *)nullptr, bitmap_words, max_id + 1, h_counts, h_err, sel_trace);
        else if (screened && ds->metric == AH_COSINE)
            hipLaunchKernelGGL((k_search_select_screened<AH_COSINE>), dim3((unsigned)nq), dim3(1024), sel_lds, s, dv, ss, d_nns, d_dist,
                               nns_stride, d_counts, d_unique, (uint32_t)k, d_qvecs, qstride, d_qhdrs, d_oi, d_od, d_err, (const PairSeg *)nullptr);
        else if (screened)
            hipLaunchKernelGGL((k_search_select_screened<AH_DOT_PRODUCT>), dim3((unsigned)nq), dim3(1024), sel_lds, s, dv, ss, d_nns, d_dist,
                               nns_stride, d_counts, d_unique, (uint32_t)k, d_qvecs, qstride, d_qhdrs, d_oi, d_od, d_err, (const PairSeg *)nullptr);
        else
            hipLaunchKernelGGL(k_search_select, dim3((unsigned)nq), dim3(kSelectThreads), 0, s, dv, d_nns, d_dist, nns_stride, d_counts,
                               d_unique, (uint32_t)k, d_oi, d_od, d_err);
        // a launch the runtime rejected (dynamic LDS beyond the limit, another architecture) would leave *err = 0 over
        // uninitialised results: such a submission takes the sorted path as well
        const hipError_t launch_err = hipGetLastError();
        // ids, distances, status words and counts: one copy (the four buffers are carved back to back on both sides)
        // (fused_flag: the selection kernel wrote ids, distances, counts and status into the pinned buffers itself)
        if (!fused_flag) AH_HIP(hipMemcpyAsync(h_oi, d_oi, 2 * pad(nq * k * 4) + pad(SS_WORDS * 4) + nq * 4, hipMemcpyDeviceToHost, s));
        // fused_flag: the selection's last block writes the status into pinned memory as the last thing the submission does (word 0
        // after everything else, system-scope release).  Polling that word returns as soon as it lands; the runtime's own wait adds
        // the end-of-kernel bookkeeping and its wake-up (AH_SEARCH_SPIN_WAIT=0: hipStreamSynchronize as before; it is also what
        // happens after 2 ms without an answer — a launch that never ran).  Everything queued later runs behind this submission in
        // stream order, so its scratch is safe to reuse.
        bool answered = false;
        if (fused_flag && launch_err == hipSuccess && tun(TUN_SEARCH_SPIN_WAIT) != 0) {
            const auto t_spin = std::chrono::steady_clock::now();
            for (uint32_t spin = 0;; spin++) {
                if (__atomic_load_n(h_err, __ATOMIC_ACQUIRE) != 0xFFFFFFFFu) {
                    answered = true;
                    break;
                }
                __builtin_ia32_pause();
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) break;
            }
        }
        if (!answered) AH_HIP(hipStreamSynchronize(s));
        if (multi_launched && tun(TUN_SEARCH_MULTI_TRACE) != 0) {  // where the blocks of query 0 spent their time (10 ns ticks -> us)
            std::vector<uint32_t> tr((size_t)kMultiMaxBlocks * 8);
            AH_HIP(hipMemcpy(tr.data(), &reinterpret_cast<MultiCtl *>(ctx->d_multi)->trace[0][0], tr.size() * 4, hipMemcpyDeviceToHost));
            for (uint32_t b = 0; b < multi_launched; b++)
                fprintf(stderr, "[ah] multi block %u: query in LDS %.2f us, descent left %.2f, counted %.2f%s\n", b, tr[b * 8] * 0.01, tr[b * 8 + 1] * 0.01,
                        tr[b * 8 + 2] * 0.01, tr[b * 8 + 5] > tr[b * 8 + 2] ? "  <- last" : "");
            for (uint32_t b = 0; b < multi_launched; b++)
                if (tr[b * 8 + 5] > tr[b * 8 + 2])
                    fprintf(stderr, "[ah]   (stale unless marked) last block %u: lists gathered %.2f us, ordered %.2f, ids copied %.2f (%u leaves known, %u taken)\n", b,
                            tr[b * 8 + 3] * 0.01, tr[b * 8 + 4] * 0.01, tr[b * 8 + 5] * 0.01, tr[b * 8 + 6], tr[b * 8 + 7]);
        }
        if (multi_launched && tun(TUN_SEARCH_MULTI_TRACE) != 0) {
            uint32_t st[16];
            AH_HIP(hipMemcpy(st, &reinterpret_cast<MultiCtl *>(ctx->d_multi)->sel_trace[0], sizeof(st), hipMemcpyDeviceToHost));
            uint32_t tt[8];
            AH_HIP(hipMemcpy(tt, &reinterpret_cast<MultiCtl *>(ctx->d_multi)->tile_trace[0], sizeof(tt), hipMemcpyDeviceToHost));
            AH_HIP(hipMemset(&reinterpret_cast<MultiCtl *>(ctx->d_multi)->tile_trace[0], 0, sizeof(tt)));
            fprintf(stderr, "[ah] tiles (latest block since ITS start): unit + visit known %.2f us, slab done %.2f us\n", tt[0] * 0.01, tt[1] * 0.01);
            fprintf(stderr, "[ah] selection: leaf + tables in LDS %.2f us, candidates in registers %.2f, duplicates flagged %.2f, key range %.2f, "
                    "k-th bin %.2f, survivors listed %.2f (%u), f32 distances %.2f, ranked + written %.2f, status written %.2f\n", st[0] * 0.01,
                    st[1] * 0.01, st[2] * 0.01, st[3] * 0.01, st[4] * 0.01, st[5] * 0.01, st[9], st[6] * 0.01, st[7] * 0.01, st[8] * 0.01);
        }
        AH_REQUIRE((*h_err & 1u) == 0 || *h_err == 0xFFFFFFFFu, AH_ERR_MISSING_ITEM, "a descendant id does not exist in the dataset");
        if ((*h_err & ~1u) == 0 && launch_err == hipSuccess) {
            if (fused_flag) ctx->clean_status = d_err;  // (its selection kernel ran to its end and wiped the block behind itself)
            for (size_t q = 0; q < nq; q++) out_counts[q] = (uint32_t)std::min<size_t>(k, h_counts[q]);
            memcpy(out_ids, h_oi, nq * k * 4);
            memcpy(out_dists, h_od, nq * k * 4);
            cs.device_words(h_err);
            cs.s.rerank_tiles = nq;
            cs.s.rerank_screened8 = screened8 ? h_err[SS_SCREENED] : 0;
            (bitmap_fits ? cs.s.dedup_flag_bitmap : cs.s.dedup_flag_hash) = nq;
            cs.commit(ix);
            return AH_OK;
        }
        if (screened8 && launch_err == hipSuccess && (*h_err & ~1u) == 8u) {
            // more survivors than the selection holds on the int8 stage (candidates closer together than its error bound): the
            // same sub-batch once more with the binary16 rows first — not the long way
            if (ix->search8_fails.fetch_add(1, std::memory_order_relaxed) + 1 >= 8) ix->search8_off.store(true, std::memory_order_relaxed);
            {
                std::lock_guard<std::mutex> lk(ix->stats_mu);
                ix->stats.screen8_retried_chunks += 1;
            }
            return search_chunk(ix, ctx, queries, query_rows, nq, count, search_k, nns_stride, d_filter_bits, filter_len_bits, filter_share,
                                wave_descent, d_leaf_kept, out_ids, out_dists, out_counts, false);
        }
        // a case the tiles do not reproduce (bits 2..5 of *err, see k_search_select / VisitSink): redo it the long way
        cs.s.fallback_chunks = 1;
        cs.s.fallback_non_finite = (*h_err & 4u) ? 1 : 0;
        cs.s.fallback_select = (*h_err & 8u) ? 1 : 0;
        cs.s.fallback_queue = (*h_err & 16u) ? 1 : 0;
        cs.s.fallback_visits = (*h_err & 32u) ? 1 : 0;
        cs.s.fallback_launch = launch_err != hipSuccess ? 1 : 0;
        AH_HIP(hipMemsetAsync(d_err, 0, SS_WORDS * 4, s));
        // (the block descent was the only writer of the query leaves under fuse_prepare: a launch the runtime rejected has left them
        // unprepared — the passes below read them; preparing them again when it did run changes nothing)
        if (queries && fuse_prepare) AH_TRY(launch_prepare_queries_only(dv, h_q, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, s));
    }
    if (wave_descent) AH_TRY(launch_wave(VisitSink{}));
    hipLaunchKernelGGL((k_descend<false>), dim3((unsigned)((nq + 7) / 8)), dim3(64), heap_lds, s, ix->nv, sp,
                       (const uint32_t *)nullptr, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow,
                       (uint64_t *)nullptr, 0u, VisitSink{}, wave_descent);
    AH_HIP(hipMemcpyAsync(h_overflow, d_overflow, nq * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipStreamSynchronize(s));
    uint32_t n_over = 0;
    for (size_t q = 0; q < nq; q++)
        if (h_overflow[q]) h_list[n_over++] = (uint32_t)q;
    if (n_over) {  // 2b. the rare big queues: re-run with the queue in global memory (hard capacity bound)
        DevMem heap;
        AH_HIP(dev_malloc(&heap.p, (size_t)n_over * heap_cap * 8));
        AH_HIP(hipMemcpyAsync(d_list, h_list, n_over * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((k_descend<true>), dim3((n_over + 7) / 8), dim3(64), 0, s, ix->nv, sp, (const uint32_t *)d_list,
                           n_over, d_qvecs, qstride, d_qhdrs, d_nns, d_counts, d_overflow, heap.as<uint64_t>(), heap_cap,
                           VisitSink{}, false);
        AH_HIP(hipStreamSynchronize(s));
    }
    // 3. sort + dedup
    uint32_t max_nn = 0;
    const bool by_bitmap = bitmap_fits;
    if (!by_bitmap) {
        AH_HIP(hipMemcpyAsync(h_counts, d_counts, nq * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipStreamSynchronize(s));
        for (size_t q = 0; q < nq; q++) max_nn = std::max(max_nn, h_counts[q]);
    }
    if (by_bitmap) {
        const size_t sh = (size_t)bitmap_words * 4;
        if (sh > 48 * 1024)
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dedup_bitmap_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_dedup_bitmap_lds, dim3((unsigned)nq), dim3(1024), sh, s, d_nns, nns_stride, d_counts,
                           bitmap_words, max_id + 1, d_err);
        cs.s.dedup_sorted_bitmap = nq;
    } else if (max_nn <= kSortLds) {
        cs.s.dedup_sort_lds = nq;
        uint32_t np2 = 2;
        while (np2 < max_nn) np2 <<= 1;
        const size_t sh = (size_t)np2 * 4;
        if (sh > 48 * 1024)
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_dedup_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_sort_dedup_lds, dim3((unsigned)nq), dim3(256), sh, s, d_nns, nns_stride, d_counts);
    } else {
        uint32_t np2 = 2;
        while (np2 < max_nn) np2 <<= 1;
        cs.s.dedup_sort_global = nq;
        // the stride was sized to the next power of two by the caller when this path is possible
        AH_REQUIRE(np2 <= nns_stride, AH_ERR_DEVICE, "internal: candidate stride %u < %u", nns_stride, np2);
        hipLaunchKernelGGL(k_pad_ids, dim3((np2 + 255) / 256, (unsigned)nq), dim3(256), 0, s, d_nns, nns_stride, d_counts, np2);
        for (uint32_t size = 2; size <= np2; size <<= 1)
            for (uint32_t str = size >> 1; str > 0; str >>= 1)
                hipLaunchKernelGGL(k_bitonic_ids, dim3(((np2 >> 1) + 255) / 256, (unsigned)nq), dim3(256), 0, s, d_nns,
                                   nns_stride, np2, size, str);
        hipLaunchKernelGGL(k_dedup_sorted, dim3((unsigned)nq), dim3(256), 0, s, d_nns, nns_stride, d_counts);
    }
    AH_HIP(hipMemcpyAsync(h_counts, d_counts, nq * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipStreamSynchronize(s));
    // 4. re-rank + top-k + normalized distances (batch.hip), candidates already resident
    uint32_t n_tiles = 0, max_n = 0, max_rounds = 0;
    uint64_t n_candidates = 0;
    const uint32_t tc = batch_tile_candidates();
    for (size_t q = 0; q < nq; q++) {
        const uint32_t n = h_counts[q];
        n_candidates += n;
        h_segs[q] = HostSeg2{(uint64_t)q * nns_stride, n, (uint32_t)std::min<size_t>(k, n)};
        out_counts[q] = h_segs[q].k;
        max_n = std::max(max_n, n);
        if (big_k) continue;
        max_rounds = std::max(max_rounds, batch_rounds(n, h_segs[q].k));
        for (uint32_t f = 0; f < n; f += tc) h_tiles[n_tiles++] = HostTile2{(uint32_t)q, f};
    }
    if (big_k) {
        // d_ka (2 x nq x kstride x 8 bytes >= the single-query scratch) is free: the batch path is not used
        AH_HIP(hipMemsetAsync(d_oi, 0xFF, nq * k * 4, s));  // padding: id 0xFFFFFFFF / NaN
        AH_HIP(hipMemsetAsync(d_od, 0xFF, nq * k * 4, s));
        for (size_t q = 0; q < nq; q++) {
            const uint32_t n = h_counts[q], kk = h_segs[q].k;
            if (kk == 0) continue;
            const uint32_t *ids_q = d_nns + q * (size_t)nns_stride;
            float *dist_q = d_dist + q * (size_t)nns_stride;
            AH_TRY(launch_distances(dv, d_qvecs + q * qstride, d_qhdrs + 2 * q, ids_q, n, dist_q, d_err, s));
            AH_TRY(launch_topk(dv, dist_q, ids_q, n, kk, d_ka, d_oi + q * k, d_od + q * k, s));
        }
    } else {
        AH_HIP(hipMemcpyAsync(d_segs, h_segs, nq * sizeof(HostSeg2), hipMemcpyHostToDevice, s));
        if (n_tiles) AH_HIP(hipMemcpyAsync(d_tiles, h_tiles, (size_t)n_tiles * sizeof(HostTile2), hipMemcpyHostToDevice, s));
        AH_TRY(launch_rerank_batch_prepared(dv, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, n_tiles, d_nns, d_dist,
                                            d_ka, d_kb, kstride, max_n, (uint32_t)k, max_rounds, d_oi, d_od, d_err, s,
                                            n_candidates, d_inv));
    }
    AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipGetLastError());
    AH_HIP(hipMemcpyAsync(h_err, d_err, SS_WORDS * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipStreamSynchronize(s));
    AH_REQUIRE((*h_err & 1u) == 0, AH_ERR_MISSING_ITEM, "a descendant id does not exist in the dataset");
    memcpy(out_ids, h_oi, nq * k * 4);
    memcpy(out_dists, h_od, nq * k * 4);
    cs.device_words(h_err);
    cs.s.rerank_sorted = nq;
    cs.commit(ix);
    return AH_OK;
}

// Route `n` items (already present in the index's dataset) down every tree of the index.
// out_leaf[t * n + i] = forest-local index of the Descendants node item i reaches in tree t.
int ah_route_items(ah_index *ix, const uint32_t *item_ids, size_t n, const uint64_t *tree_seeds, uint32_t *out_leaf) {
    AH_GUARDED("ah_route_items")
    AH_REQUIRE(ix && ix->ds, AH_ERR_INVALID_ARGUMENT, "index is NULL");
    AH_REQUIRE((item_ids && out_leaf && tree_seeds) || n == 0, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n == 0 || ix->n_trees == 0) return AH_OK;
    ah_dataset *ds = ix->ds;
    AH_HIP(hipSetDevice(ds->device));
    ContextLease lease(ds);
    AH_REQUIRE(lease.c, AH_ERR_DEVICE, "cannot create a HIP stream");
    Context *ctx = lease.c;
    const size_t pairs = n * (size_t)ix->n_trees;
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    AH_TRY(ctx->ensure_device(pad(n * 4) + pad(ix->n_trees * 8) + pad(pairs * 4) + 1024));
    AH_TRY(ctx->ensure_pinned(pad(n * 4) + pad(ix->n_trees * 8) + pad(pairs * 4) + 1024));
    uint8_t *d = reinterpret_cast<uint8_t *>(ctx->d_scratch), *h = reinterpret_cast<uint8_t *>(ctx->h_pinned);
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(d);
    uint64_t *d_seeds = reinterpret_cast<uint64_t *>(d + pad(n * 4));
    uint32_t *d_leaf = reinterpret_cast<uint32_t *>(d + pad(n * 4) + pad(ix->n_trees * 8));
    uint32_t *d_err = reinterpret_cast<uint32_t *>(d + pad(n * 4) + pad(ix->n_trees * 8) + pad(pairs * 4));
    uint32_t *h_ids = reinterpret_cast<uint32_t *>(h);
    uint64_t *h_seeds = reinterpret_cast<uint64_t *>(h + pad(n * 4));
    uint32_t *h_leaf = reinterpret_cast<uint32_t *>(h + pad(n * 4) + pad(ix->n_trees * 8));
    uint32_t *h_err = reinterpret_cast<uint32_t *>(h + pad(n * 4) + pad(ix->n_trees * 8) + pad(pairs * 4));
    memcpy(h_ids, item_ids, n * 4);
    memcpy(h_seeds, tree_seeds, (size_t)ix->n_trees * 8);
    hipStream_t s = ctx->stream;
    AH_HIP(hipMemcpyAsync(d_ids, h_ids, n * 4, hipMemcpyHostToDevice, s));
    AH_HIP(hipMemcpyAsync(d_seeds, h_seeds, (size_t)ix->n_trees * 8, hipMemcpyHostToDevice, s));
    AH_HIP(hipMemsetAsync(d_err, 0, 4, s));
    const unsigned grid = (unsigned)((pairs * 8 + 255) / 256);
    hipLaunchKernelGGL(k_route_items, dim3(grid), dim3(256), 0, s, ds->view(), ix->nv, ix->d_nodes, ix->d_roots, ix->n_trees,
                       d_ids, (uint64_t)n, d_seeds, d_leaf, d_err);
    AH_HIP(hipGetLastError());
    AH_HIP(hipMemcpyAsync(h_leaf, d_leaf, pairs * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipMemcpyAsync(h_err, d_err, 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipStreamSynchronize(s));
    AH_REQUIRE((*h_err & 1u) == 0, AH_ERR_MISSING_ITEM, "an item to route does not exist in the dataset");
    memcpy(out_leaf, h_leaf, pairs * 4);
    return AH_OK;
    AH_GUARDED_END
}

// `QueryBuilder::{by_vector, by_item}` for a batch (src/reader.rs:46-75, 317-401).
//   queries      nq x dims f32 (by_vector), or NULL with query_items = nq item ids (by_item)
//   search_k     0 = count * n_trees (reader.rs:330); oversampling 0 = D::DEFAULT_OVERSAMPLING (1, or 3 for 1-bit)
//   filter       optional ascending candidate ids (`QueryBuilder::candidates`)
// Outputs are nq x count, short lists padded with id 0xFFFFFFFF / NaN; out_counts[q] = results of query q.
int ah_search_batch(ah_index *ix, const float *queries, const uint32_t *query_items, size_t nq, size_t count,
                    size_t search_k, size_t oversampling, const uint32_t *filter_sorted, size_t n_filter, int have_filter,
                    uint32_t *out_ids, float *out_distances, uint32_t *out_counts) {
    AH_GUARDED("ah_search_batch")
    AH_REQUIRE(ix && ix->ds, AH_ERR_INVALID_ARGUMENT, "index is NULL");
    ah_dataset *ds = ix->ds;
    AH_REQUIRE((queries != nullptr) != (query_items != nullptr), AH_ERR_INVALID_ARGUMENT,
               "exactly one of queries / query_items must be given");
    AH_REQUIRE(out_ids && out_distances && out_counts, AH_ERR_INVALID_ARGUMENT, "NULL output");
    if (nq == 0) return AH_OK;
    AH_REQUIRE(count > 0 && count < 0x7FFFFFFFull, AH_ERR_INVALID_ARGUMENT, "count must be > 0");
    AH_HIP(hipSetDevice(ds->device));
    for (size_t i = 0; i < nq * count; i++) {
        out_ids[i] = 0xFFFFFFFFu;
        const uint32_t nan_bits = 0xFFFFFFFFu;
        memcpy(&out_distances[i], &nan_bits, 4);
    }
    for (size_t q = 0; q < nq; q++) out_counts[q] = 0;
    AH_REQUIRE(!have_filter || n_filter == 0 || filter_sorted, AH_ERR_INVALID_ARGUMENT, "filter_sorted is NULL");
    if (ds->n == 0 || ix->n_trees == 0) return AH_OK;  // reader.rs:323-325
    // search_k: reader.rs:330-335; nns can never exceed the blob (every Descendants node is popped at most once)
    unsigned __int128 sk = search_k ? (unsigned __int128)search_k : (unsigned __int128)count * ix->n_trees;
    sk *= oversampling ? oversampling : (metric_is_bq(ds->metric) ? 3u : 1u);
    const uint64_t sk_eff = (uint64_t)std::min<unsigned __int128>(sk, (unsigned __int128)ix->desc_len);
    uint64_t stride = std::min<uint64_t>(sk_eff + ix->max_desc, ix->desc_len);
    if (stride > kSortLds) {  // the global sort path pads to a power of two
        uint64_t p = 2;
        while (p < stride) p <<= 1;
        stride = p;
    }
    AH_REQUIRE(stride < 0x7FFFFFFFull, AH_ERR_INVALID_ARGUMENT, "search_k too large");
    std::vector<uint32_t> rows;
    if (query_items) {
        rows.resize(nq);
        for (size_t q = 0; q < nq; q++) {
            if (ds->identity_ids) {
                AH_REQUIRE(query_items[q] < ds->n, AH_ERR_MISSING_ITEM, "item %u does not exist", query_items[q]);
                rows[q] = query_items[q];
            } else {
                auto it = std::lower_bound(ds->h_ids.begin(), ds->h_ids.end(), query_items[q]);
                AH_REQUIRE(it != ds->h_ids.end() && *it == query_items[q], AH_ERR_MISSING_ITEM, "item %u does not exist",
                           query_items[q]);
                rows[q] = (uint32_t)(it - ds->h_ids.begin());
            }
        }
    }
    ContextLease lease(ds);
    AH_REQUIRE(lease.c, AH_ERR_DEVICE, "cannot create a HIP stream");
    Context *ctx = lease.c;
    // candidate filter -> bitmap over item ids (ids beyond the largest stored id cannot match: the list is ascending)
    uint32_t *d_bits = nullptr, *d_leaf_kept = nullptr;
    uint64_t bits_len = 0;
    double filter_share = 1.0;  // of the stored items, about
    if (have_filter) {
        const uint32_t max_id = ds->identity_ids ? (uint32_t)(ds->n - 1) : ds->last_id;
        bits_len = (uint64_t)max_id + 1;
        const size_t words = ((size_t)bits_len + 31) / 32;
        const size_t n_keep = n_filter ? (size_t)(std::upper_bound(filter_sorted, filter_sorted + n_filter, max_id) - filter_sorted) : 0;
        filter_share = (double)n_keep / (double)ds->n;
        AH_TRY(ctx->ensure_filter(words * 4 + n_keep * 4 + (size_t)ix->n_nodes * 4 + 64));
        d_bits = reinterpret_cast<uint32_t *>(ctx->d_filter);
        AH_HIP(hipMemsetAsync(d_bits, 0, words * 4, ctx->stream));
        if (n_keep) {
            uint32_t *d_list = d_bits + words;
            AH_HIP(hipMemcpyAsync(d_list, filter_sorted, n_keep * 4, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_filter_bitmap, dim3(256), dim3(256), 0, ctx->stream, d_list, (uint64_t)n_keep, d_bits);
        }
        d_leaf_kept = d_bits + words + n_keep;
    }
    // Descent: one wave per query unless the filter keeps under 5 % of the items (a query then pops more nodes than the
    // queues of a wave hold).  Under a filter the wave descent reads |descendants & candidates| of every leaf, computed
    // ONCE per submission by a pass over all Descendants ids (4 GB at 10M x 100 trees): a small submission on a big forest
    // is cheaper by the sequential descent, which looks only at the leaves it pops.
    // (the wave / block descents keep the query leaf in LDS: up to 32 KiB of it, i.e. 8192 f32 dimensions)
    bool wave_descent = tun(TUN_SEARCH_WAVE) != 0 && (!d_bits || filter_share >= 0.05) && ((ds->row_bytes() + 255) & ~(size_t)255) <= (32u << 10);
    uint64_t leaf_kept_passes = 0;
    if (wave_descent && d_bits) {
        if (nq >= 16 || ix->desc_len <= (32ull << 20)) {
            SearchParams lp{};
            lp.nodes = ix->d_nodes;
            lp.desc = ix->d_desc;
            lp.filter_bits = d_bits;
            lp.filter_len_bits = bits_len;
            hipLaunchKernelGGL(k_leaf_kept, dim3(1024), dim3(256), 0, ctx->stream, lp, ix->n_nodes, d_leaf_kept);
            AH_HIP(hipGetLastError());
            leaf_kept_passes = 1;
        } else {
            wave_descent = false;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ix->stats_mu);
        ix->stats.calls++;
        ix->stats.leaf_kept_passes += leaf_kept_passes;
    }
    // sub-batches bounded by scratch (~1.5 GiB of candidate buffers)
    const size_t key_bytes = batch_supported((uint32_t)std::min<size_t>(count, 0xFFFFFFFFu))
                                 ? 2 * batch_key_stride((uint32_t)stride) * 8
                                 : topk_scratch_bytes(stride, std::min<size_t>(count, stride)) + 64;
    const size_t per_query = (size_t)stride * 8 + key_bytes + count * 8 + ds->row_bytes() + 4096;
    size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, (1536ull << 20) / per_query));
    chunk = std::min<size_t>(chunk, 4096);
    auto run_range = [&](Context *c, size_t qa, size_t qb) -> int {
        int st = AH_OK;
        for (size_t q0 = qa; q0 < qb && st == AH_OK; q0 += chunk) {
            const size_t cn = std::min(chunk, qb - q0);
            st = search_chunk(ix, c, queries ? queries + q0 * (size_t)ds->dims : nullptr, query_items ? rows.data() + q0 : nullptr, cn, count,
                              (uint32_t)sk_eff, (uint32_t)stride, d_bits, bits_len, filter_share, wave_descent, d_leaf_kept,
                              out_ids + q0 * count, out_distances + q0 * count, out_counts + q0);
        }
        return st;
    };
    return run_range(ctx, 0, nq);
    AH_GUARDED_END
}

// Which kernels served the searches of this index so far (ABI v5); reset != 0 zeroes the counters after the copy.
int ah_index_search_stats(ah_index *ix, ah_search_stats *out, int reset) {
    AH_GUARDED("ah_index_search_stats")
    AH_REQUIRE(ix && out, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(ix->stats_mu);
    *out = ix->stats;
    if (reset) ix->stats = ah_search_stats{};
    return AH_OK;
    AH_GUARDED_END
}

}  // extern "C"
