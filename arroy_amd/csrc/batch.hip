// batch.hip — many re-ranks in five launches (src/reader.rs:376-400 for a whole batch of queries).
//
// A single query touches ~10k candidate rows (tens of MB): far too little to fill 256 CUs, and three launches
// per query make the host the bottleneck.  Here all queries of a submission share the launches:
//   1. k_prepare_queries      codec + D::new_header for every query            (grid = queries)
//   2. k_batch_distances_*    one block per (query, tile of 1024 candidates)   (grid = tiles)
//   3. k_batch_topk_round     tournament of LDS bitonic sorts, grid (chunk, query); repeated until every
//                             query is down to one chunk (2 rounds for search_k ~ 10k, k <= 2048)
//   4. k_batch_topk_emit      (id, normalized distance) pairs, padded with 0xFFFFFFFF / NaN
// Same arithmetic and the same (OrderedFloat(distance), position) keys as the single-query path.
#include "common.h"
#include "device_math.h"

namespace ah {

static constexpr int kBlock = 256;
static constexpr uint32_t kChunk = 4096;   // keys per LDS sort (32 KiB)
static constexpr uint32_t kTileCand = 512;  // candidates of one query per block (1024: 125-query submissions 3 % slower, 5.4 blocks per CU)
static constexpr uint64_t kSentinel = ~0ull;

struct Seg {          // one query's slice of the concatenated candidate list
    uint64_t off;     // first candidate
    uint32_t n;       // candidates
    uint32_t k;       // min(count, n)
};
struct BTile {
    uint32_t query;
    uint32_t first;   // offset inside the query's candidate list
};

// `QueryBuilder::by_vector` for every query of the batch (src/reader.rs:64-75).
__global__ void k_prepare_queries(DataView dv, const float *__restrict__ q_f32, uint8_t *qvecs, uint64_t qstride,
                                  float *qhdrs) {
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const float *src = q_f32 + (uint64_t)q * dv.dims;
    if (!metric_is_bq_dev(dv.metric)) {
        float *dst = reinterpret_cast<float *>(qvecs + q * qstride);
        for (uint32_t i = t; i < dv.pitch; i += blockDim.x) dst[i] = i < dv.dims ? src[i] : 0.0f;
        __syncthreads();
        float hdr0 = 0.0f;
        if (dv.metric == AH_COSINE && t < 8) hdr0 = f_sqrt(octet_reduce_any<OP_DOT>(dst, dst, dv.dims, t));
        if (t == 0) {
            qhdrs[2 * q] = hdr0;
            qhdrs[2 * q + 1] = 0.0f;
        }
    } else {
        uint64_t *dst = reinterpret_cast<uint64_t *>(qvecs + q * qstride);
        for (uint32_t w = t; w < dv.pitch; w += blockDim.x) {
            uint64_t word = 0;
            if (w < dv.words)
                for (uint32_t i = 0; i < 64; i++) {
                    const uint32_t e = 64 * w + i;
                    if (e < dv.dims) word |= (uint64_t)((__float_as_uint(src[e]) >> 31) == 0u) << i;
                }
            dst[w] = word;
        }
        if (t == 0) {
            qhdrs[2 * q] = dv.metric == AH_BQ_COSINE ? f_sqrt((float)(int32_t)(64u * dv.words)) : 0.0f;
            qhdrs[2 * q + 1] = 0.0f;
        }
    }
}

template <int METRIC>
__global__ __launch_bounds__(kBlock) void k_batch_distances_f32(DataView dv, const uint8_t *__restrict__ qvecs,
                                                                uint64_t qstride, const float *__restrict__ qhdrs,
                                                                const Seg *__restrict__ segs,
                                                                const BTile *__restrict__ tiles, uint32_t n_tiles,
                                                                const uint32_t *__restrict__ ids,
                                                                float *__restrict__ out, uint32_t *err) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    extern __shared__ float4 s_q4[];
    __shared__ float s_hdr[2];
    const float *s_q = reinterpret_cast<const float *>(s_q4);
    const uint32_t o = threadIdx.x >> 3, j = threadIdx.x & 7u;
    uint32_t loaded_query = 0xFFFFFFFFu;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const BTile tl = tiles[tile];
        const Seg sg = segs[tl.query];
        if (tl.query != loaded_query) {  // block-uniform
            __syncthreads();
            const float4 *g = reinterpret_cast<const float4 *>(qvecs + tl.query * qstride);
            for (uint32_t i = threadIdx.x; i < (dv.pitch >> 2); i += blockDim.x) s_q4[i] = g[i];
            if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdrs[2 * tl.query + threadIdx.x];
            loaded_query = tl.query;
            __syncthreads();
        }
        const uint32_t in_tile = min(kTileCand, sg.n - tl.first);
        const uint64_t base = sg.off + tl.first;
        for (uint32_t p = o; p < in_tile; p += kBlock / 8) {
            const uint64_t i = base + p;
            const uint32_t id = ids[i];
            const uint64_t row = row_of_id(dv, id);
            if (row == ~0ull) {
                if (j == 0) {
                    atomicOr(err, 1u);
                    out[i] = __uint_as_float(0x7FC00000u);
                }
                continue;
            }
            if ((tl.first + p) > 0 && id <= ids[i - 1] && j == 0) atomicOr(err, 2u);
            const float *rp = dv.rows_f32 + row * dv.pitch;
            float r;
            if (dv.dims >= 32) {
                if (METRIC == AH_MANHATTAN) r = octet_manhattan(s_q, rp, dv.dims, j);
                else r = octet_reduce_stream<OP>(s_q4, rp, dv.dims, j);
            } else if (METRIC == AH_MANHATTAN) {
                r = 0.0f;
                for (uint32_t e = 0; e < dv.dims; e++) r = f_add(r, fabsf(f_sub(s_q[e], rp[e])));
            } else {
                r = thread_reduce_small<OP>(s_q, rp, dv.dims);
            }
            if (j == 0) {
                float d = r;
                if (METRIC == AH_COSINE) d = cosine_from_dot(r, s_hdr[0], dv.headers[row]);
                if (METRIC == AH_DOT_PRODUCT) d = -r;
                out[i] = d;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_batch_distances_bq(DataView dv, const uint8_t *__restrict__ qvecs,
                                                               uint64_t qstride, const float *__restrict__ qhdrs,
                                                               const Seg *__restrict__ segs,
                                                               const BTile *__restrict__ tiles, uint32_t n_tiles,
                                                               const uint32_t *__restrict__ ids,
                                                               float *__restrict__ out, uint32_t *err) {
    extern __shared__ uint64_t s_qw[];
    __shared__ float s_hdr[2];
    uint32_t loaded_query = 0xFFFFFFFFu;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const BTile tl = tiles[tile];
        const Seg sg = segs[tl.query];
        if (tl.query != loaded_query) {
            __syncthreads();
            const uint64_t *g = reinterpret_cast<const uint64_t *>(qvecs + tl.query * qstride);
            for (uint32_t i = threadIdx.x; i < dv.pitch; i += blockDim.x) s_qw[i] = g[i];
            if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdrs[2 * tl.query + threadIdx.x];
            loaded_query = tl.query;
            __syncthreads();
        }
        const uint32_t in_tile = min(kTileCand, sg.n - tl.first);
        const uint64_t base = sg.off + tl.first;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x) {
            const uint64_t i = base + p;
            const uint32_t id = ids[i];
            const uint64_t row = row_of_id(dv, id);
            if (row == ~0ull) {
                atomicOr(err, 1u);
                out[i] = __uint_as_float(0x7FC00000u);
                continue;
            }
            if ((tl.first + p) > 0 && id <= ids[i - 1]) atomicOr(err, 2u);
            const uint64_t *rp = dv.rows_bq + row * dv.pitch;
            uint32_t ham = 0;
            for (uint32_t w = 0; w < dv.pitch; w++) ham += (uint32_t)__popcll(rp[w] ^ s_qw[w]);
            float d;
            if (dv.metric == AH_BQ_EUCLIDEAN) d = (float)(ham * 4u);
            else if (dv.metric == AH_BQ_MANHATTAN) d = (float)(ham * 2u);
            else d = bq_cosine_from_dot((float)bq_dot_from_hamming(ham, dv.words), s_hdr[0], dv.headers[row]);
            out[i] = d;
        }
    }
}

// ---- row-major ("inverted") distances -------------------------------------------------------------------
// A big submission re-reads the same rows many times: 1000 queries x ~10.8k candidates over 1M items touch every
// row ~11 times, and the query-major kernel above pays one HBM read of the row (4*dims bytes) per candidate.
// Here the (query, candidate) pairs are counting-sorted by row first (histogram, exclusive scan, scatter), so that
// pairs of one row sit next to each other: the row is streamed from HBM once (its other uses hit L1 / L2) while the
// queries — a few MB in total — come from L2 / Infinity Cache.  HBM traffic drops from `pairs x 4*dims` to about
// `distinct rows x 4*dims`; the kernels are then bound by the L1/L2 request rate of the operands.  Two kernels:
// k_pairs_distances_runs (>= 3 pairs per row on average: one octet per <= 4 pairs of ONE row, row chunk loaded once
// per step) and k_pairs_distances (one octet per 4 consecutive pairs, both operands loaded per pair).  The arithmetic
// per pair is the same octet reduction, so the distances are bit-identical to the query-major kernel; only the order
// in which pairs are processed changes, and every pair writes its own output slot.
static constexpr int kPairGroup = 4;   // pairs per octet (measured on 10M pairs x 1536 dims: 2: 5.4 ms, 4: 5.6 ms, 8: 6.1 ms)
static constexpr uint32_t kScanItems = 2048;  // counters per scan block (256 threads x 8)

// pass 1: validate the candidates exactly like the query-major kernel and histogram them by row
__global__ __launch_bounds__(kBlock) void k_inv_count(DataView dv, const Seg *__restrict__ segs,
                                                      const BTile *__restrict__ tiles, uint32_t n_tiles,
                                                      const uint32_t *__restrict__ ids, uint32_t *__restrict__ count,
                                                      float *__restrict__ out, uint32_t *err) {
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const BTile tl = tiles[tile];
        const Seg sg = segs[tl.query];
        const uint32_t in_tile = min(kTileCand, sg.n - tl.first);
        const uint64_t base = sg.off + tl.first;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x) {
            const uint64_t i = base + p;
            const uint32_t id = ids[i];
            const uint64_t row = row_of_id(dv, id);
            if (row == ~0ull) {
                atomicOr(err, 1u);
                out[i] = __uint_as_float(0x7FC00000u);
                continue;
            }
            if ((tl.first + p) > 0 && id <= ids[i - 1]) atomicOr(err, 2u);
            atomicAdd(&count[row], 1u);
        }
    }
}

// exclusive scan of `count` in place: per-block scan + block totals, scan of the totals (one block), add back
__global__ __launch_bounds__(256) void k_scan_block(uint32_t *__restrict__ data, uint64_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanItems + (uint64_t)threadIdx.x * 8;
    uint32_t v[8], local = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        v[e] = base + e < n ? data[base + e] : 0u;
        local += v[e];
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = incl - local;
    for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
    if (threadIdx.x == 255) sums[blockIdx.x] = before + local;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (base + e < n) data[base + e] = before;
        before += v[e];
    }
}
__global__ __launch_bounds__(256) void k_scan_sums(uint32_t *__restrict__ sums, uint32_t n_sums, uint32_t *__restrict__ total) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < n_sums; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_sums ? sums[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry + incl - v;
        for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
        if (i < n_sums) sums[i] = before;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = before + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ data, uint64_t n, const uint32_t *__restrict__ sums) {
    const uint32_t add = sums[blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * kScanItems;
    for (uint32_t e = threadIdx.x; e < kScanItems; e += 256)
        if (base + e < n) data[base + e] += add;
}

// pass 2: scatter (row, query, position) into row order; `start` holds the exclusive scan and is advanced
__global__ __launch_bounds__(kBlock) void k_inv_scatter(DataView dv, const Seg *__restrict__ segs,
                                                        const BTile *__restrict__ tiles, uint32_t n_tiles,
                                                        const uint32_t *__restrict__ ids, uint32_t *__restrict__ start,
                                                        uint64_t *__restrict__ pair_rq, uint32_t *__restrict__ pair_pos) {
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const BTile tl = tiles[tile];
        const Seg sg = segs[tl.query];
        const uint32_t in_tile = min(kTileCand, sg.n - tl.first);
        const uint64_t base = sg.off + tl.first;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x) {
            const uint64_t i = base + p;
            const uint64_t row = row_of_id(dv, ids[i]);
            if (row == ~0ull) continue;
            const uint32_t slot = atomicAdd(&start[row], 1u);
            pair_rq[slot] = (row << 32) | (uint64_t)tl.query;
            pair_pos[slot] = (uint32_t)i;
        }
    }
}

// pass 3: one octet per G consecutive pairs of the row-sorted list.  Both operands come through L1: the row chunk
// (non-temporal; the first pair of a row pulls it from HBM, its neighbours hit L1/L2) and the query chunk (L2).
// (Splitting the queries into per-XCD classes so that each L2 only sees 1/8 of them was measured and does not
// help: the kernel is bound by the L1 request rate, ~21-24 TB/s of operand traffic, not by L2 capacity.)
template <int METRIC, int G>
__global__ __launch_bounds__(kBlock) void k_pairs_distances(DataView dv, const uint8_t *__restrict__ qvecs,
                                                            uint64_t qstride, const float *__restrict__ qhdrs,
                                                            const uint64_t *__restrict__ pair_rq,
                                                            const uint32_t *__restrict__ pair_pos,
                                                            const uint32_t *__restrict__ n_pairs_p,
                                                            float *__restrict__ out) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t hi = *n_pairs_p;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.dims >> 5;
    for (uint64_t g0 = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3) * G; g0 < hi; g0 += n_octets * G) {
        // branch-free inner loop: a short last group repeats its final pair (only the store is guarded), and every
        // pair loads its row chunk itself — pairs of one row sit next to each other, so the repeats are L1 hits
        const float4 *r4[G];
        const float4 *q4[G];
        bool on[G];
        float4 acc[G];
#pragma unroll
        for (int t = 0; t < G; t++) {
            on[t] = g0 + t < hi;
            const uint64_t rq = pair_rq[on[t] ? g0 + t : hi - 1];
            r4[t] = reinterpret_cast<const float4 *>(dv.rows_f32 + (rq >> 32) * dv.pitch) + j;
            q4[t] = reinterpret_cast<const float4 *>(qvecs + (uint64_t)(uint32_t)rq * qstride) + j;
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (uint32_t k = 0; k < blocks; k++) {
            float4 x[G], q[G];
#pragma unroll
            for (int t = 0; t < G; t++) {
                x[t] = ld_stream(r4[t] + k * 8);
                q[t] = q4[t][k * 8];
            }
#pragma unroll
            for (int t = 0; t < G; t++) fma_step<OP>(acc[t], q[t], x[t]);
        }
#pragma unroll
        for (int t = 0; t < G; t++) {
            if (on[t]) {
                const float *rp = reinterpret_cast<const float *>(r4[t] - j);
                const float *qp = reinterpret_cast<const float *>(q4[t] - j);
                float r = octet_finish(acc[t]);
                r = scalar_tail<OP>(r, qp, rp, blocks << 5, dv.dims);
                if (j == 0) {
                    const uint64_t rq = pair_rq[g0 + t];
                    float d = r;
                    if (METRIC == AH_COSINE) d = cosine_from_dot(r, qhdrs[2 * (uint32_t)rq], dv.headers[rq >> 32]);
                    if (METRIC == AH_DOT_PRODUCT) d = -r;
                    out[pair_pos[g0 + t]] = d;
                }
            }
        }
    }
}

// Work items of the row-run kernel: the pairs of one row, cut into groups of <= kRunGroup.  `end[r]` is the end of
// row r's bucket after the scatter (its begin is end[r-1]).
static constexpr int kRunGroup = 4;
__global__ __launch_bounds__(256) void k_inv_item_count(const uint32_t *__restrict__ end, uint64_t n_rows,
                                                        uint32_t *__restrict__ icount) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const uint32_t c = end[r] - (r ? end[r - 1] : 0u);
        icount[r] = (c + kRunGroup - 1) / kRunGroup;
    }
}
__global__ __launch_bounds__(256) void k_inv_item_fill(const uint32_t *__restrict__ end, uint64_t n_rows,
                                                       const uint32_t *__restrict__ istart, uint64_t *__restrict__ items) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const uint32_t begin = r ? end[r - 1] : 0u, c = end[r] - begin;
        uint64_t *dst = items + istart[r];
        for (uint32_t i = 0; i * kRunGroup < c; i++)
            dst[i] = (uint64_t)(begin + i * kRunGroup) | ((uint64_t)min((uint32_t)kRunGroup, c - i * kRunGroup) << 32);
    }
}

// pass 3, row-run version: one octet per item = up to kRunGroup pairs of ONE row.  The row chunk is loaded once per
// step and serves every pair of the item (L1 traffic: 1 row + G query operands per G pairs instead of 2G); a short
// item repeats its last pair (only the store is guarded), so the inner loop has no branches.
template <int METRIC>
__global__ __launch_bounds__(kBlock) void k_pairs_distances_runs(DataView dv, const uint8_t *__restrict__ qvecs,
                                                                 uint64_t qstride, const float *__restrict__ qhdrs,
                                                                 const uint64_t *__restrict__ pair_rq,
                                                                 const uint32_t *__restrict__ pair_pos,
                                                                 const uint64_t *__restrict__ items,
                                                                 const uint32_t *__restrict__ n_items_p,
                                                                 float *__restrict__ out) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    constexpr int G = kRunGroup;
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_items = *n_items_p;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.dims >> 5;
    for (uint64_t it = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; it < n_items; it += n_octets) {
        const uint64_t item = items[it];
        const uint32_t first = (uint32_t)item, cnt = (uint32_t)(item >> 32);
        const uint64_t row = pair_rq[first] >> 32;
        const float *rp = dv.rows_f32 + row * dv.pitch;
        const float4 *r4 = reinterpret_cast<const float4 *>(rp) + j;
        const float4 *q4[G];
        float4 acc[G];
#pragma unroll
        for (int t = 0; t < G; t++) {
            const uint32_t qi = (uint32_t)pair_rq[first + min((uint32_t)t, cnt - 1)];
            q4[t] = reinterpret_cast<const float4 *>(qvecs + (uint64_t)qi * qstride) + j;
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint32_t k = 0;
        for (; k + 2 <= blocks; k += 2) {
            const float4 x0 = ld_stream(r4 + k * 8), x1 = ld_stream(r4 + (k + 1) * 8);
            float4 q0[G], q1[G];
#pragma unroll
            for (int t = 0; t < G; t++) {
                q0[t] = q4[t][k * 8];
                q1[t] = q4[t][(k + 1) * 8];
            }
#pragma unroll
            for (int t = 0; t < G; t++) {
                fma_step<OP>(acc[t], q0[t], x0);
                fma_step<OP>(acc[t], q1[t], x1);
            }
        }
        for (; k < blocks; k++) {
            const float4 x = r4[k * 8];
#pragma unroll
            for (int t = 0; t < G; t++) fma_step<OP>(acc[t], q4[t][k * 8], x);
        }
#pragma unroll
        for (int t = 0; t < G; t++) {
            if ((uint32_t)t < cnt) {
                const float *qp = reinterpret_cast<const float *>(q4[t] - j);
                float r = octet_finish(acc[t]);
                r = scalar_tail<OP>(r, qp, rp, blocks << 5, dv.dims);
                if (j == 0) {
                    float d = r;
                    if (METRIC == AH_COSINE) d = cosine_from_dot(r, qhdrs[2 * (uint32_t)pair_rq[first + t]], dv.headers[row]);
                    if (METRIC == AH_DOT_PRODUCT) d = -r;
                    out[pair_pos[first + t]] = d;
                }
            }
        }
    }
}

// AH_RERANK_INVERT=0 never uses the row-major path, =1 uses it whenever it is legal (A/B measurements)
// A/B switches (tunables of common.h): AH_RERANK_INVERT, AH_PAIR_GROUP, AH_PAIR_RUNS (the row-run kernel)
#define g_invert_force ((int)tun(TUN_RERANK_INVERT))
#define g_pair_group ((int)tun(TUN_PAIR_GROUP))
#define g_pair_runs (tun(TUN_PAIR_RUNS) != 0)

// counters: one per stored row + the scan's block totals + the grand total
// counters: per stored row a pair bucket and an item count, each with its scan block totals + grand total; then the
// item list of the row-run kernel (<= pairs / kRunGroup + one partial item per non-empty row)
size_t batch_invert_counter_bytes(uint64_t n_rows, uint64_t n_candidates) {
    const uint64_t per_scan = n_rows + (n_rows + kScanItems - 1) / kScanItems + 64;
    const uint64_t max_items = n_candidates / kRunGroup + std::min(n_rows, n_candidates) + 64;
    return (size_t)(2 * per_scan * 4 + max_items * 8);
}
static bool invert_legal(const DataView &dv, uint64_t n_pairs) {
    return !metric_is_bq(dv.metric) && dv.metric != AH_MANHATTAN && dv.dims >= 32 && n_pairs > 0 && n_pairs < 0xFFFFFFFFull &&
           dv.n > 0 && dv.n < 0xFFFFFFFFull;
}
// policy: row-major when the submission re-reads rows (>= 2 candidates per stored row on average)
bool batch_invert_wanted(const DataView &dv, uint64_t n_candidates) {
    if (!invert_legal(dv, n_candidates) || g_invert_force == 0) return false;
    return g_invert_force == 1 || n_candidates >= 2 * dv.n;
}

// The pair lists live in the tournament buffers, which are idle until the distances exist:
// keys_a: pair_rq (8 B x pairs); keys_b: pair_pos (4 B x pairs).  The counters are caller-provided scratch.
template <int METRIC>
static int launch_inverted(const DataView &dv, uint32_t /*n_queries*/, const uint8_t *d_qvecs, uint64_t qstride,
                            const float *d_qhdrs, const Seg *d_segs, const BTile *d_tiles, uint32_t n_tiles,
                            const uint32_t *d_ids, uint64_t n_pairs, float *d_dist, uint64_t *d_keys_a, uint64_t *d_keys_b,
                            uint32_t *d_counters, uint32_t *d_err, hipStream_t s) {
    uint64_t *pair_rq = d_keys_a;
    uint32_t *pair_pos = reinterpret_cast<uint32_t *>(d_keys_b);
    uint32_t *count = d_counters;
    const uint32_t n_sums = (uint32_t)((dv.n + kScanItems - 1) / kScanItems);
    uint32_t *sums = count + dv.n;
    uint32_t *total = sums + n_sums;  // valid pairs, known to the device only
    const unsigned grid = n_tiles < 4096u ? n_tiles : 4096u;
    AH_HIP(hipMemsetAsync(count, 0, (size_t)dv.n * 4, s));
    hipLaunchKernelGGL(k_inv_count, dim3(grid), dim3(kBlock), 0, s, dv, d_segs, d_tiles, n_tiles, d_ids, count, d_dist, d_err);
    hipLaunchKernelGGL(k_scan_block, dim3(n_sums), dim3(256), 0, s, count, dv.n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, sums, n_sums, total);
    hipLaunchKernelGGL(k_scan_add, dim3(n_sums), dim3(256), 0, s, count, dv.n, sums);
    hipLaunchKernelGGL(k_inv_scatter, dim3(grid), dim3(kBlock), 0, s, dv, d_segs, d_tiles, n_tiles, d_ids, count, pair_rq,
                       pair_pos);
    // row-run items pay one row operand + kRunGroup query operands each; with < 3 pairs per row most items are short and
    // the plain pair-per-slot kernel moves fewer operands through L1
    if (g_pair_runs && n_pairs >= 3 * dv.n) {
        const uint64_t per_scan = dv.n + n_sums + 64;
        uint32_t *icount = count + per_scan, *isums = icount + dv.n, *itotal = isums + n_sums;
        uint64_t *items = reinterpret_cast<uint64_t *>(count + 2 * per_scan);
        const unsigned rgrid = (unsigned)std::min<uint64_t>((dv.n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_inv_item_count, dim3(rgrid), dim3(256), 0, s, count, dv.n, icount);
        hipLaunchKernelGGL(k_scan_block, dim3(n_sums), dim3(256), 0, s, icount, dv.n, isums);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, isums, n_sums, itotal);
        hipLaunchKernelGGL(k_scan_add, dim3(n_sums), dim3(256), 0, s, icount, dv.n, isums);
        hipLaunchKernelGGL(k_inv_item_fill, dim3(rgrid), dim3(256), 0, s, count, dv.n, icount, items);
        const uint64_t max_items = n_pairs / kRunGroup + std::min<uint64_t>(dv.n, n_pairs);
        const unsigned igrid = (unsigned)std::min<uint64_t>((max_items + kBlock / 8 - 1) / (kBlock / 8), 32768);
        hipLaunchKernelGGL((k_pairs_distances_runs<METRIC>), dim3(igrid), dim3(kBlock), 0, s, dv, d_qvecs, qstride, d_qhdrs,
                           pair_rq, pair_pos, items, itotal, d_dist);
        AH_HIP(hipGetLastError());
        return AH_OK;
    }
    const int group = g_pair_group == 8 ? 8 : (g_pair_group == 2 ? 2 : kPairGroup);
    const uint64_t octets = (n_pairs + group - 1) / group;
    const unsigned pgrid = (unsigned)std::min<uint64_t>((octets + kBlock / 8 - 1) / (kBlock / 8), 32768);
    if (group == 8)
        hipLaunchKernelGGL((k_pairs_distances<METRIC, 8>), dim3(pgrid), dim3(kBlock), 0, s, dv, d_qvecs, qstride, d_qhdrs,
                           pair_rq, pair_pos, total, d_dist);
    else if (group == 2)
        hipLaunchKernelGGL((k_pairs_distances<METRIC, 2>), dim3(pgrid), dim3(kBlock), 0, s, dv, d_qvecs, qstride, d_qhdrs,
                           pair_rq, pair_pos, total, d_dist);
    else
        hipLaunchKernelGGL((k_pairs_distances<METRIC, kPairGroup>), dim3(pgrid), dim3(kBlock), 0, s, dv, d_qvecs, qstride,
                           d_qhdrs, pair_rq, pair_pos, total, d_dist);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// ---- batched top-k ----------------------------------------------------------------------------------
// State of one query's tournament after `rounds` rounds: keys in flight and the number of blocks that wrote
// them.  Pure function of (n, k), evaluated identically by every kernel and by the host.
struct Tournament {
    uint32_t n_in;    // keys entering the next round
    uint32_t blocks;  // blocks of the last executed round (0 before round 0)
    uint32_t keep;    // keys each of those blocks kept
};
__host__ __device__ inline uint32_t tour_keep(uint32_t blocks, uint32_t k) { return blocks == 1 ? k : min(k, kChunk / 2); }
__host__ __device__ inline Tournament tour_after(uint32_t n, uint32_t k, uint32_t rounds) {
    Tournament t{n, 0u, 0u};
    for (uint32_t r = 0; r < rounds; r++) {
        if (t.blocks == 1) break;  // finished: further rounds are no-ops
        t.blocks = (t.n_in + kChunk - 1) / kChunk;
        t.keep = tour_keep(t.blocks, k);
        t.n_in = t.blocks * t.keep;
    }
    return t;
}
__host__ __device__ inline uint32_t tour_rounds(uint32_t n, uint32_t k) {
    uint32_t r = 0;
    Tournament t{n, 0u, 0u};
    while (t.blocks != 1) {
        t.blocks = (t.n_in + kChunk - 1) / kChunk;
        t.keep = tour_keep(t.blocks, k);
        t.n_in = t.blocks * t.keep;
        r++;
    }
    return r;
}

__device__ __forceinline__ uint64_t batch_make_key(float d, uint32_t pos, uint32_t id, uint64_t two_k) {
    const uint32_t ok = orderable_key(d);
    if (pos >= two_k) {  // reader.rs:611,619-621: items >= (f32::MAX, u32::MAX) are skipped once 2k are buffered
        const uint32_t max_key = 0xFF7FFFFFu;
        if (ok > max_key || (ok == max_key && id == 0xFFFFFFFFu)) return kSentinel;
    }
    return ((uint64_t)ok << 32) | (uint64_t)pos;
}

__device__ __forceinline__ void batch_bitonic_sort(uint64_t *s) {
    for (uint32_t size = 2; size <= kChunk; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (kChunk >> 1); t += blockDim.x) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = s[lo], b = s[hi];
                if ((a > b) == up) {
                    s[lo] = b;
                    s[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// ---- top-k by selection (one block per query) --------------------------------------------------------------
// The tournament above sorts every key of a query (three LDS bitonic sorts of 4096 keys + one more for search_k ~ 10k) to
// keep 100 of them: ~0.4 ms per 125-query submission, a fifth of its time.  Selection: the keys' distance words are binned
// linearly between the query's smallest and largest one (2048 bins, monotone, so the order of the keys is untouched), the
// bin holding the k-th smallest key is found by a scan, and only the keys of the bins up to it (k plus a handful) are sorted.
// Exactly the k smallest (OrderedFloat(distance), position) keys in ascending order — the tournament's result; a query whose
// selected set does not fit the small sort (> 1024 keys: many equal distances, or a huge k) raises its flag and is left to
// the tournament, whose blocks return at once for all other queries.  The flag of query q lives in the last word of its
// slice of keys_b (the tournament uses at most the lower half of a slice).
constexpr uint32_t kSelBins = 2048, kSelCap = 1024;
__device__ __forceinline__ uint64_t select_key(const float *__restrict__ dist, const uint32_t *__restrict__ ids, uint64_t off,
                                               uint32_t g, uint64_t two_k) {
    const uint32_t ok = orderable_key(dist[off + g]);
    if (g >= two_k && ok >= 0xFF7FFFFFu) {  // reader.rs:611,619-621 (see batch_make_key): needs the id only here
        if (ok > 0xFF7FFFFFu || ids[off + g] == 0xFFFFFFFFu) return kSentinel;
    }
    return ((uint64_t)ok << 32) | (uint64_t)g;
}
__global__ __launch_bounds__(kBlock) void k_batch_topk_select(const Seg *__restrict__ segs, const float *__restrict__ dist,
                                                              const uint32_t *__restrict__ ids, uint64_t *keys_a,
                                                              uint64_t *keys_b, uint64_t kstride) {
    __shared__ uint32_t s_hist[kSelBins];
    __shared__ uint64_t s_cand[kSelCap];
    __shared__ uint32_t s_min, s_max, s_wave[kBlock / 64], s_bin, s_count, s_n;
    const uint32_t q = blockIdx.x;
    const Seg sg = segs[q];
    uint64_t *flag = keys_b + (uint64_t)q * kstride + (kstride - 1);
    if (sg.k == 0) {
        if (threadIdx.x == 0) *flag = 0;
        return;
    }
    const uint64_t two_k = 2ull * sg.k;
    // the buffer k_batch_topk_emit reads: the one the tournament's last round would have written
    const uint32_t rounds = tour_rounds(sg.n, sg.k);
    uint64_t *dst = ((rounds - 1) & 1u) ? keys_b + (uint64_t)q * kstride : keys_a + (uint64_t)q * kstride;
    for (uint32_t b = threadIdx.x; b < kSelBins; b += kBlock) s_hist[b] = 0;
    if (threadIdx.x == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0u;
        s_n = 0u;
    }
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t g = threadIdx.x; g < sg.n; g += kBlock) {
        const uint32_t w = (uint32_t)(select_key(dist, ids, sg.off, g, two_k) >> 32);
        lo = min(lo, w);
        hi = max(hi, w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicMin(&s_min, lo);
        atomicMax(&s_max, hi);
    }
    __syncthreads();
    const uint32_t w_min = s_min;
    const uint64_t span = (uint64_t)(s_max - w_min) + 1ull;
    // bin(w) = floor((w - w_min) * scale / 2^32), scale = floor(2048 * 2^32 / span): monotone, < 2048; one 64-bit division
    // per block instead of one per key.  Fewer than 2048 distinct words: every word its own bin.
    const bool direct = span <= kSelBins;
    const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)kSelBins << 32) / span);
    auto bin_of = [&](uint32_t w) -> uint32_t {
        return direct ? w - w_min : (uint32_t)(((uint64_t)(w - w_min) * scale) >> 32);
    };
    for (uint32_t g = threadIdx.x; g < sg.n; g += kBlock) {
        const uint32_t w = (uint32_t)(select_key(dist, ids, sg.off, g, two_k) >> 32);
        atomicAdd(&s_hist[bin_of(w)], 1u);
    }
    __syncthreads();
    {  // the bin of the k-th smallest key: thread t owns bins 8t .. 8t+7
        uint32_t c[8], mine = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            c[u] = s_hist[threadIdx.x * 8 + u];
            mine += c[u];
        }
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63u) >= off) incl += up;
        }
        if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += s_wave[w];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (before < sg.k && before + c[u] >= sg.k) {  // exactly one bin qualifies (sg.k <= n)
                s_bin = threadIdx.x * 8 + u;
                s_count = before + c[u];
            }
            before += c[u];
        }
    }
    __syncthreads();
    const uint32_t n_sel = s_count, bin_k = s_bin;
    if (n_sel > kSelCap) {  // block-uniform: leave this query to the tournament
        if (threadIdx.x == 0) *flag = 1;
        return;
    }
    if (threadIdx.x == 0) *flag = 0;
    for (uint32_t g = threadIdx.x; g < sg.n; g += kBlock) {
        const uint64_t key = select_key(dist, ids, sg.off, g, two_k);
        const uint32_t w = (uint32_t)(key >> 32);
        if (bin_of(w) <= bin_k) s_cand[atomicAdd(&s_n, 1u)] = key;
    }
    __syncthreads();
    uint32_t p2 = 64;
    while (p2 < n_sel) p2 <<= 1;
    for (uint32_t t = n_sel + threadIdx.x; t < p2; t += kBlock) s_cand[t] = kSentinel;
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (p2 >> 1); t += kBlock) {
                const uint32_t a_i = 2 * t - (t & (stride - 1)), b_i = a_i + stride;
                const bool up = (a_i & size) == 0;
                const uint64_t x = s_cand[a_i], y = s_cand[b_i];
                if ((x > y) == up) {
                    s_cand[a_i] = y;
                    s_cand[b_i] = x;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < sg.k; t += kBlock) dst[t] = s_cand[t];
}

// round r reads buffer (r odd ? B : A) ... round 0 reads the distances; it writes buffer (r even ? A : B).
__global__ __launch_bounds__(kBlock) void k_batch_topk_round(const Seg *__restrict__ segs, uint32_t round,
                                                             const float *__restrict__ dist,
                                                             const uint32_t *__restrict__ ids, uint64_t *keys_a,
                                                             uint64_t *keys_b, uint64_t kstride) {
    __shared__ uint64_t s[kChunk];
    const uint32_t q = blockIdx.y, c = blockIdx.x;
    const Seg sg = segs[q];
    if (sg.k == 0) return;
    if (keys_b[(uint64_t)q * kstride + (kstride - 1)] == 0) return;  // k_batch_topk_select served this query
    const Tournament before = tour_after(sg.n, sg.k, round);
    if (before.blocks == 1) return;  // this query finished in an earlier round
    const uint32_t blocks = (before.n_in + kChunk - 1) / kChunk;
    if (c >= blocks) return;
    const uint32_t keep = tour_keep(blocks, sg.k);
    const uint64_t *src = (round & 1u) ? keys_a + q * kstride : keys_b + q * kstride;  // written by round-1
    uint64_t *dst = (round & 1u) ? keys_b + q * kstride : keys_a + q * kstride;
    const uint64_t two_k = 2ull * sg.k;
    for (uint32_t t = threadIdx.x; t < kChunk; t += blockDim.x) {
        const uint32_t g = c * kChunk + t;
        uint64_t key = kSentinel;
        if (g < before.n_in) key = round == 0 ? batch_make_key(dist[sg.off + g], g, ids[sg.off + g], two_k) : src[g];
        s[t] = key;
    }
    batch_bitonic_sort(s);
    for (uint32_t t = threadIdx.x; t < keep; t += blockDim.x) dst[(uint64_t)c * keep + t] = s[t];
}

__global__ void k_batch_topk_emit(DataView dv, const Seg *__restrict__ segs, const float *__restrict__ dist,
                                  const uint32_t *__restrict__ ids, const uint64_t *keys_a, const uint64_t *keys_b,
                                  uint64_t kstride, uint32_t k_out, uint32_t *out_ids, float *out_dist) {
    const uint32_t q = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k_out) return;
    const Seg sg = segs[q];
    uint32_t id = 0xFFFFFFFFu;
    float d = __uint_as_float(0xFFFFFFFFu);  // NaN padding
    if (t < sg.k) {
        const uint32_t rounds = tour_rounds(sg.n, sg.k);  // the last round r = rounds-1 wrote (r even ? A : B)
        const uint64_t *keys = ((rounds - 1) & 1u) ? keys_b + q * kstride : keys_a + q * kstride;
        const uint32_t pos = (uint32_t)keys[t];
        id = ids[sg.off + pos];
        d = normalized_distance(dv.metric, dist[sg.off + pos], dv.dims);
    }
    out_ids[(uint64_t)q * k_out + t] = id;
    out_dist[(uint64_t)q * k_out + t] = d;
}

// ---- host driver --------------------------------------------------------------------------------------
// All device pointers are caller-carved scratch.  `h_segs` / `h_tiles` are host arrays already filled.
size_t batch_key_stride(uint32_t max_n) { return (size_t)((max_n + kChunk - 1) / kChunk) * kChunk; }
uint32_t batch_tile_candidates() { return kTileCand; }
bool batch_supported(uint32_t k) { return k <= kChunk / 2; }

int launch_prepare_queries_only(const DataView &dv, const float *d_q_f32, uint32_t n_queries, uint8_t *d_qvecs,
                                uint64_t qstride, float *d_qhdrs, hipStream_t s) {
    if (n_queries)
        hipLaunchKernelGGL(k_prepare_queries, dim3(n_queries), dim3(metric_is_bq_dev(dv.metric) ? 64 : 512), 0, s, dv, d_q_f32, d_qvecs, qstride,
                           d_qhdrs);  // (d_q_f32 may be pinned host memory: few dependent reads per thread)
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// distances + tournament top-k + emit for queries whose leaves (qvecs / qhdrs) are already on the device
int launch_rerank_batch_prepared(const DataView &dv, uint32_t n_queries, const uint8_t *d_qvecs, uint64_t qstride,
                                 const float *d_qhdrs, const void *d_segs_v, const void *d_tiles_v, uint32_t n_tiles,
                                 const uint32_t *d_ids, float *d_dist, uint64_t *d_keys_a, uint64_t *d_keys_b,
                                 uint64_t kstride, uint32_t max_n, uint32_t k_out, uint32_t max_rounds,
                                 uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_err, hipStream_t s,
                                 uint64_t n_candidates, uint32_t *d_inv_counters) {
    const Seg *d_segs = reinterpret_cast<const Seg *>(d_segs_v);
    const BTile *d_tiles = reinterpret_cast<const BTile *>(d_tiles_v);
    if (n_tiles) {
        const unsigned grid = n_tiles < 4096u ? n_tiles : 4096u;
        const bool invert = d_inv_counters != nullptr && batch_invert_wanted(dv, n_candidates);
        if (invert) {
            switch (dv.metric) {
            case AH_EUCLIDEAN:
                AH_TRY(launch_inverted<AH_EUCLIDEAN>(dv, n_queries, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, n_tiles, d_ids,
                                              n_candidates, d_dist, d_keys_a, d_keys_b, d_inv_counters, d_err, s));
                break;
            case AH_COSINE:
                AH_TRY(launch_inverted<AH_COSINE>(dv, n_queries, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, n_tiles, d_ids,
                                           n_candidates, d_dist, d_keys_a, d_keys_b, d_inv_counters, d_err, s));
                break;
            default:
                AH_TRY(launch_inverted<AH_DOT_PRODUCT>(dv, n_queries, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, n_tiles, d_ids,
                                                n_candidates, d_dist, d_keys_a, d_keys_b, d_inv_counters, d_err, s));
                break;
            }
        } else if (metric_is_bq(dv.metric)) {
            hipLaunchKernelGGL(k_batch_distances_bq, dim3(grid), dim3(kBlock), dv.pitch * 8, s, dv, d_qvecs, qstride,
                               d_qhdrs, d_segs, d_tiles, n_tiles, d_ids, d_dist, d_err);
        } else {
            const size_t sh = (size_t)dv.pitch * 4;
#define AH_LAUNCH(M)                                                                                          \
    hipLaunchKernelGGL((k_batch_distances_f32<M>), dim3(grid), dim3(kBlock), sh, s, dv, d_qvecs, qstride,      \
                       d_qhdrs, d_segs, d_tiles, n_tiles, d_ids, d_dist, d_err)
            switch (dv.metric) {
            case AH_EUCLIDEAN: AH_LAUNCH(AH_EUCLIDEAN); break;
            case AH_MANHATTAN: AH_LAUNCH(AH_MANHATTAN); break;
            case AH_COSINE: AH_LAUNCH(AH_COSINE); break;
            default: AH_LAUNCH(AH_DOT_PRODUCT); break;
            }
#undef AH_LAUNCH
        }
        hipLaunchKernelGGL(k_batch_topk_select, dim3(n_queries), dim3(kBlock), 0, s, d_segs, d_dist, d_ids, d_keys_a, d_keys_b,
                           kstride);
        const uint32_t max_blocks = (max_n + kChunk - 1) / kChunk;
        uint32_t blocks_bound = max_blocks;
        for (uint32_t r = 0; r < max_rounds; r++) {
            hipLaunchKernelGGL(k_batch_topk_round, dim3(blocks_bound, n_queries), dim3(kBlock), 0, s, d_segs, r, d_dist,
                               d_ids, d_keys_a, d_keys_b, kstride);
            blocks_bound = (blocks_bound * (kChunk / 2) + kChunk - 1) / kChunk;  // every round at least halves
            if (blocks_bound < 1) blocks_bound = 1;
        }
    }
    hipLaunchKernelGGL(k_batch_topk_emit, dim3((k_out + 255) / 256, n_queries), dim3(256), 0, s, dv, d_segs, d_dist,
                       d_ids, d_keys_a, d_keys_b, kstride, k_out, d_out_ids, d_out_dist);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

int launch_rerank_batch(const DataView &dv, const float *d_q_f32, uint32_t n_queries, uint8_t *d_qvecs, uint64_t qstride,
                        float *d_qhdrs, const void *d_segs_v, const void *d_tiles_v, uint32_t n_tiles,
                        const uint32_t *d_ids, float *d_dist, uint64_t *d_keys_a, uint64_t *d_keys_b, uint64_t kstride,
                        uint32_t max_n, uint32_t k_out, uint32_t max_rounds, uint32_t *d_out_ids, float *d_out_dist,
                        uint32_t *d_err, hipStream_t s, uint64_t n_candidates, uint32_t *d_inv_counters) {
    AH_TRY(launch_prepare_queries_only(dv, d_q_f32, n_queries, d_qvecs, qstride, d_qhdrs, s));
    return launch_rerank_batch_prepared(dv, n_queries, d_qvecs, qstride, d_qhdrs, d_segs_v, d_tiles_v, n_tiles, d_ids,
                                        d_dist, d_keys_a, d_keys_b, kstride, max_n, k_out, max_rounds, d_out_ids,
                                        d_out_dist, d_err, s, n_candidates, d_inv_counters);
}

uint32_t batch_rounds(uint32_t n, uint32_t k) { return (n == 0 || k == 0) ? 0u : tour_rounds(n, k); }

}  // namespace ah
