// forest.hip — whole-forest build on one GPU: `make_tree_in_file` (src/writer.rs:1167-1261) for every tree
// of the batch, level-synchronously.
//
// The reference recurses depth-first per tree under rayon (src/writer.rs:568-591,798-828); every node of
// every tree at the same depth is independent, so here ONE set of launches handles one level of ALL trees:
//
//   per level   create_split (one wave per node)  ->  margins + sides over node-major tiles (the HBM-bound
//               pass: 4*dims bytes per (item, node visit))  ->  accept / retry (<= 4 attempts,
//               src/writer.rs:1193-1216)  ->  random fallback (:1220-1227)  ->  stable partition of every
//               node's id list into its two children (ascending ids preserved, :1230-1231).
//
// Item lists live in HBM as one permutation of row indices per tree: a node owns perm[start, start+count),
// its children subdivide that range, so positions never move between nodes.  Randomness is the
// counter-based policy of include/arroy_hip_policy.h: a pure function of (tree_seed, node path, attempt,
// draw), so this breadth-first build and the depth-first CPU oracle produce the same forest bit for bit.
// No atomics on floats, no dependence on workgroup scheduling: results are deterministic.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdlib>
#include <new>

#include "common.h"
#include "split_device.h"

namespace ah {

static constexpr uint32_t kTile = 2048;  // items per tile = 32 octets x 64 items
static constexpr int kBlock = 256;
static constexpr int kMaxBlocks = 4096;

enum { ST_PENDING = 0, ST_ACCEPTED = 1, ST_RANDOM = 2 };

struct FNode {
    uint64_t key;      // ah_node_key_* of this node
    uint64_t start;    // first position inside the batch permutation (absolute: tree base + offset)
    uint32_t tree;     // tree index inside the batch
    uint32_t count;    // items under the node
    uint32_t n_left;   // accumulated by the margin kernel (integer atomics)
    uint32_t attempt;  // current / final split attempt (0..3)
    uint32_t state;    // ST_*
    uint32_t tile_begin, n_tiles;
    uint32_t rec;      // host record index
};
struct FTile {
    uint32_t node;
    uint32_t first;  // offset of the tile inside its node
};

__global__ void k_init_perm(uint32_t *perm, uint64_t n, uint32_t n_trees) {
    const uint64_t total = n * n_trees;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride)
        perm[g] = (uint32_t)(g % n);
}
// item ids -> rows, in place (sub-tree builds start from caller-given id lists)
__global__ void k_ids_to_rows(DataView dv, uint32_t *perm, uint64_t total, uint32_t *err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint64_t row = row_of_id(dv, perm[g]);
        if (row == ~0ull) atomicOr(err, 1u);
        perm[g] = (uint32_t)row;
    }
}

// One wave per pending node: sample 2+10 items with the policy RNG, run create_split.
//
// Normal records (device layout == the layout handed to the caller): [vector, row_bytes][header, 16-byte slot].
__global__ __launch_bounds__(64) void k_forest_create_split(DataView dv, FNode *nodes, const uint32_t *__restrict__ perm,
                                                            uint64_t n_items, uint8_t *normals, uint64_t nstride,
                                                            uint64_t hdr_off) {
    FNode &nd = nodes[blockIdx.x];
    if (nd.state != ST_PENDING) return;
    extern __shared__ float4 s_buf4[];
    float *s_buf = reinterpret_cast<float *>(s_buf4);
    __shared__ uint32_t s_rows[AH_SPLIT_SAMPLES];
    const uint32_t fpitch = f32_space_pitch(dv.metric, dv.dims);
    const uint32_t *pp = perm + nd.start;  // starts are absolute positions in the batch permutation
    if (threadIdx.x == 0) {
        uint64_t a, b;
        ah_choose_two(nd.key, nd.attempt, nd.count, &a, &b);  // src/parallel.rs:342-355
        s_rows[0] = pp[a];
        s_rows[1] = pp[b];
        nd.n_left = 0;
    } else if (threadIdx.x >= 2 && threadIdx.x < AH_SPLIT_SAMPLES) {
        s_rows[threadIdx.x] = pp[ah_choose(nd.key, nd.attempt, threadIdx.x - 2, nd.count)];  // :358-367
    }
    __syncthreads();
    uint8_t *rec = normals + blockIdx.x * nstride;
    float *hdr = reinterpret_cast<float *>(rec + hdr_off);
    wave_create_split_any(dv, s_rows, s_buf, s_buf + fpitch, s_buf + 2 * fpitch, rec, hdr, threadIdx.x);
    if (threadIdx.x == 0) hdr[2] = hdr[3] = 0.0f;  // deterministic padding
}

// The margin loop (src/writer.rs:1201-1207) for all pending nodes of the level, tile by tile.
// Output per tile: 32 side masks (bit i of mask o = side of item o + 32 i of the tile; 1 = Right) and the
// number of Left items; per node: n_left.
// Algorithmic traffic: 4*dims bytes per item (+4 B of permutation, +1 bit out).
template <int METRIC>
__global__ __launch_bounds__(kBlock) void k_forest_margin_f32(DataView dv, FNode *nodes, const FTile *__restrict__ tiles,
                                                              uint32_t n_tiles, const uint32_t *__restrict__ perm,
                                                              uint64_t n_items, const uint8_t *__restrict__ normals,
                                                              uint64_t nstride, uint64_t hdr_off,
                                                              uint64_t *__restrict__ masks,
                                                              uint32_t *__restrict__ tile_left) {
    extern __shared__ float4 s_n4[];
    __shared__ uint32_t s_left;
    const float *s_n = reinterpret_cast<const float *>(s_n4);
    const uint32_t o = threadIdx.x >> 3, j = threadIdx.x & 7u;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_PENDING) continue;  // block-uniform
        __syncthreads();
        const float4 *g_n4 = reinterpret_cast<const float4 *>(normals + tl.node * nstride);
        for (uint32_t i = threadIdx.x; i < (dv.pitch >> 2); i += blockDim.x) s_n4[i] = g_n4[i];
        if (threadIdx.x == 0) s_left = 0;
        __syncthreads();
        const float *g_h = reinterpret_cast<const float *>(normals + tl.node * nstride + hdr_off);
        const LeafHdr nh = {g_h[0], g_h[1]};
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        uint64_t mask = 0;
        uint32_t lefts = 0;
        for (uint32_t i = 0; i < 64; i++) {
            const uint32_t p = o + 32 * i;
            if (p >= in_tile) break;
            const uint64_t row = pp[p];
            const float m = margin_f32<METRIC>(dv, s_n, nh, row, j);
            const uint32_t side = side_of_margin(m);
            mask |= (uint64_t)side << i;
            lefts += side ^ 1u;
        }
        if (j == 0) {
            masks[(uint64_t)tile * 32 + o] = mask;
            if (lefts) atomicAdd(&s_left, lefts);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_left[tile] = s_left;
            if (s_left) atomicAdd(&nodes[tl.node].n_left, s_left);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_forest_margin_bq(DataView dv, FNode *nodes, const FTile *__restrict__ tiles,
                                                             uint32_t n_tiles, const uint32_t *__restrict__ perm,
                                                             uint64_t n_items, const uint8_t *__restrict__ normals,
                                                             uint64_t nstride, uint64_t hdr_off,
                                                             uint64_t *__restrict__ masks,
                                                             uint32_t *__restrict__ tile_left) {
    extern __shared__ uint64_t s_nw[];
    __shared__ uint32_t s_left;
    __shared__ uint8_t s_side[kTile];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_PENDING) continue;
        __syncthreads();
        const uint64_t *g_n = reinterpret_cast<const uint64_t *>(normals + tl.node * nstride);
        for (uint32_t i = threadIdx.x; i < dv.pitch; i += blockDim.x) s_nw[i] = g_n[i];
        if (threadIdx.x == 0) s_left = 0;
        __syncthreads();
        const float *g_h = reinterpret_cast<const float *>(normals + tl.node * nstride + hdr_off);
        const LeafHdr nh = {g_h[0], g_h[1]};
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x)
            s_side[p] = (uint8_t)side_of_margin(margin_bq(dv, s_nw, nh, pp[p]));
        __syncthreads();
        if (threadIdx.x < 32) {  // pack into the same mask layout as the f32 kernel
            uint64_t mask = 0;
            uint32_t lefts = 0;
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t p = threadIdx.x + 32 * i;
                if (p >= in_tile) break;
                mask |= (uint64_t)s_side[p] << i;
                lefts += s_side[p] ^ 1u;
            }
            masks[(uint64_t)tile * 32 + threadIdx.x] = mask;
            if (lefts) atomicAdd(&s_left, lefts);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_left[tile] = s_left;
            if (s_left) atomicAdd(&nodes[tl.node].n_left, s_left);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row-major margin pass (first attempt of a level, full-dataset trees, f32 metrics).
//
// Node-major tiles read every item row once PER TREE: T x N x 4*dims bytes of HBM per level.  But at a given
// level every tree partitions the same N rows, so one pass over the rows can serve several trees at once: an
// octet streams row r ONCE and, for each of up to TC trees, looks up the node that owns r in that tree
// (`node_of[t][r]`) and accumulates the dot with that node's normal.  The normals of one level
// (trees x nodes x 4*dims bytes) are small and re-used by thousands of rows, i.e. they come from L2 / the
// Infinity Cache, while HBM traffic drops to ceil(T / TC) x N x 4*dims bytes.  The arithmetic per (row, normal)
// pair is unchanged — the same 32-chain FMA order, the same reduction tree — so the sides are bit-identical to
// the node-major kernel; the host picks whichever mode moves fewer bytes for the level (deep levels, where few
// rows are still active and the normals no longer fit in cache, stay node-major).
// Output: one side byte per (tree, row); k_forest_masks_from_bytes turns them into the tile masks / counts the
// rest of the pipeline consumes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_forest_assign_node_of(const FNode *__restrict__ nodes,
                                                                  const FTile *__restrict__ tiles, uint32_t n_tiles,
                                                                  const uint32_t *__restrict__ perm, uint64_t n_items,
                                                                  uint32_t *__restrict__ node_of) {
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        uint32_t *dst = node_of + (uint64_t)nd->tree * n_items;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x) dst[pp[p]] = tl.node;
    }
}

// node_of for the next level, in row order: when the level before was row-major too, every (tree, row) pair already
// knows its node and its side, so its child is one table lookup away — a coalesced sweep instead of the scattered
// 4-byte writes of k_forest_assign_node_of (36 ms -> 2 ms per level at 10M x 100 trees).  `child[2 * node + side]` is
// the index of the child in the next level, or 0xFFFFFFFF when it is a leaf (or when the parent's sides were
// re-drawn after the row-major pass: those few nodes are then assigned the scattered way).
__global__ __launch_bounds__(kBlock) void k_forest_advance_node_of(uint32_t *__restrict__ node_of,
                                                                   const uint8_t *__restrict__ side_bytes,
                                                                   const uint32_t *__restrict__ child, uint64_t total) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; g < total; g += stride) {
        if (g + 4 <= total) {
            uint4 n = *reinterpret_cast<const uint4 *>(node_of + g);
            const uchar4 sd = *reinterpret_cast<const uchar4 *>(side_bytes + g);
            if (n.x != 0xFFFFFFFFu) n.x = child[2 * n.x + sd.x];
            if (n.y != 0xFFFFFFFFu) n.y = child[2 * n.y + sd.y];
            if (n.z != 0xFFFFFFFFu) n.z = child[2 * n.z + sd.z];
            if (n.w != 0xFFFFFFFFu) n.w = child[2 * n.w + sd.w];
            *reinterpret_cast<uint4 *>(node_of + g) = n;
        } else {
            for (uint64_t e = g; e < total; e++) {
                const uint32_t n = node_of[e];
                if (n != 0xFFFFFFFFu) node_of[e] = child[2 * n + side_bytes[e]];
            }
        }
    }
}

template <int METRIC, int TC>
__global__ __launch_bounds__(kBlock) void k_forest_margin_rows(DataView dv, const uint32_t *__restrict__ node_of,
                                                               uint32_t tree0, uint32_t n_pass,
                                                               const uint8_t *__restrict__ normals, uint64_t nstride,
                                                               uint64_t hdr_off, uint8_t *__restrict__ side_bytes) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.dims >> 5;
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < dv.n; row += n_octets) {
        const float *rp = dv.rows_f32 + row * dv.pitch;
        const float4 *r4 = reinterpret_cast<const float4 *>(rp) + j;
        const float4 *n4[TC];
        bool on[TC];
        float4 acc[TC];
#pragma unroll
        for (int t = 0; t < TC; t++) {
            uint32_t node = 0xFFFFFFFFu;
            if ((uint32_t)t < n_pass) node = node_of[(uint64_t)(tree0 + t) * dv.n + row];
            on[t] = node != 0xFFFFFFFFu;
            n4[t] = reinterpret_cast<const float4 *>(normals + (on[t] ? (uint64_t)node : 0ull) * nstride) + j;
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint32_t k = 0;
        for (; k + 8 <= blocks; k += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = ld_stream(r4 + (k + u) * 8);
#pragma unroll
            for (int t = 0; t < TC; t++) {
                if (on[t]) {
#pragma unroll
                    for (int u = 0; u < 8; u++) fma_step<OP_DOT>(acc[t], n4[t][(k + u) * 8], x[u]);
                }
            }
        }
        for (; k < blocks; k++) {
            const float4 x = r4[k * 8];
#pragma unroll
            for (int t = 0; t < TC; t++)
                if (on[t]) fma_step<OP_DOT>(acc[t], n4[t][k * 8], x);
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            if (on[t]) {
                const float *np = reinterpret_cast<const float *>(n4[t] - j);
                float d = octet_finish(acc[t]);
                d = scalar_tail<OP_DOT>(d, np, rp, blocks << 5, dv.dims);
                const float *nh = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(np) + hdr_off);
                float m = d;
                if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) m = f_add(nh[0], d);
                if (METRIC == AH_DOT_PRODUCT) m = f_add(d, f_mul(nh[0], dv.headers[2 * row]));
                if (j == 0) side_bytes[(uint64_t)(tree0 + t) * dv.n + row] = (uint8_t)side_of_margin(m);
            }
        }
    }
}

// Top levels of the forest: a group of trees has so few nodes that ALL its normals of the level fit in LDS (a level's
// nodes are ordered by tree, so a group owns the contiguous range [first_node, first_node + n_group_nodes)).  The
// row-major pass above is bound by the L1/L2 request rate of the normals (0.097 ns per margin at 16 trees); serving them
// from LDS leaves L1 to the row stream and the pass approaches the HBM time of the rows.  One big block per CU shares one
// copy of the normals: 1024 threads for 8 trees (114 VGPRs), 512 for 16 trees (the 64 accumulators need > 128 VGPRs);
// same arithmetic, bit-identical sides.
template <int METRIC, int TC>
__global__ __launch_bounds__(TC >= 16 ? 512 : 1024) void k_forest_margin_rows_lds(DataView dv, const uint32_t *__restrict__ node_of,
                                                                 uint32_t tree0, uint32_t n_pass,
                                                                 const uint8_t *__restrict__ normals, uint64_t nstride,
                                                                 uint64_t hdr_off, uint8_t *__restrict__ side_bytes,
                                                                 uint32_t first_node, uint32_t n_group_nodes) {
    extern __shared__ float4 s_norm4[];
    const uint32_t stride4 = (uint32_t)(nstride >> 4);  // record size in float4 (row bytes are a multiple of 128, + 16)
    {
        const float4 *g = reinterpret_cast<const float4 *>(normals + (uint64_t)first_node * nstride);
        const uint32_t total = n_group_nodes * stride4;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) s_norm4[i] = g[i];
    }
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.dims >> 5;
    const uint32_t hdr4 = (uint32_t)(hdr_off >> 4);
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < dv.n; row += n_octets) {
        const float *rp = dv.rows_f32 + row * dv.pitch;
        const float4 *r4 = reinterpret_cast<const float4 *>(rp) + j;
        uint32_t off[TC];  // float4 index of the normal record in LDS
        bool on[TC];
        float4 acc[TC];
#pragma unroll
        for (int t = 0; t < TC; t++) {
            uint32_t node = 0xFFFFFFFFu;
            if ((uint32_t)t < n_pass) node = node_of[(uint64_t)(tree0 + t) * dv.n + row];
            on[t] = node != 0xFFFFFFFFu;
            off[t] = (on[t] ? node - first_node : 0u) * stride4;
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint32_t k = 0;
        for (; k + 8 <= blocks; k += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = ld_stream(r4 + (k + u) * 8);
#pragma unroll
            for (int t = 0; t < TC; t++) {
                if (on[t]) {
#pragma unroll
                    for (int u = 0; u < 8; u++) fma_step<OP_DOT>(acc[t], s_norm4[off[t] + (k + u) * 8 + j], x[u]);
                }
            }
        }
        for (; k < blocks; k++) {
            const float4 x = r4[k * 8];
#pragma unroll
            for (int t = 0; t < TC; t++)
                if (on[t]) fma_step<OP_DOT>(acc[t], s_norm4[off[t] + k * 8 + j], x);
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            if (on[t]) {
                const float *np = reinterpret_cast<const float *>(s_norm4 + off[t]);
                float d = octet_finish(acc[t]);
                d = scalar_tail<OP_DOT>(d, np, rp, blocks << 5, dv.dims);
                const float *nh = reinterpret_cast<const float *>(s_norm4 + off[t] + hdr4);
                float m = d;
                if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) m = f_add(nh[0], d);
                if (METRIC == AH_DOT_PRODUCT) m = f_add(d, f_mul(nh[0], dv.headers[2 * row]));
                if (j == 0) side_bytes[(uint64_t)(tree0 + t) * dv.n + row] = (uint8_t)side_of_margin(m);
            }
        }
    }
}

// side bytes (by row) -> the per-tile masks / left counts of the node-major pipeline
__global__ __launch_bounds__(kBlock) void k_forest_masks_from_bytes(FNode *nodes, const FTile *__restrict__ tiles,
                                                                    uint32_t n_tiles, const uint32_t *__restrict__ perm,
                                                                    uint64_t n_items,
                                                                    const uint8_t *__restrict__ side_bytes,
                                                                    uint64_t *__restrict__ masks,
                                                                    uint32_t *__restrict__ tile_left) {
    __shared__ uint32_t s_left;
    __shared__ uint8_t s_side[kTile];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_PENDING) continue;
        __syncthreads();
        if (threadIdx.x == 0) s_left = 0;
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        const uint8_t *sb = side_bytes + (uint64_t)nd->tree * n_items;
        for (uint32_t p = threadIdx.x; p < in_tile; p += blockDim.x) s_side[p] = sb[pp[p]];
        __syncthreads();
        if (threadIdx.x < 32) {
            uint64_t mask = 0;
            uint32_t lefts = 0;
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t p = threadIdx.x + 32 * i;
                if (p >= in_tile) break;
                mask |= (uint64_t)s_side[p] << i;
                lefts += s_side[p] ^ 1u;
            }
            masks[(uint64_t)tile * 32 + threadIdx.x] = mask;
            if (lefts) atomicAdd(&s_left, lefts);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_left[tile] = s_left;
            if (s_left) atomicAdd(&nodes[tl.node].n_left, s_left);
        }
    }
}

// split_imbalance (src/writer.rs:1348-1353, f64) and the accept / retry / random decision (:1209-1227).
__global__ void k_forest_decide(FNode *nodes, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    FNode &nd = nodes[i];
    if (nd.state != ST_PENDING) return;
    const double ls = (double)nd.n_left, rs = (double)(nd.count - nd.n_left);
    const double f = ls / (ls + rs + 2.220446049250313e-16);
    const double g = 1.0 - f;
    const double imb = f > g ? f : g;
    if (imb < 0.95 || nd.attempt == 3) {
        if (imb > 0.99) {
            nd.state = ST_RANDOM;
            nd.n_left = 0;
        } else {
            nd.state = ST_ACCEPTED;
        }
    } else {
        nd.attempt += 1;  // remaining_attempts -= 1; n_left is reset by the next create_split
    }
}

// randomly_split_children (src/writer.rs:1310-1326) with the policy coin, same mask layout.
__global__ __launch_bounds__(64) void k_forest_random_sides(FNode *nodes, const FTile *__restrict__ tiles,
                                                            uint32_t n_tiles, uint64_t *__restrict__ masks,
                                                            uint32_t *__restrict__ tile_left) {
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_RANDOM) continue;
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        uint32_t lefts = 0;
        if (threadIdx.x < 32) {
            uint64_t mask = 0;
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t p = threadIdx.x + 32 * i;
                if (p >= in_tile) break;
                const uint32_t left = ah_random_side_is_left(nd->key, (uint64_t)tl.first + p);
                mask |= (uint64_t)(left ^ 1u) << i;
                lefts += left;
            }
            masks[(uint64_t)tile * 32 + threadIdx.x] = mask;
        }
        for (int off = 32; off > 0; off >>= 1) lefts += __shfl_down(lefts, off);
        if (threadIdx.x == 0) {
            tile_left[tile] = lefts;
            if (lefts) atomicAdd(&nodes[tl.node].n_left, lefts);
        }
    }
}

// Exclusive scan of the per-tile left counts inside every node: one wave per node.
__global__ __launch_bounds__(64) void k_forest_tile_offsets(const FNode *__restrict__ nodes, uint32_t n_nodes,
                                                            const uint32_t *__restrict__ tile_left,
                                                            uint32_t *__restrict__ tile_left_off) {
    for (uint32_t node = blockIdx.x; node < n_nodes; node += gridDim.x) {
        const FNode nd = nodes[node];
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nd.n_tiles; base += 64) {
            const uint32_t t = base + threadIdx.x;
            const uint32_t v = t < nd.n_tiles ? tile_left[nd.tile_begin + t] : 0u;
            uint32_t incl = v;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if ((int)threadIdx.x >= off) incl += up;
            }
            if (t < nd.n_tiles) tile_left_off[nd.tile_begin + t] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
}

// Stable partition of every split node into its children (ascending order kept inside each child =
// RoaringBitmap::from_sorted_iter, src/writer.rs:1230-1231).  A child that fits in a Descendants node
// (count <= split_after, :474-477) is written to `final_perm`, the others to the next level's permutation.
__global__ __launch_bounds__(kBlock) void k_forest_scatter(const FNode *__restrict__ nodes,
                                                           const FTile *__restrict__ tiles, uint32_t n_tiles,
                                                           const uint32_t *__restrict__ perm_cur,
                                                           uint32_t *__restrict__ perm_next,
                                                           uint32_t *__restrict__ final_perm, uint64_t n_items,
                                                           const uint64_t *__restrict__ masks,
                                                           const uint32_t *__restrict__ tile_left_off,
                                                           uint32_t split_after) {
    __shared__ uint64_t s_masks[32];
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode nd = nodes[tl.node];
        __syncthreads();
        if (threadIdx.x < 32) s_masks[threadIdx.x] = masks[(uint64_t)tile * 32 + threadIdx.x];
        __syncthreads();
        const uint32_t in_tile = min(kTile, nd.count - tl.first);
        const uint32_t p0 = threadIdx.x * 8;
        const uint32_t i = p0 >> 5;
        const uint32_t nvalid = p0 < in_tile ? min(8u, in_tile - p0) : 0u;
        uint32_t sidebits = 0;
        for (uint32_t e = 0; e < nvalid; e++) sidebits |= (uint32_t)((s_masks[(p0 + e) & 31u] >> i) & 1ull) << e;
        const uint32_t my_left = nvalid - (uint32_t)__popc(sidebits);
        // block-wide exclusive scan of my_left
        uint32_t incl = my_left;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int)lane >= off) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0;
        for (uint32_t w = 0; w < wave; w++) wave_base += s_wave[w];
        uint32_t left_before = tile_left_off[tile] + wave_base + incl - my_left;  // lefts before p0, inside the node
        uint32_t right_before = (tl.first + p0) - left_before;
        const uint64_t base = nd.start;
        const uint32_t n_right = nd.count - nd.n_left;
        uint32_t *dst_l = (nd.n_left <= split_after ? final_perm : perm_next) + base;
        uint32_t *dst_r = (n_right <= split_after ? final_perm : perm_next) + base + nd.n_left;
        const uint32_t *src = perm_cur + base + tl.first + p0;
        for (uint32_t e = 0; e < nvalid; e++) {
            const uint32_t row = src[e];
            if ((sidebits >> e) & 1u) dst_r[right_before++] = row;
            else dst_l[left_before++] = row;
        }
    }
}

__global__ void k_rows_to_ids(uint32_t *perm, uint64_t total, const uint32_t *__restrict__ ids) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) perm[g] = ids[perm[g]];
}

}  // namespace ah

using namespace ah;

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct ah_forest {
    std::vector<uint32_t> roots;
    std::vector<ah_node> nodes;
    uint8_t *normals = nullptr;  // raw (uninitialised) buffers: filled by D2H copies only, never repacked
    uint64_t normals_len = 0;
    uint32_t *descendants = nullptr;
    uint64_t descendants_len = 0;
    uint64_t normal_stride = 0, normal_vector_offset = 0, normal_header_offset = 0;
    ah_build_stats stats{};
    ~ah_forest() {
        free(normals);
        free(descendants);
    }
};

namespace {

struct HostRec {  // one tree node, in creation (breadth-first) order
    uint8_t kind, has_normal;
    uint32_t tree;
    uint32_t left = 0, right = 0;  // HostRec indices
    uint64_t start;
    uint32_t count;
    uint32_t depth;
    uint64_t normal_off = 0;  // byte offset of the normal record inside the forest's normals buffer
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return AH_OK;
        if (p) AH_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)1024);
        AH_HIP(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return AH_OK;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

struct LevelChunk {  // the normal records of one level stay in HBM until the end of the batch
    uint8_t *d = nullptr;
    uint64_t bytes = 0;
    uint64_t host_off = 0;
};

struct EventPair {
    hipEvent_t a, b;
};

// AH_DEBUG=1: synchronise after every launch and say which kernel finished (debugging aid only)
bool g_debug = getenv("AH_DEBUG") != nullptr;
// AH_ROWMAJOR=0 disables the row-major margin pass, =1 forces it whenever it is legal (A/B measurements);
// AH_ROWMAJOR_CACHE_MB = budget for one group's normals (default 6.5 MB: per-level traces of the 10M x 768 build show
// the per-margin cost of a pass rising from 0.10-0.16 ns below it to 0.27-0.5 ns at 12.6 MB, see build_batch).
int g_rows_force = getenv("AH_ROWMAJOR") ? atoi(getenv("AH_ROWMAJOR")) : -1;
// grid caps of the two margin kernels (grid-stride beyond them).  One tile / 32 rows per block measured 2 % faster
// than a 4096-block persistent grid on the 10M x 768 build (3.55 vs 3.63 s of device time).
uint32_t g_tile_blocks = getenv("AH_FOREST_TILE_BLOCKS") ? (uint32_t)atoi(getenv("AH_FOREST_TILE_BLOCKS")) : (1u << 20);
uint32_t g_row_blocks = getenv("AH_FOREST_ROW_BLOCKS") ? (uint32_t)atoi(getenv("AH_FOREST_ROW_BLOCKS")) : (1u << 20);
bool g_rows_advance = !(getenv("AH_ROWMAJOR_ADVANCE") && atoi(getenv("AH_ROWMAJOR_ADVANCE")) == 0);  // A/B switch
bool g_rows_lds = !(getenv("AH_ROWMAJOR_LDS") && atoi(getenv("AH_ROWMAJOR_LDS")) == 0);              // A/B switch
constexpr size_t kLdsNormalsBytes = 128u << 10;  // LDS given to the normals of one tree group (of 160 KiB per CU)
uint32_t g_rows_max_tc = getenv("AH_ROWMAJOR_MAX_TC") ? (uint32_t)atoi(getenv("AH_ROWMAJOR_MAX_TC")) : 16u;
uint64_t g_rows_cache_bytes = (uint64_t)((getenv("AH_ROWMAJOR_CACHE_MB") ? atof(getenv("AH_ROWMAJOR_CACHE_MB")) : 6.5) * 1e6);
#define AH_DBG(s, what)                                                       \
    do {                                                                      \
        if (g_debug) {                                                        \
            hipError_t _e = hipStreamSynchronize(s);                          \
            fprintf(stderr, "[ah] %s: %s\n", what, hipGetErrorString(_e));    \
            fflush(stderr);                                                   \
        }                                                                     \
    } while (0)

struct BatchCleanup {
    std::vector<LevelChunk> chunks;
    std::vector<EventPair> events;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    ~BatchCleanup() {
        for (LevelChunk &c : chunks)
            if (c.d) (void)hipFree(c.d);
        for (EventPair &e : events) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
    }
};

}  // namespace

// `subset_ids` == nullptr: every tree covers all items (Writer::build with missing trees).  Otherwise tree t covers
// the ascending id list subset_ids[subset_offsets[first_tree + t] .. subset_offsets[first_tree + t + 1]) — the
// "descendants that became too large" of an incremental build (src/writer.rs:660-739).
// Device -> pageable host copies off the build's critical path.  hipMemcpy into pageable memory is staged by the
// runtime on one thread and pays the first-touch page faults of the fresh destination there (measured: 4 GB of item
// ids in 0.60 s = 6.7 GB/s, whatever the number of concurrent calls).  Instead a worker thread owns a pinned double
// buffer: the DMA engine fills one half while a few threads copy the other half to its final place, and the level
// loop never waits for it — each level's normals travel while the next levels are computed.
struct Readback {
    struct Job {
        void *dst;
        const void *src;
        size_t bytes;
    };
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    size_t pending = 0;
    bool stop = false, started = false;
    hipError_t err = hipSuccess;
    int device = 0;
    uint8_t *pin = nullptr;  // 2 x half bytes of pinned memory (owned by the build's Context)
    size_t half = 0;

    void start(int dev, void *pinned, size_t pinned_bytes) {
        device = dev;
        pin = reinterpret_cast<uint8_t *>(pinned);
        half = pinned_bytes / 2;
        started = true;
        th = std::thread([this] { run(); });
    }
    static void spread(uint8_t *dst, const uint8_t *src, size_t bytes) {  // memcpy on up to 8 cores
        const size_t n_threads = std::min<size_t>(8, bytes >> 20);
        if (n_threads <= 1) {
            memcpy(dst, src, bytes);
            return;
        }
        std::vector<std::thread> pool;
        const size_t per = ((bytes + n_threads - 1) / n_threads + 4095) & ~(size_t)4095;
        for (size_t lo = 0; lo < bytes; lo += per) {
            const size_t len = std::min(per, bytes - lo);
            pool.emplace_back([=] { memcpy(dst + lo, src + lo, len); });
        }
        for (auto &t : pool) t.join();
    }
    hipError_t copy(const Job &job, hipStream_t cs, hipEvent_t *ev) {
        static const bool direct = getenv("AH_READBACK_DIRECT") != nullptr;  // A/B: let the runtime stage the copy
        if (direct) {
            hipError_t e = hipMemcpyAsync(job.dst, job.src, job.bytes, hipMemcpyDeviceToHost, cs);
            return e == hipSuccess ? hipStreamSynchronize(cs) : e;
        }
        const uint8_t *src = reinterpret_cast<const uint8_t *>(job.src);
        uint8_t *dst = reinterpret_cast<uint8_t *>(job.dst);
        size_t issued = 0, landed = 0, prev_len = 0;
        int b = 0;
        hipError_t e = hipSuccess;
        while (landed < job.bytes && e == hipSuccess) {
            size_t len = 0;
            if (issued < job.bytes) {
                len = std::min(half, job.bytes - issued);
                e = hipMemcpyAsync(pin + (size_t)b * half, src + issued, len, hipMemcpyDeviceToHost, cs);
                if (e == hipSuccess) e = hipEventRecord(ev[b], cs);
                issued += len;
            }
            if (prev_len && e == hipSuccess) {  // the other half is in flight while this one is spread out
                e = hipEventSynchronize(ev[b ^ 1]);
                if (e == hipSuccess) spread(dst + landed, pin + (size_t)(b ^ 1) * half, prev_len);
                landed += prev_len;
            }
            prev_len = len;
            b ^= 1;
        }
        return e;
    }
    void run() {
        hipStream_t cs = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) break;
                job = q.front();
                q.pop_front();
            }
            if (e == hipSuccess) e = copy(job, cs, ev);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (e != hipSuccess && err == hipSuccess) err = e;
                pending--;
            }
            cv.notify_all();
        }
        if (ev[0]) (void)hipEventDestroy(ev[0]);
        if (ev[1]) (void)hipEventDestroy(ev[1]);
        if (cs) (void)hipStreamDestroy(cs);
    }
    void push(void *dst, const void *src, size_t bytes) {
        if (!bytes) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(Job{dst, src, bytes});
            pending++;
        }
        cv.notify_all();
    }
    hipError_t drain() {  // every pushed copy has landed
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return pending == 0; });
        return err;
    }
    ~Readback() {
        if (!started) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
};

static int build_batch(ah_dataset *ds, const ah_build_options *opt, uint32_t first_tree, uint32_t n_trees,
                       uint32_t split_after, ah_forest *forest, Context *ctx, const uint32_t *subset_ids,
                       const uint64_t *subset_offsets) {
    const uint64_t N = ds->n;
    const auto t_batch = std::chrono::steady_clock::now();
    std::vector<uint64_t> tree_base(n_trees + 1, 0);
    for (uint32_t t = 0; t < n_trees; t++)
        tree_base[t + 1] = tree_base[t] + (subset_ids ? subset_offsets[first_tree + t + 1] - subset_offsets[first_tree + t] : N);
    const uint64_t M = tree_base[n_trees];  // entries of the batch permutation
    const DataView dv = ds->view();
    hipStream_t s = ctx->stream;
    const bool bq = metric_is_bq(ds->metric);
    const uint64_t hdr_off = ds->row_bytes();
    const uint64_t nstride = hdr_off + 16;

    // Upper bounds known up front (every split node owns > split_after items), so nothing is reallocated
    // between levels: nodes per level <= n_trees * N / (split_after + 1), tiles <= items / kTile + nodes.
    const uint64_t max_nodes = M / ((uint64_t)split_after + 1) + n_trees;
    const uint64_t max_tiles = M / kTile + n_trees + max_nodes;
    DevBuf<uint32_t> perm_a, perm_b, final_perm, tile_left, tile_left_off;
    DevBuf<FNode> d_nodes;
    DevBuf<FTile> d_tiles;
    DevBuf<uint64_t> masks;
    AH_TRY(perm_a.ensure(M));
    AH_TRY(perm_b.ensure(M));
    AH_TRY(final_perm.ensure(M));
    AH_TRY(d_nodes.ensure(max_nodes));
    AH_TRY(d_tiles.ensure(max_tiles));
    AH_TRY(masks.ensure(max_tiles * 32));
    AH_TRY(tile_left.ensure(max_tiles));
    AH_TRY(tile_left_off.ensure(max_tiles));
    const size_t kBounce = 64ull << 20;  // pinned double buffer of the read-back worker
    const size_t pin_tables = (max_nodes * sizeof(FNode) + max_tiles * sizeof(FTile) + 4096 + 4095) & ~(size_t)4095;
    AH_TRY(ctx->ensure_pinned(pin_tables + kBounce));
    // row-major margin mode (full-dataset trees, f32 metrics): node index and side byte per (tree, row)
    const bool rows_allowed = !subset_ids && !bq && ds->dims >= 32 && n_trees >= 2 && g_rows_force != 0;
    DevBuf<uint32_t> node_of, d_child;
    DevBuf<FTile> d_fix_tiles;
    DevBuf<uint8_t> side_bytes;
    if (rows_allowed) {
        AH_TRY(node_of.ensure((size_t)n_trees * N + 4));
        AH_TRY(side_bytes.ensure((size_t)n_trees * N + 4));
        AH_TRY(d_child.ensure(2 * max_nodes));
    }
    // state carried from one level to the next for k_forest_advance_node_of
    bool prev_rows = false;            // the previous level ran row-major: node_of / side_bytes describe it
    std::vector<uint32_t> h_child;     // per node of the previous level: index of its two children in this level
    std::vector<uint32_t> fix_nodes;   // nodes of this level whose parent's sides were re-drawn (retry / random)
    FNode *h_nodes = reinterpret_cast<FNode *>(ctx->h_pinned);
    FTile *h_tiles = reinterpret_cast<FTile *>(h_nodes + max_nodes);
    if (!subset_ids) {
        hipLaunchKernelGGL(k_init_perm, dim3(2048), dim3(256), 0, s, perm_a.p, N, n_trees);
    } else if (M) {
        const uint32_t *src = subset_ids + subset_offsets[first_tree];
        AH_HIP(hipMemcpyAsync(perm_a.p, src, M * 4, hipMemcpyHostToDevice, s));
        DevBuf<uint32_t> d_err;
        AH_TRY(d_err.ensure(1));
        AH_HIP(hipMemsetAsync(d_err.p, 0, 4, s));
        hipLaunchKernelGGL(k_ids_to_rows, dim3(2048), dim3(256), 0, s, dv, perm_a.p, M, d_err.p);
        uint32_t e = 0;
        AH_HIP(hipMemcpyAsync(&e, d_err.p, 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipStreamSynchronize(s));
        AH_REQUIRE(e == 0, AH_ERR_MISSING_ITEM, "a sub-tree item id does not exist in the dataset");
    }
    AH_HIP(hipGetLastError());

    // The item-id lists come back in one piece at the very end; their size is known now, so the host memory is
    // allocated up front and its pages are touched in the background while the GPU works (first-touch faults of
    // fresh memory, not the copy, bound a read-back of several GB).
    const uint64_t desc_base = forest->descendants_len;
    {
        uint32_t *grown = (uint32_t *)realloc(forest->descendants, (desc_base + M) * 4 + 16);
        AH_REQUIRE(grown, AH_ERR_OUT_OF_MEMORY, "host allocation of the descendants failed");
        forest->descendants = grown;
    }
    struct Toucher {  // commits the pages of a fresh (still unwritten) host range in the background
        std::thread th;
        void start(void *ptr, size_t bytes) {
            join();
            if (bytes < (8u << 20)) return;
            uint8_t *lo = reinterpret_cast<uint8_t *>(ptr);
            th = std::thread([lo, bytes] {
                const size_t parts = std::min<size_t>(4, std::max<size_t>(1, bytes >> 26));
                std::vector<std::thread> pool;
                for (size_t p = 0; p < parts; p++)
                    pool.emplace_back([=] {
                        const size_t a = bytes * p / parts, b = bytes * (p + 1) / parts;
                        for (size_t off = a; off < b; off += 4096) reinterpret_cast<volatile uint8_t *>(lo)[off] = 0;
                    });
                for (auto &t : pool) t.join();
            });
        }
        void join() {
            if (th.joinable()) th.join();
        }
        ~Toucher() { join(); }
    } prefault, prefault_normals;
    prefault.start(forest->descendants + desc_base, M * 4);
    BatchCleanup bc;
    Readback rb;  // declared after `bc`: joined before the level chunks it reads are freed
    rb.start(ds->device, reinterpret_cast<uint8_t *>(ctx->h_pinned) + pin_tables, kBounce);
    std::vector<HostRec> recs;
    std::vector<FNode> level;  // active (to be split) nodes of the current level
    std::vector<uint32_t> tree_root(n_trees);
    recs.reserve(4 * max_nodes / 3 + 16);
    for (uint32_t t = 0; t < n_trees; t++) {
        const uint32_t cnt = (uint32_t)(tree_base[t + 1] - tree_base[t]);
        HostRec r{};
        r.tree = t;
        r.start = tree_base[t];
        r.count = cnt;
        r.depth = 0;
        tree_root[t] = (uint32_t)recs.size();
        if (cnt <= split_after) {  // fit_in_descendant at the root: the tree is one Descendants node
            r.kind = AH_NODE_DESCENDANTS;
            recs.push_back(r);
            continue;
        }
        r.kind = AH_NODE_SPLIT;
        FNode nd{};
        nd.key = ah_node_key_root(opt->tree_seeds[first_tree + t]);
        nd.start = tree_base[t];
        nd.tree = t;
        nd.count = cnt;
        nd.rec = (uint32_t)recs.size();
        recs.push_back(r);
        level.push_back(nd);
    }

    uint32_t *cur = perm_a.p, *nxt = perm_b.p;
    AH_HIP(hipEventCreate(&bc.ev_begin));
    AH_HIP(hipEventCreate(&bc.ev_end));
    AH_HIP(hipEventRecord(bc.ev_begin, s));
    const size_t cs_shared = (size_t)f32_space_pitch(ds->metric, ds->dims) * 4 * 3;
    AH_REQUIRE(cs_shared <= 150 * 1024, AH_ERR_INVALID_DIMENSION, "dimensions %u too large for the LDS-resident two-means",
               ds->dims);
    if (cs_shared > 48 * 1024)
        AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_create_split),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs_shared));
    const uint64_t normals_base = forest->normals_len;  // this batch appends its levels after earlier batches
    uint64_t normals_bytes = 0;
    // The host blob gets lazily committed head-room (splits in total ~1.3-1.6 x items / split_after) so that finished
    // levels can land in their final place while the build continues; it only moves when no copy is in flight.
    uint64_t normals_cap = forest->normals_len;
    auto reserve_normals = [&](uint64_t need) -> int {
        if (need <= normals_cap) return AH_OK;
        AH_REQUIRE(rb.drain() == hipSuccess, AH_ERR_DEVICE, "device -> host copy of the normals failed");
        uint64_t want = std::max<uint64_t>(need + need / 2, normals_base + 2 * max_nodes * nstride);
        uint8_t *grown = (uint8_t *)realloc(forest->normals, want + 16);
        if (!grown) {
            want = need;
            grown = (uint8_t *)realloc(forest->normals, want + 16);
        }
        AH_REQUIRE(grown, AH_ERR_OUT_OF_MEMORY, "host allocation of %llu bytes of normals failed", (unsigned long long)want);
        forest->normals = grown;
        normals_cap = want;
        return AH_OK;
    };
    uint32_t depth = 0;
    uint64_t items_routed = 0;
    while (!level.empty()) {
        if (opt->cancel && *opt->cancel) {
            set_error("build cancelled");
            return AH_ERR_CANCELLED;  // Error::BuildCancelled, polled per level (src/writer.rs:1178,1196)
        }
        AH_REQUIRE(depth < 100000, AH_ERR_DEVICE, "forest build: depth %u exceeded (internal error)", depth);
        const uint32_t n_nodes = (uint32_t)level.size();
        AH_REQUIRE(n_nodes <= max_nodes, AH_ERR_DEVICE, "forest build: node bound exceeded (internal error)");
        uint32_t n_tiles = 0;
        for (uint32_t i = 0; i < n_nodes; i++) {
            FNode &nd = level[i];
            nd.tile_begin = n_tiles;
            nd.n_tiles = (nd.count + kTile - 1) / kTile;
            AH_REQUIRE((uint64_t)n_tiles + nd.n_tiles <= max_tiles, AH_ERR_DEVICE, "forest build: tile bound exceeded");
            for (uint32_t t = 0; t < nd.n_tiles; t++) h_tiles[n_tiles++] = FTile{i, t * kTile};
            h_nodes[i] = nd;
        }
        LevelChunk chunk;
        chunk.bytes = (uint64_t)n_nodes * nstride;
        chunk.host_off = normals_base + normals_bytes;
        AH_HIP(hipMalloc((void **)&chunk.d, chunk.bytes));
        bc.chunks.push_back(chunk);
        normals_bytes += chunk.bytes;
        // host side of this level's normals: reserved now and page-touched while the level is computed
        prefault_normals.join();
        AH_TRY(reserve_normals(chunk.host_off + chunk.bytes));
        prefault_normals.start(forest->normals + chunk.host_off, chunk.bytes);
        AH_HIP(hipMemcpyAsync(d_nodes.p, h_nodes, n_nodes * sizeof(FNode), hipMemcpyHostToDevice, s));
        AH_HIP(hipMemcpyAsync(d_tiles.p, h_tiles, n_tiles * sizeof(FTile), hipMemcpyHostToDevice, s));
        const unsigned tile_grid = std::min<uint32_t>(n_tiles, g_tile_blocks);
        // Row-major or node-major for the first attempt of this level?  Row-major streams all N rows once per group of
        // row_tc trees; node-major reads only the still-active items, once per tree.  The group is sized so that the
        // level's normals of one group stay cache-resident.
        uint32_t row_tc = 0;
        if (rows_allowed) {
            // Group size: the largest row_tc whose normals for this level (ws) fit the cache budget (6.5 MB: beyond it
            // the normals spill from the XCD L2s to the Infinity Cache and every extra tree costs more than it saves).
            // Row-major or not is then decided by a cost model in ns, fitted to per-level rocprofv3 traces of the
            // 10M x 768 x 100-tree build and scaled by the row size:
            //   node-major  0.47 per (item, tree) pair: one HBM read of the row at ~6.85 TB/s;
            //   row-major   per pass and row: base(row_tc) = 0.60 / 0.87 / 1.18 / 1.55 for 2 / 4 / 8 / 16 trees (the HBM
            //               read of the row + normals served by L1/L2) + row_tc x 0.018 per MB of ws beyond 2.7 MB,
            //               scaled by the share of (row, tree) pairs that are still splitting; plus the node_of / mask
            //               conversion of the level (0.01-0.05 per (row, tree); 0.007-0.02 when node_of is advanced in
            //               row order from the previous row-major level).
            uint64_t pairs = 0;
            for (uint32_t i = 0; i < n_nodes; i++) pairs += level[i].count;
            const double row_b = (double)ds->row_bytes();
            const double active = (double)pairs / ((double)n_trees * (double)N);  // share of (row, tree) pairs still splitting
            const uint64_t nodes_per_tree = (n_nodes + n_trees - 1) / n_trees;
            const double scale = row_b / 3072.0;
            const double cost_node = (double)pairs * 0.47 * scale;
            const double convert = (double)n_trees * (double)N *
                                   ((prev_rows && g_rows_advance ? 0.007 : 0.01 + 0.03 * std::min(1.0, (double)nodes_per_tree / 256.0)) +
                                    0.012 * std::min(1.0, (double)nodes_per_tree / 512.0));
            uint32_t tc = g_rows_max_tc;
            while (tc > 1 && (uint64_t)tc * nodes_per_tree * nstride > g_rows_cache_bytes) tc >>= 1;
            while (tc > 2 && tc / 2 >= n_trees) tc >>= 1;  // do not instantiate more slots than trees
            // measured anomaly: the 8-tree instantiation is slower per margin than the 4-tree one as soon as its
            // normals leave the L2 (0.25-0.31 vs 0.22-0.27 ns at 3.2-6.3 MB)
            if (tc == 8 && (double)((uint64_t)tc * nodes_per_tree * nstride) > 2.7e6) tc = 4;
            if (tc >= 2) {
                const double passes = (double)((n_trees + tc - 1) / tc);
                const double base = tc >= 16 ? 1.55 : tc == 8 ? 1.18 : tc == 4 ? 0.87 : 0.60;
                const double ws_mb = (double)((uint64_t)tc * nodes_per_tree * nstride) / 1e6;
                const double per_row = 0.45 + ((base - 0.45) + tc * 0.018 * std::max(0.0, ws_mb - 2.7)) * std::min(1.0, active);
                const double cost_rows = passes * (double)N * per_row * scale + convert;
                if (g_rows_force == 1 || cost_rows < 0.95 * cost_node) row_tc = tc;
            }
        }
        // Top levels: all normals of a group of >= 8 trees fit in LDS -> the LDS-resident variant of the row-major pass.
        uint32_t lds_tc = 0;
        std::vector<uint32_t> tree_first;  // first node of every tree in this level (nodes are ordered by tree)
        if (rows_allowed && g_rows_lds && row_tc >= 2 && (hdr_off & 15) == 0) {
            tree_first.assign(n_trees + 1, n_nodes);
            bool ordered = true;
            for (uint32_t i = n_nodes; i-- > 0;) {
                tree_first[level[i].tree] = i;
                if (i + 1 < n_nodes && level[i].tree > level[i + 1].tree) ordered = false;
            }
            for (uint32_t t = n_trees; t-- > 0;) tree_first[t] = std::min(tree_first[t], tree_first[t + 1]);
            for (uint32_t tc = std::min<uint32_t>(16, g_rows_max_tc); ordered && tc >= 8; tc >>= 1) {
                uint32_t worst = 0;
                for (uint32_t t0 = 0; t0 < n_trees; t0 += tc)
                    worst = std::max(worst, tree_first[std::min(n_trees, t0 + tc)] - tree_first[t0]);
                if ((uint64_t)worst * nstride <= kLdsNormalsBytes) {
                    lds_tc = tc;
                    break;
                }
            }
            if (lds_tc) row_tc = lds_tc;
        }
        for (int attempt = 0; attempt < 4; attempt++) {
            hipLaunchKernelGGL(k_forest_create_split, dim3(n_nodes), dim3(64), cs_shared, s, dv, d_nodes.p, cur, N,
                               chunk.d, nstride, hdr_off);
            AH_DBG(s, "create_split");
            EventPair ep;
            AH_HIP(hipEventCreate(&ep.a));
            AH_HIP(hipEventCreate(&ep.b));
            bc.events.push_back(ep);
            AH_HIP(hipEventRecord(ep.a, s));
            if (attempt == 0 && row_tc >= 2) {
                // one pass over the rows serves up to row_tc trees (see k_forest_margin_rows)
                if (prev_rows && g_rows_advance) {
                    AH_HIP(hipMemcpyAsync(d_child.p, h_child.data(), h_child.size() * 4, hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(k_forest_advance_node_of, dim3(kMaxBlocks), dim3(kBlock), 0, s, node_of.p, side_bytes.p,
                                       d_child.p, (uint64_t)n_trees * N);
                    if (!fix_nodes.empty()) {
                        std::vector<FTile> fix;
                        for (uint32_t i : fix_nodes)
                            for (uint32_t t = 0; t < level[i].n_tiles; t++) fix.push_back(FTile{i, t * kTile});
                        AH_TRY(d_fix_tiles.ensure(fix.size()));
                        AH_HIP(hipMemcpyAsync(d_fix_tiles.p, fix.data(), fix.size() * sizeof(FTile), hipMemcpyHostToDevice, s));
                        hipLaunchKernelGGL(k_forest_assign_node_of, dim3(std::min<uint32_t>((uint32_t)fix.size(), kMaxBlocks)),
                                           dim3(kBlock), 0, s, d_nodes.p, d_fix_tiles.p, (uint32_t)fix.size(), cur, N, node_of.p);
                        AH_HIP(hipStreamSynchronize(s));  // `fix` is a pageable staging vector
                    }
                } else {
                    AH_HIP(hipMemsetAsync(node_of.p, 0xFF, (size_t)n_trees * N * 4, s));
                    hipLaunchKernelGGL(k_forest_assign_node_of, dim3(tile_grid), dim3(kBlock), 0, s, d_nodes.p, d_tiles.p,
                                       n_tiles, cur, N, node_of.p);
                }
                const unsigned row_grid = (unsigned)std::min<uint64_t>((N + 31) / 32, g_row_blocks);
                for (uint32_t t0 = 0; t0 < n_trees && lds_tc; t0 += lds_tc) {
                    const uint32_t np = std::min<uint32_t>(lds_tc, n_trees - t0);
                    const uint32_t first = tree_first[t0], cnt = tree_first[t0 + np] - first;
                    if (cnt == 0) continue;
                    const size_t sh = (size_t)cnt * nstride;
                    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(2, (150u << 10) / std::max<size_t>(sh, 1)));
                    const unsigned lthreads = lds_tc >= 16 ? 512u : 1024u;
                    const unsigned lgrid = (unsigned)std::min<uint64_t>((N * 8 + lthreads - 1) / lthreads, 256u * per_cu);
#define AH_ROWS_LDS(M, TCV)                                                                                             \
    do {                                                                                                                \
        static std::atomic<bool> lds_opt_in[64]; /* once per instantiation and device: the call waits for the stream */ \
        if (!lds_opt_in[ds->device & 63].load(std::memory_order_acquire)) {                                                                             \
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_margin_rows_lds<M, TCV>),                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsNormalsBytes));             \
            lds_opt_in[ds->device & 63].store(true, std::memory_order_release);                                         \
        }                                                                                                               \
        hipLaunchKernelGGL((k_forest_margin_rows_lds<M, TCV>), dim3(lgrid), dim3(lthreads), sh, s, dv, node_of.p, t0, np, \
                           chunk.d, nstride, hdr_off, side_bytes.p, first, cnt);                                        \
    } while (0)
#define AH_ROWS_LDS_TC(M)                  \
    if (lds_tc == 16) AH_ROWS_LDS(M, 16);  \
    else AH_ROWS_LDS(M, 8)
                    switch (ds->metric) {
                    case AH_EUCLIDEAN: AH_ROWS_LDS_TC(AH_EUCLIDEAN); break;
                    case AH_MANHATTAN: AH_ROWS_LDS_TC(AH_MANHATTAN); break;
                    case AH_COSINE: AH_ROWS_LDS_TC(AH_COSINE); break;
                    default: AH_ROWS_LDS_TC(AH_DOT_PRODUCT); break;
                    }
#undef AH_ROWS_LDS_TC
#undef AH_ROWS_LDS
                }
                for (uint32_t t0 = 0; t0 < n_trees && !lds_tc; t0 += row_tc) {
                    const uint32_t np = std::min<uint32_t>(row_tc, n_trees - t0);
#define AH_ROWS(M, TCV)                                                                                          \
    hipLaunchKernelGGL((k_forest_margin_rows<M, TCV>), dim3(row_grid), dim3(kBlock), 0, s, dv, node_of.p, t0, np, \
                       chunk.d, nstride, hdr_off, side_bytes.p)
#define AH_ROWS_TC(M)                       \
    switch (row_tc) {                       \
    case 16: AH_ROWS(M, 16); break;         \
    case 8: AH_ROWS(M, 8); break;           \
    case 4: AH_ROWS(M, 4); break;           \
    default: AH_ROWS(M, 2); break;          \
    }
                    switch (ds->metric) {
                    case AH_EUCLIDEAN: AH_ROWS_TC(AH_EUCLIDEAN); break;
                    case AH_MANHATTAN: AH_ROWS_TC(AH_MANHATTAN); break;
                    case AH_COSINE: AH_ROWS_TC(AH_COSINE); break;
                    default: AH_ROWS_TC(AH_DOT_PRODUCT); break;
                    }
#undef AH_ROWS_TC
#undef AH_ROWS
                }
                hipLaunchKernelGGL(k_forest_masks_from_bytes, dim3(tile_grid), dim3(kBlock), 0, s, d_nodes.p, d_tiles.p, n_tiles,
                                   cur, N, side_bytes.p, masks.p, tile_left.p);
                forest->stats.margin_row_passes += (n_trees + row_tc - 1) / row_tc;
            } else if (bq) {
                hipLaunchKernelGGL(k_forest_margin_bq, dim3(tile_grid), dim3(kBlock), dv.pitch * 8, s, dv, d_nodes.p,
                                   d_tiles.p, n_tiles, cur, N, chunk.d, nstride, hdr_off, masks.p, tile_left.p);
            } else {
                const size_t sh = (size_t)dv.pitch * 4;
#define AH_LAUNCH(M)                                                                                            \
    hipLaunchKernelGGL((k_forest_margin_f32<M>), dim3(tile_grid), dim3(kBlock), sh, s, dv, d_nodes.p, d_tiles.p, \
                       n_tiles, cur, N, chunk.d, nstride, hdr_off, masks.p, tile_left.p)
                switch (ds->metric) {
                case AH_EUCLIDEAN: AH_LAUNCH(AH_EUCLIDEAN); break;
                case AH_MANHATTAN: AH_LAUNCH(AH_MANHATTAN); break;
                case AH_COSINE: AH_LAUNCH(AH_COSINE); break;
                default: AH_LAUNCH(AH_DOT_PRODUCT); break;
                }
#undef AH_LAUNCH
            }
            AH_HIP(hipEventRecord(ep.b, s));
            AH_DBG(s, "margin");
            hipLaunchKernelGGL(k_forest_decide, dim3((n_nodes + 255) / 256), dim3(256), 0, s, d_nodes.p, n_nodes);
            AH_DBG(s, "decide");
        }
        hipLaunchKernelGGL(k_forest_random_sides, dim3(std::min<uint32_t>(n_tiles, kMaxBlocks)), dim3(64), 0, s,
                           d_nodes.p, d_tiles.p, n_tiles, masks.p, tile_left.p);
        AH_DBG(s, "random_sides");
        hipLaunchKernelGGL(k_forest_tile_offsets, dim3(std::min<uint32_t>(n_nodes, kMaxBlocks)), dim3(64), 0, s,
                           d_nodes.p, n_nodes, tile_left.p, tile_left_off.p);
        AH_DBG(s, "tile_offsets");
        hipLaunchKernelGGL(k_forest_scatter, dim3(tile_grid), dim3(kBlock), 0, s, d_nodes.p, d_tiles.p, n_tiles, cur, nxt,
                           final_perm.p, N, masks.p, tile_left_off.p, split_after);
        AH_DBG(s, "scatter");
        AH_HIP(hipGetLastError());
        AH_HIP(hipMemcpyAsync(h_nodes, d_nodes.p, n_nodes * sizeof(FNode), hipMemcpyDeviceToHost, s));
        AH_HIP(hipStreamSynchronize(s));
        // this level's normals are final: the worker copies them while the next level runs
        prefault_normals.join();  // never touch a page the worker may already have filled
        rb.push(forest->normals + chunk.host_off, chunk.d, chunk.bytes);

        // host: materialise the split records and the next level (children lists subdivide the parent range)
        level.clear();
        prev_rows = row_tc >= 2;
        fix_nodes.clear();
        if (prev_rows) h_child.assign(2 * (size_t)n_nodes, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < n_nodes; i++) {
            const FNode nd = h_nodes[i];
            AH_REQUIRE(nd.state != ST_PENDING && nd.n_left <= nd.count, AH_ERR_DEVICE,
                       "forest build: node %u left pending (internal error)", i);
            forest->stats.margin_evaluations += (uint64_t)(nd.attempt + 1) * nd.count;
            forest->stats.retries += nd.attempt;
            items_routed += nd.count;
            const uint32_t rec_idx = nd.rec;
            recs[rec_idx].has_normal = nd.state == ST_ACCEPTED;
            recs[rec_idx].normal_off = chunk.host_off + (uint64_t)i * nstride;
            if (nd.state != ST_ACCEPTED) forest->stats.dummy_normals++;
            const uint32_t child_cnt[2] = {nd.n_left, nd.count - nd.n_left};
            const uint64_t child_start[2] = {nd.start, nd.start + nd.n_left};
            for (uint32_t side = 0; side < 2; side++) {
                HostRec c{};
                c.tree = nd.tree;
                c.start = child_start[side];
                c.count = child_cnt[side];
                c.depth = depth + 1;
                const uint32_t cidx = (uint32_t)recs.size();
                if (c.count <= split_after) {
                    c.kind = AH_NODE_DESCENDANTS;
                } else {
                    c.kind = AH_NODE_SPLIT;
                    FNode cn{};
                    cn.key = ah_node_key_child(nd.key, side);
                    cn.start = c.start;
                    cn.tree = nd.tree;
                    cn.count = c.count;
                    cn.rec = cidx;
                    if (prev_rows) {
                        // sides of a first-attempt accept are the ones the row-major pass left in side_bytes
                        if (nd.state == ST_ACCEPTED && nd.attempt == 0) h_child[2 * (size_t)i + side] = (uint32_t)level.size();
                        else fix_nodes.push_back((uint32_t)level.size());
                    }
                    level.push_back(cn);
                }
                recs.push_back(c);
                if (side == 0) recs[rec_idx].left = cidx;
                else recs[rec_idx].right = cidx;
            }
        }
        std::swap(cur, nxt);
        depth++;
        forest->stats.levels = std::max(forest->stats.levels, depth);
        if (opt->progress) opt->progress(opt->progress_user, depth, recs.size(), items_routed);
    }
    AH_HIP(hipEventRecord(bc.ev_end, s));
    const auto t_levels = std::chrono::steady_clock::now();

    // Results come back with plain D2H copies straight into their final place — no host-side repacking:
    //   normals      one copy per level chunk (device record layout == caller-visible layout), already under way
    //   descendants  the final permutations themselves (rows -> item ids on device first)
    {
        prefault.join();
        forest->descendants_len = desc_base + M;
        // trees that are a single Descendants node never went through a scatter: their list is the input itself
        for (uint32_t t = 0; t < n_trees; t++)
            if (recs[tree_root[t]].kind == AH_NODE_DESCENDANTS && tree_base[t + 1] > tree_base[t])
                AH_HIP(hipMemcpyAsync(final_perm.p + tree_base[t], perm_a.p + tree_base[t],
                                      (tree_base[t + 1] - tree_base[t]) * 4, hipMemcpyDeviceToDevice, s));
        if (!ds->identity_ids && M)
            hipLaunchKernelGGL(k_rows_to_ids, dim3(2048), dim3(256), 0, s, final_perm.p, M, ds->d_ids);
        AH_HIP(hipStreamSynchronize(s));
        rb.push(forest->descendants + desc_base, final_perm.p, M * 4);  // lands while the host emits the node list
    }
    float ms = 0.0f;
    AH_HIP(hipEventElapsedTime(&ms, bc.ev_begin, bc.ev_end));
    forest->stats.seconds_device += ms * 1e-3;
    for (EventPair &ep : bc.events) {
        float m = 0.0f;
        if (hipEventElapsedTime(&m, ep.a, ep.b) == hipSuccess) forest->stats.seconds_margin += m * 1e-3;
    }
    forest->stats.margin_launches += bc.events.size();

    // Emit per tree in post-order (children before parents: the order TmpNodes::put receives them,
    // src/writer.rs:1235-1258), with forest-local indices.
    std::vector<uint32_t> new_index(recs.size(), 0xFFFFFFFFu);
    std::vector<std::pair<uint32_t, int>> stack;
    forest->nodes.reserve(forest->nodes.size() + recs.size());
    for (uint32_t t = 0; t < n_trees; t++) {
        stack.clear();
        stack.push_back({tree_root[t], 0});
        while (!stack.empty()) {
            const uint32_t ri = stack.back().first;
            const HostRec &r = recs[ri];
            if (r.kind == AH_NODE_SPLIT && stack.back().second == 0) {
                stack.back().second = 1;
                const uint32_t l = r.left, rr = r.right;
                stack.push_back({rr, 0});
                stack.push_back({l, 0});
                continue;
            }
            ah_node nd{};
            nd.kind = r.kind;
            nd.has_normal = r.has_normal;
            nd.tree = (uint16_t)(first_tree + t);
            nd.count = r.count;
            nd.depth = r.depth;
            if (r.kind == AH_NODE_SPLIT) {
                nd.left = new_index[r.left];
                nd.right = new_index[r.right];
                nd.offset = r.normal_off;
                forest->stats.split_nodes++;
            } else {
                nd.offset = desc_base + r.start;
                forest->stats.descendant_nodes++;
            }
            new_index[ri] = (uint32_t)forest->nodes.size();
            forest->nodes.push_back(nd);
            stack.pop_back();
        }
        forest->roots.push_back(new_index[tree_root[t]]);
    }
    const auto t_emitted = std::chrono::steady_clock::now();
    AH_REQUIRE(rb.drain() == hipSuccess, AH_ERR_DEVICE, "device -> host copy of the forest failed");
    forest->normals_len = normals_base + normals_bytes;
    if (normals_cap > forest->normals_len) {  // give the head-room back (shrinks in place)
        uint8_t *fit = (uint8_t *)realloc(forest->normals, forest->normals_len + 16);
        if (fit) forest->normals = fit;
    }
    if (getenv("AH_TIMING")) {
        const auto t_end = std::chrono::steady_clock::now();
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        fprintf(stderr, "[ah] batch of %u trees: levels %.3f s (device %.3f), emit %.3f s, read-back still in flight after it "
                        "%.3f s (%.2f GB normals, %.2f GB ids)\n",
                n_trees, sec(t_batch, t_levels), ms * 1e-3, sec(t_levels, t_emitted), sec(t_emitted, t_end),
                normals_bytes / 1e9, M * 4 / 1e9);
    }
    return AH_OK;
}

extern "C" {

static int build_forest_impl(ah_dataset *ds, const ah_build_options *options, const uint32_t *subset_ids,
                             const uint64_t *subset_offsets, ah_forest **out) {
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(ds && options, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE(ds->finalized, AH_ERR_NOT_FINALIZED, "dataset not finalized (call ah_dataset_finalize)");
    AH_REQUIRE(options->n_trees == 0 || options->tree_seeds, AH_ERR_INVALID_ARGUMENT, "tree_seeds is NULL");
    AH_REQUIRE(options->n_trees <= 0xFFFF || subset_ids, AH_ERR_INVALID_ARGUMENT, "at most 65535 trees per call");
    AH_REQUIRE(ds->metric != AH_DOT_PRODUCT || ds->dot_preprocessed, AH_ERR_NEED_PREPROCESS,
               "DotProduct needs ah_preprocess_dot before the build (src/writer.rs:964-976)");
    AH_HIP(hipSetDevice(ds->device));
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t split_after = options->split_after ? options->split_after : ds->dims;  // src/writer.rs:474-477
    ah_forest *forest = new (std::nothrow) ah_forest();
    AH_REQUIRE(forest, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
    forest->normal_vector_offset = 0;
    forest->normal_header_offset = ds->row_bytes();
    forest->normal_stride = ds->row_bytes() + 16;
    int st = AH_OK;
    if (!subset_ids && ds->n <= split_after) {
        // fit_in_descendant at the root (src/writer.rs:1183-1188): every tree is one Descendants node
        const uint64_t total = ds->n * options->n_trees;
        forest->descendants = (uint32_t *)malloc(total * 4 + 16);
        if (!forest->descendants) {
            delete forest;
            set_error("host allocation failed");
            return AH_ERR_OUT_OF_MEMORY;
        }
        forest->descendants_len = total;
        for (uint32_t t = 0; t < options->n_trees; t++) {
            for (uint64_t i = 0; i < ds->n; i++)
                forest->descendants[t * ds->n + i] = ds->identity_ids ? (uint32_t)i : ds->h_ids[i];
            ah_node nd{};
            nd.kind = AH_NODE_DESCENDANTS;
            nd.tree = (uint16_t)t;
            nd.count = (uint32_t)ds->n;
            nd.offset = t * ds->n;
            forest->roots.push_back((uint32_t)forest->nodes.size());
            forest->nodes.push_back(nd);
            forest->stats.descendant_nodes++;
        }
    } else if (options->n_trees) {
        ContextLease lease(ds);
        if (!lease.c) {
            set_error("cannot create a HIP stream");
            st = AH_ERR_DEVICE;
        } else {
            // Trees in flight: bounded by HBM (per item and tree: 3 permutations + node index + side byte + masks = 18
            // bytes, plus the normals of all levels) or by the caller.
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            uint32_t batch = options->n_trees;
            if (!subset_ids) {
                const uint64_t per_tree =
                    ds->n * 18 + ((ds->n / ((uint64_t)split_after + 1)) + 2) * 2 * (ds->row_bytes() + 128) + (1u << 20);
                uint64_t fit = (uint64_t)(free_b * 0.8) / per_tree;
                if (fit < 1) fit = 1;
                batch = (uint32_t)std::min<uint64_t>(fit, options->n_trees);
            }
            if (options->max_trees_in_flight) batch = std::min(batch, options->max_trees_in_flight);
            try {
                for (uint32_t first = 0; first < options->n_trees && st == AH_OK; first += batch)
                    st = build_batch(ds, options, first, std::min(batch, options->n_trees - first), split_after, forest,
                                     lease.c, subset_ids, subset_offsets);
            } catch (const std::bad_alloc &) {
                set_error("host allocation failed during the forest build");
                st = AH_ERR_OUT_OF_MEMORY;
            }
        }
    }
    if (st != AH_OK) {
        delete forest;
        return st;
    }
    forest->stats.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = forest;
    return AH_OK;
}

int ah_build_forest(ah_dataset *ds, const ah_build_options *options, ah_forest **out) {
    return build_forest_impl(ds, options, nullptr, nullptr, out);
}

// `incremental_index_large_descendant` (src/writer.rs:660-739) for many descendants at once: tree t of the result is
// `make_tree_in_file` over the ascending id list item_ids[offsets[t] .. offsets[t+1]).
int ah_build_subtrees(ah_dataset *ds, const ah_build_options *options, const uint32_t *item_ids, const uint64_t *offsets,
                      ah_forest **out) {
    AH_REQUIRE(options, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE((item_ids && offsets) || options->n_trees == 0, AH_ERR_INVALID_ARGUMENT, "NULL id lists");
    for (uint32_t t = 0; t < options->n_trees; t++) {
        AH_REQUIRE(offsets[t + 1] >= offsets[t], AH_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
        for (uint64_t i = offsets[t] + 1; i < offsets[t + 1]; i++)
            AH_REQUIRE(item_ids[i] > item_ids[i - 1], AH_ERR_INVALID_ARGUMENT,
                       "sub-tree id lists must be strictly ascending (RoaringBitmap order)");
    }
    static const uint32_t dummy = 0;
    return build_forest_impl(ds, options, item_ids ? item_ids : &dummy, offsets, out);
}

int ah_forest_view_get(const ah_forest *forest, ah_forest_view *out) {
    AH_REQUIRE(forest && out, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    out->n_trees = (uint32_t)forest->roots.size();
    out->n_nodes = forest->nodes.size();
    out->roots = forest->roots.data();
    out->nodes = forest->nodes.data();
    out->normals = forest->normals;
    out->normals_len = forest->normals_len;
    out->normal_stride = forest->normal_stride;
    out->normal_vector_offset = forest->normal_vector_offset;
    out->normal_header_offset = forest->normal_header_offset;
    out->descendants = forest->descendants;
    out->descendants_len = forest->descendants_len;
    return AH_OK;
}

int ah_forest_stats(const ah_forest *forest, ah_build_stats *out) {
    AH_REQUIRE(forest && out, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = forest->stats;
    return AH_OK;
}

// payload: SPLIT -> the normal record (vector at normal_vector_offset, header at normal_header_offset) or
// NULL for `normal: None`; DESCENDANTS -> u32 item ids.
int ah_forest_visit(const ah_forest *forest, ah_node_sink_fn sink, void *user) {
    AH_REQUIRE(forest && sink, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    for (size_t i = 0; i < forest->nodes.size(); i++) {
        const ah_node &nd = forest->nodes[i];
        const void *payload = nullptr;
        size_t len = 0;
        if (nd.kind == AH_NODE_SPLIT) {
            if (nd.has_normal) {
                payload = forest->normals + nd.offset;
                len = forest->normal_stride;
            }
        } else {
            payload = forest->descendants + nd.offset;
            len = (size_t)nd.count * 4;
        }
        const int rc = sink(user, nd.tree, (uint32_t)i, nd.kind, nd.left, nd.right, payload, len);
        AH_REQUIRE(rc == 0, AH_ERR_CANCELLED, "node sink asked to stop (code %d)", rc);
    }
    return AH_OK;
}

int ah_forest_destroy(ah_forest *forest) {
    delete forest;
    return AH_OK;
}

}  // extern "C"
