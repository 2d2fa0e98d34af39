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
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <new>

#include "common.h"
#include "split_device.h"
#include "screen_device.h"

namespace ah {

static constexpr uint32_t kTile = 2048;  // items per tile = 32 octets x 64 items
static constexpr int kBlock = 256;
static constexpr int kMaxBlocks = 4096;

enum { ST_PENDING = 0, ST_ACCEPTED = 1, ST_RANDOM = 2 };

// The build's abort word (device memory, written by a DMA on the side stream when the caller's cancel flag turns
// non-zero).  Every block of the margin kernels reads it first and drains if it is set: launches that start after the
// write has landed cost nothing.  The read is an ordinary cached load on purpose — an uncached (system-scope) read of
// one word by ~100 000 blocks per launch serialises on that address and was measured to cost 70 % of the build
// (0.088 s -> 0.150 s at 1M x 768), so a kernel that is already running may keep seeing a stale line and finish.
// `gate` (the retry attempts of a level, nullptr otherwise): the number of nodes the attempt before left pending.  On most
// data a level's first attempt settles every node, and the three attempts that follow — five launches each, over every node
// and tile of the level — found nothing to do at 0.9-2.3 ms per level of the 10M x 100-tree build: with the count at zero
// their blocks leave at once.
struct AbortFlags {
    const uint32_t *dev;
    const uint32_t *gate;
};
// (read once, when a block starts: a second dependent load in front of every tile of the node-major kernels' loops cost the
// retries of a build that HAS retries 75 ms — 10M clustered rows, 741 k re-drawn splits)
__device__ __forceinline__ bool abort_requested(const AbortFlags f) { return __builtin_nontemporal_load(f.dev) != 0u; }
__device__ __forceinline__ bool gate_closed(const uint32_t *gate) { return gate != nullptr && __builtin_nontemporal_load(gate) == 0u; }

struct FNode {
    uint64_t key;      // ah_node_key_* of this node
    uint64_t start;    // first position inside the batch permutation (absolute: tree base + offset)
    uint32_t tree;     // tree index inside the batch
    uint32_t count;    // items under the node
    uint32_t n_left;   // accumulated by the margin kernel (integer atomics)
    uint32_t attempt;  // current / final split attempt (0..3)
    uint32_t state;    // ST_*
    uint32_t tile_begin, n_tiles;
    uint32_t fix;      // the parent's sides were re-drawn after its row-major pass (retry / random fallback): the rows of
                       // this node cannot take their node index from k_forest_advance_node_of
};
// What the host needs to know about a level before it can launch it (written by the k_next_* kernels, read back
// through pinned memory): sizes, the cost model's inputs, and the first node of every tree (nodes are ordered by tree).
struct LevelInfo {
    uint32_t n_nodes, n_tiles, n_fix, pad;
    unsigned long long pairs;  // items under the level's nodes = margin evaluations of a first attempt
    unsigned long long pad2;
    // followed by uint32_t tree_first[n_trees + 1]
};
struct ScreenCounters {
    unsigned long long fallbacks, violations, stage8_pairs, stage8_decided, stage8b_decided, pad;
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
// M = f32_space_metric(dv.metric): one instantiation per arithmetic (the switch inside one kernel made every build carry the
// registers of the heaviest branch: 115 VGPRs and 80 spilled SGPRs for a kernel that lives on the number of waves in flight)
template <int M>
__global__ __launch_bounds__(64) void k_forest_create_split(DataView dv, FNode *nodes, uint32_t n_nodes,
                                                            const uint32_t *__restrict__ perm, uint64_t n_items,
                                                            uint8_t *normals, uint64_t nstride, uint64_t hdr_off,
                                                            const uint32_t *gate) {
    extern __shared__ float4 s_buf4[];
    float *s_buf = reinterpret_cast<float *>(s_buf4);
    __shared__ uint32_t s_rows[AH_SPLIT_SAMPLES];
    if (gate_closed(gate)) return;
    const uint32_t fpitch = f32_space_pitch(dv.metric, dv.dims);
    for (uint32_t node = blockIdx.x; node < n_nodes; node += gridDim.x) {  // grid-stride: see node_grid in build_batch
        FNode &nd = nodes[node];
        if (nd.state != ST_PENDING) continue;
        __syncthreads();  // the previous node's samples / two-means buffers are dead
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
        uint8_t *rec = normals + node * nstride;
        float *hdr = reinterpret_cast<float *>(rec + hdr_off);
        wave_create_split<M>(dv, s_rows, s_buf, s_buf + fpitch, s_buf + 2 * fpitch, rec, hdr, threadIdx.x);
        if (threadIdx.x == 0) hdr[2] = hdr[3] = 0.0f;  // deterministic padding
        for (uint32_t i = 4 + threadIdx.x; i < (uint32_t)((nstride - hdr_off) >> 2); i += 64) hdr[i] = 0.0f;
    }
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
                                                              uint32_t *__restrict__ tile_left,
                                                              const AbortFlags abort_flag) {
    extern __shared__ float4 s_n4[];
    __shared__ uint32_t s_left;
    const float *s_n = reinterpret_cast<const float *>(s_n4);
    const uint32_t o = threadIdx.x >> 3, j = threadIdx.x & 7u;
    if (gate_closed(abort_flag.gate)) return;  // a retry attempt with no node left pending
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (abort_requested(abort_flag)) return;  // build cancelled: drain (block-uniform)
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
                                                             uint32_t *__restrict__ tile_left,
                                                             const AbortFlags abort_flag) {
    extern __shared__ uint64_t s_nw[];
    __shared__ uint32_t s_left;
    __shared__ uint8_t s_side[kTile];
    if (gate_closed(abort_flag.gate)) return;  // a retry attempt with no node left pending
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (abort_requested(abort_flag)) return;
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
                                                                  uint32_t *__restrict__ node_of, uint32_t only_fix) {
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (only_fix && !nd->fix) continue;
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
                                                               uint64_t hdr_off, uint8_t *__restrict__ side_bytes,
                                                               const AbortFlags abort_flag) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.dims >> 5;
    if (abort_requested(abort_flag)) return;
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
                                                                 uint32_t first_node, uint32_t n_group_nodes,
                                                                 const AbortFlags abort_flag) {
    extern __shared__ float4 s_norm4[];
    if (abort_requested(abort_flag)) return;
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

// The side bytes of a level (0 / 1 per (tree, row)) as one bit each, bit g of the flat array = pair g = tree * N + row.
// k_forest_masks_from_bytes reads the sides in the order of the trees' permutations, i.e. at random: a tree's bits are
// N / 8 bytes (1.25 MB at 10M rows: they stay in an L2) where its bytes are N (10 MB: most gathers go out to the fabric).
__global__ __launch_bounds__(256) void k_forest_pack_sides(const uint8_t *__restrict__ side_bytes, uint64_t n_words,
                                                           uint32_t *__restrict__ bits) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint4 a = *reinterpret_cast<const uint4 *>(side_bytes + w * 32);
        const uint4 b = *reinterpret_cast<const uint4 *>(side_bytes + w * 32 + 16);
        // bits 0 / 8 / 16 / 24 of a word -> bits 24..27 of the product (no two partial products meet, nothing carries)
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t out = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) out |= ((((v[e] & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * e);
        bits[w] = out;
    }
}

// side bytes (by row) -> the per-tile masks / left counts of the node-major pipeline.  `side_bits` != nullptr: the same
// sides, packed by k_forest_pack_sides.
__global__ __launch_bounds__(kBlock) void k_forest_masks_from_bytes(FNode *nodes, const FTile *__restrict__ tiles,
                                                                    uint32_t n_tiles, const uint32_t *__restrict__ perm,
                                                                    uint64_t n_items,
                                                                    const uint8_t *__restrict__ side_bytes,
                                                                    const uint32_t *__restrict__ side_bits,
                                                                    uint64_t *__restrict__ masks,
                                                                    uint32_t *__restrict__ tile_left) {
    // Item p of a tile is bit (p >> 5) of mask (p & 31).  Thread (wave w, lane l) takes the items 256 k + 64 w + l, k = 0..7:
    // the ballot of its k-th side over the wave holds, for every mask o, the two bits 8 k + 2 w (lane o) and 8 k + 2 w + 1
    // (lane o + 32).  The 32 ballots of a tile go through LDS; thread o < 32 assembles mask o from them.
    __shared__ uint64_t s_ballot[8][kBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_PENDING) continue;
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        const uint8_t *sb = side_bytes + (uint64_t)nd->tree * n_items;
        uint32_t rows[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t p = 256u * k + threadIdx.x;
            rows[k] = p < in_tile ? pp[p] : 0xFFFFFFFFu;
        }
        uint32_t side[8];
        if (side_bits) {
            const uint64_t g0 = (uint64_t)nd->tree * n_items;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint64_t g = g0 + rows[k];
                side[k] = rows[k] != 0xFFFFFFFFu ? side_bits[g >> 5] >> (uint32_t)(g & 31u) : 0u;
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) side[k] = rows[k] != 0xFFFFFFFFu ? (uint32_t)sb[rows[k]] : 0u;
        }
        __syncthreads();  // the previous tile's ballots have been consumed
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const unsigned long long bal = __ballot(side[k] & 1u);
            if (lane == 0) s_ballot[k][wave] = bal;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const uint32_t o = threadIdx.x;
            uint64_t mask = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
#pragma unroll
                for (uint32_t w = 0; w < kBlock / 64; w++) {
                    const uint64_t bal = s_ballot[k][w];
                    mask |= ((bal >> o) & 1ull) << (8 * k + 2 * w);
                    mask |= ((bal >> (o + 32)) & 1ull) << (8 * k + 2 * w + 1);
                }
            const uint32_t n_mine = in_tile > o ? (in_tile - o + 31u) / 32u : 0u;  // items o, o + 32, ... of the tile
            uint32_t lefts = n_mine - (uint32_t)__popcll((unsigned long long)mask);
            masks[(uint64_t)tile * 32 + o] = mask;
            for (int off = 16; off > 0; off >>= 1) lefts += __shfl_down(lefts, off, 32);
            if (o == 0) {
                tile_left[tile] = lefts;
                if (lefts) atomicAdd(&nodes[tl.node].n_left, lefts);
            }
        }
    }
}

// split_imbalance (src/writer.rs:1348-1353, f64) and the accept / retry / random decision (:1209-1227).
__global__ void k_forest_decide(FNode *nodes, uint32_t n_nodes, uint32_t *pend_out, const uint32_t *gate) {
    if (gate_closed(gate)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool again = false;
    if (i < n_nodes && nodes[i].state == ST_PENDING) {
        FNode &nd = nodes[i];
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
            again = true;
        }
    }
    // nodes left pending, counted once per wave (the gate of the next attempt's launches)
    const unsigned long long m = __ballot(again);
    if (pend_out && m && (threadIdx.x & 63u) == 0) atomicAdd(pend_out, (uint32_t)__popcll(m));
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
    __shared__ uint32_t s_in[kTile], s_out[kTile];  // the tile's items as read / partitioned (lefts first)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const FTile tl = tiles[tile];
        const FNode nd = nodes[tl.node];
        const uint32_t in_tile = min(kTile, nd.count - tl.first);
        const uint64_t base = nd.start;
        __syncthreads();
        if (threadIdx.x < 32) s_masks[threadIdx.x] = masks[(uint64_t)tile * 32 + threadIdx.x];
        {  // coalesced read of the tile
            const uint32_t *src = perm_cur + base + tl.first;
            for (uint32_t p = threadIdx.x; p < in_tile; p += kBlock) s_in[p] = src[p];
        }
        __syncthreads();
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
        uint32_t wave_base = 0, tile_lefts = 0;
        for (uint32_t w = 0; w < kBlock / 64; w++) {
            if (w < wave) wave_base += s_wave[w];
            tile_lefts += s_wave[w];
        }
        // stable partition inside LDS: lefts to [0, tile_lefts), rights to [tile_lefts, in_tile), order kept
        uint32_t lp = wave_base + incl - my_left;  // lefts of the tile before p0
        uint32_t rp = tile_lefts + (p0 - lp);      // rights of the tile before p0, behind all lefts
        for (uint32_t e = 0; e < nvalid; e++) {
            const uint32_t row = s_in[p0 + e];
            if ((sidebits >> e) & 1u) s_out[rp++] = row;
            else s_out[lp++] = row;
        }
        __syncthreads();
        // two contiguous runs out, coalesced: the node's lefts before this tile = tile_left_off, its rights before it =
        // items before the tile - lefts before the tile
        const uint32_t lefts_before = tile_left_off[tile];
        const uint32_t n_right = nd.count - nd.n_left;
        uint32_t *dst_l = (nd.n_left <= split_after ? final_perm : perm_next) + base + lefts_before;
        uint32_t *dst_r = (n_right <= split_after ? final_perm : perm_next) + base + nd.n_left + (tl.first - lefts_before);
        for (uint32_t p = threadIdx.x; p < in_tile; p += kBlock) {
            const uint32_t row = s_out[p];
            if (p < tile_lefts) dst_l[p] = row;
            else dst_r[p - tile_lefts] = row;
        }
    }
}

__global__ void k_rows_to_ids(uint32_t *perm, uint64_t total, const uint32_t *__restrict__ ids) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) perm[g] = ids[perm[g]];
}

// ------------------------------------------------------------------------------------------------
// Certified binary16 screen (screen_device.h): shadow copies and the screened variants of the three margin kernels.
// ------------------------------------------------------------------------------------------------

// (to_shadow_half and kTinyBits: screen_device.h — the search's certified top-k screen makes the same copies of its queries)

// rows -> binary16 shadow + per-row stats.  One octet per row, lane j converts elements 32k + 4j .. +3 (8 bytes out).
__global__ __launch_bounds__(kBlock) void k_shadow_rows(DataView dv, uint16_t *__restrict__ h_rows, uint32_t hpitch,
                                                        float4 *__restrict__ stats) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.pitch >> 5;  // the padding of the f32 row is zero
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < dv.n; row += n_octets) {
        const float4 *r4 = reinterpret_cast<const float4 *>(dv.rows_f32 + row * dv.pitch) + j;
        uint2 *o2 = reinterpret_cast<uint2 *>(h_rows + row * hpitch) + j;
        float sa = 0.f, sb = 0.f, sc = 0.f;
        uint32_t xbits = 0u;
        for (uint32_t k = 0; k < blocks; k++) {
            float4 x = ld_stream(r4 + k * 8);
            const uint32_t e0 = 32 * k + 4 * j;  // elements beyond dims (row padding) count as zeros
            if (e0 + 0 >= dv.dims) x.x = 0.0f;
            if (e0 + 1 >= dv.dims) x.y = 0.0f;
            if (e0 + 2 >= dv.dims) x.z = 0.0f;
            if (e0 + 3 >= dv.dims) x.w = 0.0f;
            xbits = max(max(xbits, __float_as_uint(x.x) & 0x7FFFFFFFu), max(__float_as_uint(x.y) & 0x7FFFFFFFu,
                        max(__float_as_uint(x.z) & 0x7FFFFFFFu, __float_as_uint(x.w) & 0x7FFFFFFFu)));
            const _Float16 h0 = to_shadow_half(x.x), h1 = to_shadow_half(x.y), h2 = to_shadow_half(x.z), h3 = to_shadow_half(x.w);
            const float y0 = (float)h0, y1 = (float)h1, y2 = (float)h2, y3 = (float)h3;
            sa += y0 * y0 + y1 * y1 + y2 * y2 + y3 * y3;
            const float d0 = x.x - y0, d1 = x.y - y1, d2 = x.z - y2, d3 = x.w - y3;
            sb += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            sc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            f16x2_t lo = {h0, h1}, hi = {h2, h3};
            o2[k * 8] = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
        }
        if (hpitch > dv.pitch) o2[blocks * 8] = make_uint2(0u, 0u);  // pitch is 32 mod 64: zero the last half line
        sa = octet_sum(sa);
        sb = octet_sum(sb);
        sc = octet_sum(sc);
        // 2-norms rounded UP: the f32 sums of squares carry a relative error below (dims + 8) * 2^-24
        const float up = 1.0f + (float)(dv.pitch + 64u) * 1.2e-7f;
        xbits = max(xbits, (uint32_t)__shfl_xor((int)xbits, 1, 8));
        xbits = max(xbits, (uint32_t)__shfl_xor((int)xbits, 2, 8));
        xbits = max(xbits, (uint32_t)__shfl_xor((int)xbits, 4, 8));
        const bool tiny = xbits != 0u && xbits < kTinyBits;
        const float inf = __uint_as_float(0x7F800000u);
        if (j == 0) stats[row] = tiny ? make_float4(inf, inf, inf, 0.0f) : make_float4(sqrtf(sa) * up, sqrtf(sb) * up, sqrtf(sc) * up, 0.0f);
    }
}

// Component-wise maximum of the per-row stats (all >= 0, so the bit patterns order like the values; inf / NaN end up on
// top and simply disable the cheap test below).  The node-major screen tries the bound built from these maxima first — it
// needs no per-row load — and fetches the row's own stats (a 16-byte random read = one more 128-byte line per pair) only
// when that bound cannot decide: for rows of similar norms that is ~1.5 % of the pairs.
__global__ __launch_bounds__(256) void k_stats_max(const float4 *__restrict__ stats, uint64_t n, uint32_t *__restrict__ out) {
    __shared__ uint32_t s_m[3];
    __shared__ float s_sum;
    if (threadIdx.x < 3) s_m[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_sum = 0.0f;
    __syncthreads();
    uint32_t a = 0, b = 0, c = 0;
    float sum = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 v = stats[i];
        a = max(a, __float_as_uint(v.x));
        b = max(b, __float_as_uint(v.y));
        c = max(c, __float_as_uint(v.z));
        sum += v.z;
    }
    atomicMax(&s_m[0], a);
    atomicMax(&s_m[1], b);
    atomicMax(&s_m[2], c);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&s_sum, sum);
    __syncthreads();
    if (threadIdx.x < 3) atomicMax(&out[threadIdx.x], s_m[threadIdx.x]);
    // out[3]: sum of the row norms (a float; its summation order is not fixed — it only feeds the heuristic that decides
    // whether the int8 copy is worth keeping, never a result)
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<float *>(out) + 3, s_sum);
}

// ------------------------------------------------------------------------------------------------
// First stage of the node-major screen: an int8 copy.  The deep levels move one row per (item, node) pair and run at the
// device's gather rate (1.6 KB per pair with the binary16 copy); the same certified-sign argument holds for ANY copy whose
// distance to the original is measured, so a coarser copy goes first.
//
//   rows      y = x / d (d = one power of two per DIMENSION: the largest |x_i| of the column rounded up, so a few
//             "outlier dimensions" do not eat the 8 bits of all the others; exact, <n, x> = <n o d, x / d>), then
//             y~ = s_r q with ONE SCALE PER ROW (s_r = max|y| / 127, q = int8): rows of very different norms and data with
//             long tails (N(0,1): the 5-sigma entries of the dataset) quantise as well as uniform data;
//   normals   n' = n o d, n~' = s_n (q_hi + q_lo / 256): TWO int8 digits.  The normal sits in LDS, so its second digit
//             costs no HBM byte, and it removes the normal's half of the error bound (76 % -> ~90 % of the pairs decided
//             on the uniform benchmark rows, ~69 % -> ~84 % on N(0,1) rows);
//   screen    S = s_n (<q_hi, q> + <q_lo, q> / 256), integer dot products (v_dot4_i32_i8), exact;
//   bound     |s_r S - r| <= s_r (|n' - n~'| |q| + |n'| |y/s_r - q| + g8 |n~'||q|) + gamma_r |n||x|.
//
// The sign of a cosine margin does not depend on s_r > 0, so for Cosine the test runs in units of s_r with the
// DATASET-WIDE maxima of |q|, |y/s_r - q|, |x|/s_r (all dimensionless, the same for every row up to a few percent): a decided
// pair costs its 768-byte int8 row and nothing else — no per-row load at all.  Euclidean / Manhattan add the bias in real
// units and fetch s_r (4 bytes per pair from a 40 MB array).  DotProduct has no int8 stage (its margin needs the row's header
// anyway).  Rows that are all zero, not finite, or so small that f32 squares underflow (max|x| < 2^-40) get q = 0 and
// s_r = inf: they are never decided here.  Sides stay identical by construction; AH_SCREEN_VERIFY checks every pair at
// every stage.
// ------------------------------------------------------------------------------------------------
struct NormalStats8 {
    float an, bn, cn, extra;    // |n~'|, |n' - n~'|, |n'| (in the column-scaled space), bias
    float scale, cn0, pad0, pad1;  // s_n; |n| in the original space (for the reference's own rounding error)
};

// column-wise max |x| (as bits) over all rows: thread t of a block owns columns t, t + 256, ...
__global__ __launch_bounds__(256) void k_col_maxabs(DataView dv, uint32_t *__restrict__ out_bits) {
    const uint64_t rows_per_block = (dv.n + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block, r1 = min(dv.n, r0 + rows_per_block);
    for (uint32_t c = threadIdx.x; c < dv.dims; c += blockDim.x) {
        uint32_t m = 0;
        for (uint64_t r = r0; r < r1; r++) m = max(m, __float_as_uint(dv.rows_f32[r * dv.pitch + c]) & 0x7FFFFFFFu);
        atomicMax(&out_bits[c], m);
    }
}
// column maxima -> one power of two per dimension (d >= max|x_i|; 1 for empty / non-finite columns), and its inverse
__global__ void k_dim_scales(const uint32_t *__restrict__ max_bits, uint32_t dims, uint32_t pitch8, float *__restrict__ d,
                             float *__restrict__ inv_d) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pitch8) return;
    float v = 1.0f;
    if (i < dims) {
        const uint32_t b = max_bits[i];
        uint32_t e = b >> 23;                      // biased exponent of the column maximum
        if ((b & 0x7FFFFFu) != 0u) e += 1;         // not a power of two itself: round up
        if (b != 0u && b < 0x7F800000u && e >= 64u && e <= 190u) v = __uint_as_float(e << 23);  // 2^-63 .. 2^63, else 1
    }
    d[i] = v;
    inv_d[i] = 1.0f / v;  // exact: a power of two
}
__device__ __forceinline__ uint32_t octet_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__shfl_xor((int)v, 1, 8));
    v = max(v, (uint32_t)__shfl_xor((int)v, 2, 8));
    return max(v, (uint32_t)__shfl_xor((int)v, 4, 8));
}
// rows -> int8 copy with one scale per row; max_bits[0..2] = max over the rows of |q|, |y / s_r - q|, |x| / s_r (rounded up).
// One octet per row: a pass for the row's max |y|, a pass that quantises (the row comes back from L2).
// rows8_lo: the rows' SECOND int8 digit, q2 = round((y / s_r - q) 256) (stage 1 of the node-major screen: a pair the first
// digit cannot decide reads 768 more bytes instead of the 1536-byte binary16 row); max_bits[3..4] = max |q + q2/256| and
// max |y / s_r - q - q2/256|.
__global__ __launch_bounds__(kBlock) void k_shadow_rows8(DataView dv, const float *__restrict__ inv_d, int8_t *__restrict__ rows8,
                                                         int8_t *__restrict__ rows8_lo, uint32_t pitch8,
                                                         float *__restrict__ row_scale, uint32_t *__restrict__ max_bits) {
    __shared__ uint32_t s_m[5];
    if (threadIdx.x < 5) s_m[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    const uint32_t blocks = dv.pitch >> 5;
    const float4 *id4 = reinterpret_cast<const float4 *>(inv_d) + j;
    float ma = 0.f, mb = 0.f, mc = 0.f, ma2 = 0.f, mb2 = 0.f;
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < dv.n; row += n_octets) {
        const float4 *r4 = reinterpret_cast<const float4 *>(dv.rows_f32 + row * dv.pitch) + j;
        uint32_t *o = reinterpret_cast<uint32_t *>(rows8 + row * pitch8);
        uint32_t *o2 = reinterpret_cast<uint32_t *>(rows8_lo + row * pitch8);
        uint32_t mbits = 0u, xbits = 0u;  // max |y|, max |x| (non-finite values end up on top)
        for (uint32_t k = 0; k < blocks; k++) {
            const float4 x = r4[k * 8];  // (the padding of a row is zero)
            const float4 g = id4[k * 8];
            mbits = max(max(mbits, __float_as_uint(x.x * g.x) & 0x7FFFFFFFu), max(__float_as_uint(x.y * g.y) & 0x7FFFFFFFu,
                        max(__float_as_uint(x.z * g.z) & 0x7FFFFFFFu, __float_as_uint(x.w * g.w) & 0x7FFFFFFFu)));
            xbits = max(max(xbits, __float_as_uint(x.x) & 0x7FFFFFFFu), max(__float_as_uint(x.y) & 0x7FFFFFFFu,
                        max(__float_as_uint(x.z) & 0x7FFFFFFFu, __float_as_uint(x.w) & 0x7FFFFFFFu)));
        }
        mbits = octet_max_u32(mbits);
        xbits = octet_max_u32(xbits);
        // usable: finite, and neither the row nor its column-scaled image is so small that squares underflow
        const bool ok = mbits >= kTinyBits && xbits >= kTinyBits && mbits < 0x7F800000u && xbits < 0x7F800000u;
        const float m = __uint_as_float(mbits);
        const float scale = ok ? m / 127.0f : 0.0f, inv_scale = ok ? 127.0f / m : 0.0f;
        float sa = 0.f, sb = 0.f, sc = 0.f, sa2 = 0.f, sb2 = 0.f;
        for (uint32_t k = 0; k < blocks; k++) {
            float4 x = ld_stream(r4 + k * 8);
            const float4 g = id4[k * 8];
            const uint32_t e0 = 32 * k + 4 * j;
            if (e0 + 0 >= dv.dims) x.x = 0.0f;
            if (e0 + 1 >= dv.dims) x.y = 0.0f;
            if (e0 + 2 >= dv.dims) x.z = 0.0f;
            if (e0 + 3 >= dv.dims) x.w = 0.0f;
            const float y[4] = {x.x * g.x, x.y * g.y, x.z * g.z, x.w * g.w};  // exact: g is a power of two
            uint32_t w1 = 0u, w2 = 0u;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // (!ok rows — all zero, tiny, or not finite — store q = r = 0: inf * 0 = NaN would otherwise clamp to -127 and,
                // with Cosine's scale-free test, let a non-finite row be "decided" by the sign of garbage)
                const float t = y[c] * inv_scale;
                const int q = ok ? quantize8(y[c], inv_scale) : 0;
                const int r = ok ? (int)fminf(fmaxf(rintf((t - (float)q) * 256.0f), -127.0f), 127.0f) : 0;
                const float z = (float)q * scale, z2 = ((float)q + (float)r * 0.00390625f) * scale;  // the digits sum exactly
                sa += (float)(q * q);  // exact integers (< 2^24 per lane up to 8000 dims)
                const float d = y[c] - z, d2 = y[c] - z2, v2 = (float)q + (float)r * 0.00390625f;
                sb += d * d;
                sa2 += v2 * v2;
                sb2 += d2 * d2;
                w1 |= ((uint32_t)q & 0xFFu) << (8 * c);
                w2 |= ((uint32_t)r & 0xFFu) << (8 * c);
            }
            sc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            o[k * 8 + j] = w1;
            if (rows8_lo) o2[k * 8 + j] = w2;
        }
        for (uint32_t w = (dv.pitch >> 2) + j; w < (pitch8 >> 2); w += 8) {  // zero tail of the int8 rows
            o[w] = 0u;
            if (rows8_lo) o2[w] = 0u;
        }
        sa = octet_sum(sa);
        sb = octet_sum(sb);
        sc = octet_sum(sc);
        sa2 = octet_sum(sa2);
        sb2 = octet_sum(sb2);
        // inf: the row never decides here.  An all-zero row is exact at scale 0 (q = 0, every product with it is 0): the top-k
        // screen of the search (search.hip) then needs no fall-back for the zero vectors real corpora hold
        if (j == 0) row_scale[row] = ok ? scale : (xbits == 0u ? 0.0f : __uint_as_float(0x7F800000u));
        if (ok) {
            // in units of the row's scale, rounded UP: f32 sums of squares (relative error < (pitch + 8) 2^-24), the
            // rounding of scale * q inside the difference (2^-24 |z| per element, |z| <= 127 scale), the division
            const float up = (1.0f + (float)(dv.pitch + 64u) * 1.2e-7f) * 1.000001f;
            ma = fmaxf(ma, sqrtf(sa) * up);
            mb = fmaxf(mb, (sqrtf(sb) * up + 127.0f * scale * 6.0e-8f * sqrtf((float)dv.pitch)) / scale * 1.000001f);
            mc = fmaxf(mc, sqrtf(sc) * up / scale * 1.000001f);
            ma2 = fmaxf(ma2, sqrtf(sa2) * up);
            mb2 = fmaxf(mb2, (sqrtf(sb2) * up + 128.0f * scale * 6.0e-8f * sqrtf((float)dv.pitch)) / scale * 1.000001f);
        }
    }
    atomicMax(&s_m[0], __float_as_uint(ma));
    atomicMax(&s_m[1], __float_as_uint(mb));
    atomicMax(&s_m[2], __float_as_uint(mc));
    atomicMax(&s_m[3], __float_as_uint(ma2));
    atomicMax(&s_m[4], __float_as_uint(mb2));
    __syncthreads();
    if (threadIdx.x < 5) atomicMax(&max_bits[threadIdx.x], s_m[threadIdx.x]);
}
// The level's normals -> int8 records [pitch8 bytes q_hi][pitch8 bytes q_lo][NormalStats8], one wave per pending node.
__global__ __launch_bounds__(64) void k_forest_shadow_normals8(DataView dv, const FNode *__restrict__ nodes, uint32_t n_nodes,
                                                               const uint8_t *__restrict__ normals, uint64_t nstride,
                                                               uint64_t hdr_off, const float *__restrict__ dim_scale,
                                                               uint8_t *__restrict__ shadow8, uint64_t stride8,
                                                               uint32_t pitch8, const uint32_t *gate) {
    if (gate_closed(gate)) return;
    for (uint32_t node = blockIdx.x; node < n_nodes; node += gridDim.x) {
        if (nodes[node].state != ST_PENDING) continue;
        const float *nv = reinterpret_cast<const float *>(normals + node * nstride);
        int8_t *out_hi = reinterpret_cast<int8_t *>(shadow8 + node * stride8);
        int8_t *out_lo = out_hi + pitch8;
        uint32_t mbits = 0;
        for (uint32_t i = threadIdx.x; i < dv.dims; i += 64) mbits = max(mbits, __float_as_uint(nv[i] * dim_scale[i]) & 0x7FFFFFFFu);
        for (int off = 32; off > 0; off >>= 1) mbits = max(mbits, (uint32_t)__shfl_xor((int)mbits, off));
        const float m = __uint_as_float(mbits);
        const bool ok = mbits >= kTinyBits && mbits < 0x7F800000u;  // finite, not (nearly) zero
        const float scale = ok ? m / 127.0f : 0.0f, inv_scale = ok ? 127.0f / m : 0.0f;
        float sa = 0.f, sb = 0.f, sc = 0.f, s0 = 0.f;
        for (uint32_t i = threadIdx.x; i < pitch8; i += 64) {
            const float x0 = i < dv.dims ? nv[i] : 0.0f;
            const float x = i < dv.dims ? x0 * dim_scale[i] : 0.0f;  // exact: a power of two
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
        if (threadIdx.x == 0) {
            const float up = 1.0f + (float)(pitch8 + 64u) * 1.2e-7f;
            const float *nh = reinterpret_cast<const float *>(normals + node * nstride + hdr_off);
            NormalStats8 st;
            st.an = sqrtf(sa) * up;
            st.bn = ok ? sqrtf(sb) * up + 128.0f * scale * 6.0e-8f * sqrtf((float)pitch8) : __uint_as_float(0x7F800000u);
            st.cn = sqrtf(sc) * up;
            st.cn0 = sqrtf(s0) * up;
            st.extra = dv.metric == AH_COSINE ? 0.0f : nh[0];
            st.scale = scale;
            st.pad0 = st.pad1 = 0.0f;
            *reinterpret_cast<NormalStats8 *>(shadow8 + node * stride8 + 2 * (uint64_t)pitch8) = st;
        }
    }
}
// integer dot of one int8 row against the two int8 digits of the normal in LDS, octet-cooperative: lane j covers bytes
// 128 k + 16 j .. +15.  hi4 / lo4 = normal digits (LDS) + j, r4 = row (global, streamed) + j.
// Result: <q_hi, q> + <q_lo, q> / 256 as float (the integers are exact; one rounding per lane total and per octet add).
__device__ __forceinline__ float screen8_octet_dot(const uint4 *hi4, const uint4 *lo4, const uint4 *r4, uint32_t steps) {
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
    return octet_sum((float)(h0 + h1) + (float)(l0 + l1) * 0.00390625f);
}
// the same with the row's chunks already in registers (PRE of them; chunk u exists when u < steps)
template <int PRE>
__device__ __forceinline__ float screen8_octet_dot_regs(const uint4 *hi4, const uint4 *lo4, const uint4 (&x)[PRE], uint32_t steps) {
    int h0 = 0, h1 = 0, l0 = 0, l1 = 0;
#pragma unroll
    for (int u = 0; u < PRE; u += 2) {
        if ((uint32_t)u < steps) {
            h0 = dot16_i8(hi4[u * 8], x[u], h0);
            l0 = dot16_i8(lo4[u * 8], x[u], l0);
        }
        if ((uint32_t)(u + 1) < steps) {
            h1 = dot16_i8(hi4[(u + 1) * 8], x[u + 1], h1);
            l1 = dot16_i8(lo4[(u + 1) * 8], x[u + 1], l1);
        }
    }
    return octet_sum((float)(h0 + h1) + (float)(l0 + l1) * 0.00390625f);
}
// the decision of the int8 stage.  S = s_n x (the digits' dot products), i.e. the screen value in units of the row's scale;
// max8 = dataset-wide {|q|, |y / s_r - q|, |x| / s_r}; s_row = the row's scale (Euclidean / Manhattan; unused for Cosine).
template <int METRIC>
__device__ __forceinline__ bool screen8_decides(float S, float s_row, const float4 max8, const NormalStats8 ns, float gamma_r,
                                                uint32_t &side) {
    // g8 = 2e-6: the conversions and sums of the lane totals (each exact below 2^24) and the scale product
    float e = ns.bn * max8.x + ns.cn * max8.y + 2.0e-6f * (ns.an * max8.x) + gamma_r * (ns.cn0 * max8.z);
    float m = S;
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) {
        // back to real units (s_row = inf for rows that must not be decided here: m and e turn NaN / inf -> undecided),
        // then r = fl(bias + fl_ref(dot)) as in screen_decides
        const float sd = S * s_row;
        e = e * s_row * 1.000001f + 1.2e-7f * fabsf(sd);
        m = ns.extra + sd;
        e += 2.4e-7f * (fabsf(ns.extra) + fabsf(sd) + e);
    }
    e = e * 1.002f + 1e-30f;
    side = (__float_as_uint(m) >> 31) ^ 1u;
    return fabsf(m) > e;  // false for NaN / inf
}

// The level's normals (records [vector][header slot]) -> shadow records [hpitch halves][NormalStats], one wave per node.
__global__ __launch_bounds__(64) void k_forest_shadow_normals(DataView dv, const FNode *__restrict__ nodes,
                                                              const uint8_t *__restrict__ normals, uint64_t nstride,
                                                              uint64_t hdr_off, uint8_t *__restrict__ shadow,
                                                              uint64_t hstride, uint32_t hpitch, uint32_t n_nodes,
                                                              const uint32_t *gate) {
    if (gate_closed(gate)) return;
    for (uint32_t node = blockIdx.x; node < n_nodes; node += gridDim.x) {
    if (nodes[node].state != ST_PENDING) continue;
    const float *nv = reinterpret_cast<const float *>(normals + node * nstride);
    uint16_t *out = reinterpret_cast<uint16_t *>(shadow + node * hstride);
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (uint32_t i = threadIdx.x; i < hpitch; i += 64) {
        const float x = i < dv.dims ? nv[i] : 0.0f;
        const _Float16 h = to_shadow_half(x);
        const float y = (float)h, d = x - y;
        sa += y * y;
        sb += d * d;
        sc += x * x;
        out[i] = __builtin_bit_cast(uint16_t, h);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_xor(sa, off);
        sb += __shfl_xor(sb, off);
        sc += __shfl_xor(sc, off);
    }
    if (threadIdx.x == 0) {
        const float up = 1.0f + (float)(hpitch + 64u) * 1.2e-7f;
        const float *nh = reinterpret_cast<const float *>(normals + node * nstride + hdr_off);
        NormalStats st;
        st.an = sqrtf(sa) * up;
        st.bn = sqrtf(sb) * up;
        st.cn = sqrtf(sc) * up;
        st.extra = dv.metric == AH_COSINE ? 0.0f : nh[0];  // bias (Euclidean / Manhattan) or the normal's extra dimension
        *reinterpret_cast<NormalStats *>(shadow + node * hstride + (uint64_t)hpitch * 2) = st;
    }
    }
}

// screen dot product of one row against one normal, octet-cooperative: lane j covers halves 64k + 8j .. +7
// a4 = normal (LDS or global), r4 = row (global, streamed); steps = hpitch / 64.  Result on every lane of the octet.
__device__ __forceinline__ float screen_octet_dot(const uint4 *a4, const uint4 *r4, uint32_t steps) {
    float acc0 = 0.f, acc1 = 0.f;
    uint32_t k = 0;
    for (; k + 8 <= steps; k += 8) {
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = ld_stream_u4(r4 + (k + u) * 8);
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = screen_dot8(a4[(k + u) * 8], x[u], acc0);
            acc1 = screen_dot8(a4[(k + u + 1) * 8], x[u + 1], acc1);
        }
    }
    if (k + 4 <= steps) {
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = ld_stream_u4(r4 + (k + u) * 8);
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            acc0 = screen_dot8(a4[(k + u) * 8], x[u], acc0);
            acc1 = screen_dot8(a4[(k + u + 1) * 8], x[u + 1], acc1);
        }
        k += 4;
    }
    for (; k < steps; k++) acc0 = screen_dot8(a4[k * 8], ld_stream_u4(r4 + k * 8), acc0);
    return octet_sum(acc0 + acc1);
}

// Node-major margin pass with the screen: as k_forest_margin_f32, but an item costs 2*dims bytes of HBM unless the
// screen cannot decide its side (then the reference arithmetic runs for that item, from the f32 normal kept in LDS).
// PRE > 0 (int8 rows of at most PRE 128-byte steps): the int8 chunks of the NEXT item of an octet are requested before the
// current item is evaluated, and its row index one item earlier still — an item otherwise costs two dependent memory round
// trips (perm -> row) with nothing else in flight on the octet.
template <int METRIC, int PRE>
__global__ __launch_bounds__(kBlock) void k_forest_screen_node(DataView dv, ScreenView sv, FNode *nodes,
                                                               const FTile *__restrict__ tiles, uint32_t n_tiles,
                                                               const uint32_t *__restrict__ perm,
                                                               const uint8_t *__restrict__ normals, uint64_t nstride,
                                                               uint64_t hdr_off, const uint8_t *__restrict__ shadow,
                                                               uint64_t hstride, const uint8_t *__restrict__ shadow8,
                                                               uint64_t stride8, uint64_t *__restrict__ masks,
                                                               uint32_t *__restrict__ tile_left,
                                                               const AbortFlags abort_flag,
                                                               ScreenCounters *__restrict__ counters, uint32_t verify) {
    extern __shared__ float4 s_n4[];  // [pitch floats f32 normal][hpitch halves shadow normal][2 x pitch8 bytes: the int8 digits]
    __shared__ uint32_t s_left, s_fb, s_bad, s_n8, s_d8, s_d8b;
    const float *s_n = reinterpret_cast<const float *>(s_n4);
    const uint4 *s_h4 = reinterpret_cast<const uint4 *>(s_n4 + (dv.pitch >> 2));
    const uint4 *s_q4 = s_h4 + (sv.hpitch >> 3);        // q_hi
    const uint4 *s_ql4 = s_q4 + (sv.pitch8 >> 4);        // q_lo
    const bool stage8 = METRIC != AH_DOT_PRODUCT && sv.rows8 != nullptr && shadow8 != nullptr;  // (DotProduct needs the row's header anyway)
    const uint32_t steps8 = sv.pitch8 >> 7;
    const uint32_t o = threadIdx.x >> 3, j = threadIdx.x & 7u;
    const uint32_t steps = sv.hpitch >> 6;
    uint32_t fallbacks = 0, bad = 0, met8 = 0, decided8 = 0, decided8b = 0;
    if (gate_closed(abort_flag.gate)) return;  // a retry attempt with no node left pending
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (abort_requested(abort_flag)) return;
        const FTile tl = tiles[tile];
        const FNode *nd = nodes + tl.node;
        if (nd->state != ST_PENDING) continue;  // block-uniform
        __syncthreads();
        {
            const float4 *g_n4 = reinterpret_cast<const float4 *>(normals + tl.node * nstride);
            for (uint32_t i = threadIdx.x; i < (dv.pitch >> 2); i += blockDim.x) s_n4[i] = g_n4[i];
            const uint4 *g_h4 = reinterpret_cast<const uint4 *>(shadow + tl.node * hstride);
            uint4 *d_h4 = reinterpret_cast<uint4 *>(s_n4 + (dv.pitch >> 2));
            for (uint32_t i = threadIdx.x; i < (sv.hpitch >> 3); i += blockDim.x) d_h4[i] = g_h4[i];
            if (stage8) {
                const uint4 *g_q4 = reinterpret_cast<const uint4 *>(shadow8 + tl.node * stride8);
                uint4 *d_q4 = d_h4 + (sv.hpitch >> 3);
                for (uint32_t i = threadIdx.x; i < (sv.pitch8 >> 3); i += blockDim.x) d_q4[i] = g_q4[i];  // both digits
            }
        }
        if (threadIdx.x == 0) s_left = 0;
        __syncthreads();
        const float *g_h = reinterpret_cast<const float *>(normals + tl.node * nstride + hdr_off);
        const LeafHdr nh = {g_h[0], g_h[1]};
        const NormalStats ns = *reinterpret_cast<const NormalStats *>(shadow + tl.node * hstride + (uint64_t)sv.hpitch * 2);
        NormalStats8 ns8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (stage8) ns8 = *reinterpret_cast<const NormalStats8 *>(shadow8 + tl.node * stride8 + 2 * (uint64_t)sv.pitch8);
        const uint32_t in_tile = min(kTile, nd->count - tl.first);
        const uint32_t *pp = perm + nd->start + tl.first;
        uint64_t mask = 0;
        uint32_t lefts = 0;
        // software pipeline of the int8 stage (PRE > 0): row index two items ahead, row chunks one item ahead
        constexpr int NPRE = PRE > 0 ? PRE : 1;
        uint4 xn[NPRE];
        uint32_t row_n = 0, row_nn = 0;
        if (PRE > 0 && stage8) {
            if (o < in_tile) row_n = pp[o];
            if (o + 32 < in_tile) row_nn = pp[o + 32];
            if (o < in_tile) {
                const uint4 *r8 = reinterpret_cast<const uint4 *>(sv.rows8 + (uint64_t)row_n * sv.pitch8) + j;
#pragma unroll
                for (int u = 0; u < NPRE; u++)
                    if ((uint32_t)u < steps8) xn[u] = ld_stream_u4(r8 + u * 8);
            }
        }
        for (uint32_t i = 0; i < 64; i++) {
            const uint32_t p = o + 32 * i;
            if (p >= in_tile) break;
            uint64_t row;
            uint4 x[NPRE];
            if (PRE > 0 && stage8) {
                row = row_n;
#pragma unroll
                for (int u = 0; u < NPRE; u++) x[u] = xn[u];
                row_n = row_nn;
                if (p + 64 < in_tile) row_nn = pp[p + 64];
                if (p + 32 < in_tile) {
                    const uint4 *r8n = reinterpret_cast<const uint4 *>(sv.rows8 + (uint64_t)row_n * sv.pitch8) + j;
#pragma unroll
                    for (int u = 0; u < NPRE; u++)
                        if ((uint32_t)u < steps8) xn[u] = ld_stream_u4(r8n + u * 8);
                }
            } else {
                row = pp[p];
            }
            uint32_t side = 0;
            bool decided = false;
            if (stage8) {  // first stage: the int8 copy, 768 bytes of a 768-d row; bound from the dataset-wide maxima
                const uint4 *r8 = reinterpret_cast<const uint4 *>(sv.rows8 + row * sv.pitch8) + j;
                const float s_row = METRIC == AH_COSINE ? 1.0f : sv.scale8_rows[row];  // a cosine margin's sign needs no scale
                const float u8 = PRE > 0 ? screen8_octet_dot_regs<NPRE>(s_q4 + j, s_ql4 + j, x, steps8)
                                         : screen8_octet_dot(s_q4 + j, s_ql4 + j, r8, steps8);
                decided = screen8_decides<METRIC>(u8 * ns8.scale, s_row, sv.max8, ns8, sv.gamma_r, side);
                met8++;
                decided8 += decided ? 1u : 0u;
                if (!decided && sv.rows8_lo != nullptr) {  // octet-uniform: the row's second int8 digit (768 more bytes)
                    const uint4 *r8l = reinterpret_cast<const uint4 *>(sv.rows8_lo + row * sv.pitch8) + j;
                    const float u8b = u8 + screen8_octet_dot(s_q4 + j, s_ql4 + j, r8l, steps8) * 0.00390625f;
                    decided = screen8_decides<METRIC>(u8b * ns8.scale, s_row, sv.max8b, ns8, sv.gamma_r, side);
                    decided8b += decided ? 1u : 0u;
                }
            }
            if (!decided) {  // octet-uniform: second stage, the binary16 copy
                const uint4 *r4 = reinterpret_cast<const uint4 *>(sv.rows + row * sv.hpitch) + j;
                const float s = screen_octet_dot(s_h4 + j, r4, steps);
                const float row_extra = METRIC == AH_DOT_PRODUCT ? dv.headers[2 * row] : 0.0f;
                // the bound from the dataset-wide maxima first (monotone in every stat, so it is >= the row's own bound);
                // the row's stats only when that one cannot decide
                decided = screen_decides<METRIC>(s, sv.max_stats, ns, row_extra, sv.gamma_s, sv.gamma_r, side);
                if (!decided) decided = screen_decides<METRIC>(s, sv.stats[row], ns, row_extra, sv.gamma_s, sv.gamma_r, side);
            }
            if (!decided || verify) {  // octet-uniform
                const uint32_t exact = side_of_margin(margin_f32<METRIC>(dv, s_n, nh, row, j));
                if (decided && exact != side) bad++;
                if (!decided) fallbacks++;
                side = exact;
            }
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
    // statistics: one atomic per block
    __syncthreads();
    if (threadIdx.x == 0) s_fb = s_bad = s_n8 = s_d8 = s_d8b = 0;
    __syncthreads();
    if (j == 0 && fallbacks) atomicAdd(&s_fb, fallbacks);
    if (j == 0 && bad) atomicAdd(&s_bad, bad);
    if (j == 0 && met8) atomicAdd(&s_n8, met8);
    if (j == 0 && decided8) atomicAdd(&s_d8, decided8);
    if (j == 0 && decided8b) atomicAdd(&s_d8b, decided8b);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_fb) atomicAdd(&counters->fallbacks, (unsigned long long)s_fb);
        if (s_bad) atomicAdd(&counters->violations, (unsigned long long)s_bad);
        if (s_n8) atomicAdd(&counters->stage8_pairs, (unsigned long long)s_n8);
        if (s_d8) atomicAdd(&counters->stage8_decided, (unsigned long long)s_d8);
        if (s_d8b) atomicAdd(&counters->stage8b_decided, (unsigned long long)s_d8b);
    }
}

// exact margin of (row, normal record in global memory) by one octet: the arithmetic of k_forest_margin_rows
template <int METRIC>
__device__ __forceinline__ float rows_exact_margin(const DataView &dv, uint64_t row, const uint8_t *nrec, uint64_t hdr_off,
                                                   uint32_t j) {
    const float *rp = dv.rows_f32 + row * dv.pitch;
    const float *np = reinterpret_cast<const float *>(nrec);
    const float d = octet_reduce_stream<OP_DOT>(reinterpret_cast<const float4 *>(np), rp, dv.dims, j);
    const float *nh = reinterpret_cast<const float *>(nrec + hdr_off);
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) return f_add(nh[0], d);
    if (METRIC == AH_DOT_PRODUCT) return f_add(d, f_mul(nh[0], dv.headers[2 * row]));
    return d;
}

// The same bits with every line of the row AND of the normal requested before the first multiply-add (octet_reduce_wide: two
// trips to memory per pair — node index, then both operands — instead of one per eight lines): k_forest_exact_pairs, whose waves
// spend their time between dependent loads (192 registers: for kernels that live on loads in flight, not on occupancy).
template <int METRIC>
__device__ __forceinline__ float rows_exact_margin_wide(const DataView &dv, uint64_t row, const uint8_t *nrec, uint64_t hdr_off,
                                                        uint32_t j) {
    const float *rp = dv.rows_f32 + row * dv.pitch;
    const float *np = reinterpret_cast<const float *>(nrec);
    const float *nh = reinterpret_cast<const float *>(nrec + hdr_off);
    float h0 = 0.0f, rh = 0.0f;
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN || METRIC == AH_DOT_PRODUCT) h0 = nh[0];
    if (METRIC == AH_DOT_PRODUCT) rh = dv.headers[2 * row];
    const float d = octet_reduce_wide<OP_DOT, false>(np, rp, dv.dims, j);  // (normal x row: the operand order of octet_reduce_stream above)
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) return f_add(h0, d);
    if (METRIC == AH_DOT_PRODUCT) return f_add(d, f_mul(h0, rh));
    return d;
}

#ifndef AH_SCREEN_CHUNK
#define AH_SCREEN_CHUNK 8
#endif
// NS steps (64 dims each) of one row against the normals of TC trees.  Branch-free over the trees (a row that is already
// a leaf in tree t reads record 0 for nothing): the loads of a tree's NS chunks are issued together and the compiler is
// free to run ahead into the next trees, so a wave keeps dozens of 16-byte loads in flight.
template <int TC, int NS>
__device__ __forceinline__ void screen_rows_chunk(float (&acc)[TC], const uint4 *base4, const uint32_t (&noff)[TC],
                                                  const uint4 *r4, uint32_t k, bool nt_rows = false) {
    uint4 x[NS];
    if (nt_rows) {
#pragma unroll
        for (int u = 0; u < NS; u++) x[u] = ld_stream_u4(r4 + (k + u) * 8);
    } else {
#pragma unroll
        for (int u = 0; u < NS; u++) x[u] = r4[(k + u) * 8];  // cached on purpose: the other tree groups re-read the chunk
    }
#pragma unroll
    for (int t = 0; t < TC; t++) {
        const uint4 *np = base4 + noff[t] + k * 8;
        uint4 nv[NS];
#pragma unroll
        for (int u = 0; u < NS; u++) nv[u] = np[u * 8];
        float a0 = acc[t], a1 = 0.f;
#pragma unroll
        for (int u = 0; u < NS; u += 2) {
            a0 = screen_dot8(nv[u], x[u], a0);
            a1 = screen_dot8(nv[u + 1], x[u + 1], a1);
        }
        acc[t] = a0 + a1;
    }
}

// Row-major pass with the screen.  LDS_NORMALS: the shadow records of the group's nodes [first_node, +n_group_nodes) are
// resident in LDS (top levels); otherwise they come from L2.  The f32 normals (fallback) always come from global memory.
// The loads of a tree's 8 normal chunks are issued together, ahead of the 32 dot2c that consume them (two independent
// accumulation chains): the pass is bound by the vector-memory pipeline, so what matters is how many loads a wave keeps
// in flight, not the arithmetic.
#ifndef AH_SCREEN_WAVES
#define AH_SCREEN_WAVES 1  // minimum waves per SIMD the compiler must leave room for (caps the VGPRs of the global variant)
#endif
// Schedule of a launch: ONE launch serves `n_groups` groups of TC trees (tree0 + g * TC ...) and blocks are numbered
// chunk-major — block b works on rows [chunk * chunk_rows + tile * rows_per_block, ...) of group g with
// chunk = b / (n_groups * tiles), g = (b / tiles) % n_groups, tile = b % tiles — so the passes of all groups over one chunk
// of rows (48 MB of binary16 rows) run back to back and every pass but the first finds the chunk in the 256 MB Infinity
// Cache (which is also why the blocks must be launched one per work item: a grid-stride loop lets the blocks drift apart,
// the window of rows in flight spreads over many chunks and the pass takes 1.5-2x as long; bigger blocks — 128 / 512 rows —
// lose 20-60 %).  Measured on the pass in isolation (scripts/micro/rows_pass_model.hip): a pass costs the L2 gather of its normals
// PLUS ~4.6 ms of HBM-latency-bound row stream (11.9 ms for 16 trees, 10M rows); with the rows coming from the Infinity
// Cache 9.1-9.6 ms; 7.8 ms if they came from L2.  The row loads must be ordinary cached loads for that (non-temporal
// ones do not stay: 11.1 ms).  group_nodes (LDS variant): first node of every tree in the level, nodes ordered by tree.
struct RowsSchedule {
    uint32_t n_groups, tiles, chunk_rows, rows_per_block;
    uint32_t xcd_slots;  // 0: every (chunk, group) is spread over all XCDs; k > 0: one XCD per (chunk, group), k groups per XCD
                         // (bit 31: stream the rows with non-temporal loads)
    uint32_t chunk0;     // first chunk of this launch (a launch is limited to 2^32 work-items, so big levels take several)
    const uint32_t *tree_first;  // LDS variant: device copy of tree_first[n_trees + 1]
};
// block -> (group, rows) of a row-major launch; false when the block has nothing to do.  Shared by the margin kernel and
// by k_rows_schedule_coverage (the test that every (group, row) pair is served exactly once runs THIS function on the
// device, with the grids plan_rows_launches hands the build).
__device__ __forceinline__ bool rows_block_map(const RowsSchedule &sch, uint32_t b, uint64_t n_rows, uint32_t &group,
                                               uint64_t &row_begin, uint64_t &row_end) {
    uint32_t chunk, tile;
    if (sch.xcd_slots & 0x7FFFFFFFu) {
        // One XCD per (chunk, group): workgroups go to the XCDs round-robin (block b -> XCD b & 7), so XCD x takes, for
        // chunk c, the groups g = ((x - c) mod 8) + 8 k — the rotation by c evens out n_groups mod 8 over the chunks.  An
        // XCD's L2 (4 MiB) then holds the normals of ONE group next to the row stream instead of those of the two
        // groups that are in flight at any time when every group is spread over all eight.
        const uint32_t x = b & 7u, slot = b >> 3;
        const uint32_t per_chunk = (sch.xcd_slots & 0x7FFFFFFFu) * sch.tiles;
        chunk = sch.chunk0 + slot / per_chunk;
        const uint32_t rem = slot % per_chunk;
        tile = rem % sch.tiles;
        group = ((x + 8u - (chunk & 7u)) & 7u) + 8u * (rem / sch.tiles);
        if (group >= sch.n_groups) return false;
    } else {
        const uint32_t per_chunk = sch.n_groups * sch.tiles;
        chunk = sch.chunk0 + b / per_chunk;
        const uint32_t in_chunk = b % per_chunk;
        group = in_chunk / sch.tiles;
        tile = in_chunk % sch.tiles;
    }
    row_begin = (uint64_t)chunk * sch.chunk_rows + (uint64_t)tile * sch.rows_per_block;
    row_end = min(min(row_begin + sch.rows_per_block, (uint64_t)(chunk + 1) * sch.chunk_rows), n_rows);
    return row_begin < row_end;
}
// test aid (ah_debug_launch_coverage): counts[group * n_rows + row] += 1 for every row the block would serve
__global__ void k_rows_schedule_coverage(RowsSchedule sch, uint64_t n_rows, uint32_t *__restrict__ counts) {
    uint32_t group;
    uint64_t row_begin, row_end;
    if (!rows_block_map(sch, blockIdx.x, n_rows, group, row_begin, row_end)) return;
    for (uint64_t row = row_begin + (threadIdx.x >> 3); row < row_end; row += blockDim.x >> 3)
        if ((threadIdx.x & 7u) == 0) atomicAdd(&counts[(uint64_t)group * n_rows + row], 1u);
}

template <int METRIC, int TC, bool LDS_NORMALS>
__global__ __launch_bounds__(LDS_NORMALS ? (TC >= 16 ? 512 : 1024) : kBlock,
                             LDS_NORMALS ? 1 : (AH_SCREEN_WAVES > 1 ? AH_SCREEN_WAVES : (TC == 8 ? 4 : 1))) void k_forest_screen_rows(
    DataView dv, ScreenView sv, const uint32_t *__restrict__ node_of, uint32_t tree_base, uint32_t n_pass,
    const uint8_t *__restrict__ normals, uint64_t nstride, uint64_t hdr_off, const uint8_t *__restrict__ shadow,
    uint64_t hstride, uint8_t *__restrict__ side_bytes, RowsSchedule sch, const AbortFlags abort_flag,
    ScreenCounters *__restrict__ counters, uint32_t verify) {
    extern __shared__ uint4 s_shadow4[];
    __shared__ uint32_t s_fb, s_bad;
    if (abort_requested(abort_flag)) return;
    uint32_t group;
    uint64_t row_begin, row_end;
    if (!rows_block_map(sch, blockIdx.x, dv.n, group, row_begin, row_end)) return;  // block-uniform
    const uint32_t tree0 = tree_base + group * TC;
    const uint32_t hstride4 = (uint32_t)(hstride >> 4);
    uint32_t first_node = 0;
    if (LDS_NORMALS) {
        first_node = sch.tree_first[tree0];
        const uint32_t n_group_nodes = sch.tree_first[tree0 + n_pass] - first_node;
        const uint4 *g = reinterpret_cast<const uint4 *>(shadow + (uint64_t)first_node * hstride);
        const uint32_t total = n_group_nodes * hstride4;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) s_shadow4[i] = g[i];
    }
    if (threadIdx.x == 0) s_fb = s_bad = 0;
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u;
    const uint32_t octets = blockDim.x >> 3;
    const uint32_t steps = sv.hpitch >> 6;
    const uint32_t stats4 = sv.hpitch >> 3;  // uint4 index of the NormalStats inside a shadow record
    // all shadow records of the level, as uint4: LDS copy of the group or the global chunk (32-bit indices either way)
    const uint4 *base4 = LDS_NORMALS ? s_shadow4 : reinterpret_cast<const uint4 *>(shadow);
    const uint32_t node0 = first_node;
    const uint32_t last_tree = n_pass - 1;
    constexpr int N = TC >= 8 ? 8 : TC;        // trees per transposing reduction
    constexpr int SETS = TC >= 8 ? TC / 8 : 1;  // sets of N trees
    const uint32_t my_t = octet_owned_tree<N>(j);
    const bool owner = octet_is_owner<N>(j);
    uint32_t fallbacks = 0, bad = 0;
    for (uint64_t row = row_begin + (threadIdx.x >> 3); row < row_end; row += octets) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(sv.rows + row * sv.hpitch) + j;
        // Lane j owns tree octet_owned_tree(j) of each set of 8 trees: it loads that tree's node index (one load instruction
        // for 8 trees), broadcasts it to the octet, and later runs the tree's epilogue.  n_pass == TC except for the last
        // trees of a forest: spare slots repeat the last tree and are not stored.
        uint32_t my_node[SETS];
        uint32_t noff[TC];  // uint4 index of (record of the node that owns the row in tree t) + j
        float acc[TC];
#pragma unroll
        for (int st = 0; st < SETS; st++)
            my_node[st] = node_of[(uint64_t)(tree0 + min(my_t + 8u * st, last_tree)) * dv.n + row];
#pragma unroll
        for (int t = 0; t < TC; t++) {
            const uint32_t node = (uint32_t)__shfl((int)my_node[t / 8], (int)octet_owner_lane<N>(t % 8), 8);
            noff[t] = (node != 0xFFFFFFFFu ? node - node0 : 0u) * hstride4 + j;
            acc[t] = 0.f;
        }
        // chunks of AH_SCREEN_CHUNK steps (64 dims each), then of 4, then single steps.  Measured on 10M x 768 (12 steps):
        // 8 + 4 beats a single chunk of 12 (the extra 16 row registers cost a wave of occupancy per SIMD: 7.7 -> 10.8 ms
        // per 8-tree pass).
        uint32_t k = 0;
        const bool nt_rows = (sch.xcd_slots & 0x80000000u) != 0;
        for (; k + AH_SCREEN_CHUNK <= steps; k += AH_SCREEN_CHUNK) screen_rows_chunk<TC, AH_SCREEN_CHUNK>(acc, base4, noff, r4, k, nt_rows);
#if AH_SCREEN_CHUNK > 4
        if (k + 4 <= steps) {
            screen_rows_chunk<TC, 4>(acc, base4, noff, r4, k, nt_rows);
            k += 4;
        }
#endif
        for (; k < steps; k++) {
            const uint4 x = r4[k * 8];
#pragma unroll
            for (int t = 0; t < TC; t++) acc[t] = screen_dot8(base4[noff[t] + k * 8], x, acc[t]);
        }
        // Epilogue, once per tree on its owner lane: total of the octet (transposing reduction), error bound, decision.
        // Pairs the screen cannot decide are recomputed in the reference arithmetic by the whole octet.
        const float4 rs = sv.stats[row];
        const float row_extra = METRIC == AH_DOT_PRODUCT ? dv.headers[2 * row] : 0.0f;
#pragma unroll
        for (int st = 0; st < SETS; st++) {
            const float total = octet_transpose_sum<N>(acc + 8 * st, j);
            const uint32_t t = my_t + 8u * st;
            const bool valid = owner && t <= last_tree && my_node[st] != 0xFFFFFFFFu;
            uint32_t side = 0;
            bool decided = false;
            if (valid) {
                const uint4 raw = base4[(my_node[st] - node0) * hstride4 + stats4];
                NormalStats ns;
                ns.an = __uint_as_float(raw.x);
                ns.bn = __uint_as_float(raw.y);
                ns.cn = __uint_as_float(raw.z);
                ns.extra = __uint_as_float(raw.w);
                decided = screen_decides<METRIC>(total, rs, ns, row_extra, sv.gamma_s, sv.gamma_r, side);
            }
            const unsigned long long und = __ballot(valid && (!decided || verify));
            const uint32_t mask8 = (uint32_t)(und >> (8u * ((threadIdx.x & 63u) >> 3))) & 0xFFu;
            if (mask8) {  // octet-uniform, ~8 % of the (octet, set) pairs
#pragma unroll
                for (int tt = 0; tt < N; tt++) {
                    constexpr uint32_t kOwner[8] = {octet_owner_lane<N>(0), octet_owner_lane<N>(1), octet_owner_lane<N>(2),
                                                    octet_owner_lane<N>(3), octet_owner_lane<N>(4), octet_owner_lane<N>(5),
                                                    octet_owner_lane<N>(6), octet_owner_lane<N>(7)};
                    const uint32_t ol = kOwner[tt];
                    if ((mask8 >> ol) & 1u) {
                        const uint32_t node = (noff[8 * st + tt] - j) / hstride4 + node0;
                        const uint32_t exact =
                            side_of_margin(rows_exact_margin<METRIC>(dv, row, normals + (uint64_t)node * nstride, hdr_off, j));
                        if (j == ol) {
                            if (decided && exact != side) bad++;
                            if (!decided) fallbacks++;
                            side = exact;
                        }
                    }
                }
            }
            if (valid) side_bytes[(uint64_t)(tree0 + t) * dv.n + row] = (uint8_t)side;
        }
    }
    if (fallbacks) atomicAdd(&s_fb, fallbacks);
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_fb) atomicAdd(&counters->fallbacks, (unsigned long long)s_fb);
        if (s_bad) atomicAdd(&counters->violations, (unsigned long long)s_bad);
    }
}

// ------------------------------------------------------------------------------------------------
// The next level, built on the device: children of every split node that are still too large for a Descendants node
// (src/writer.rs:474-477) become the next level's nodes, in node order, left child first — the order the host used to
// produce.  Three small kernels (count per block, scan of the block sums, emit) plus the tile list; the host reads back
// only the LevelInfo block before it launches the level, the node tables follow asynchronously for the final node list.
// ------------------------------------------------------------------------------------------------
struct NextCounts {
    uint32_t kids, tiles;
};
__device__ __forceinline__ void child_counts(const FNode &nd, uint32_t split_after, uint32_t cnt[2], uint32_t split[2]) {
    cnt[0] = nd.n_left;
    cnt[1] = nd.count - nd.n_left;
    split[0] = cnt[0] > split_after ? 1u : 0u;
    split[1] = cnt[1] > split_after ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_next_count(const FNode *__restrict__ nodes, uint32_t n_nodes, uint32_t split_after,
                                                    NextCounts *__restrict__ block_sums) {
    __shared__ uint32_t s_k[4], s_t[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t kids = 0, tiles = 0;
    if (i < n_nodes) {
        uint32_t cnt[2], sp[2];
        child_counts(nodes[i], split_after, cnt, sp);
        kids = sp[0] + sp[1];
        tiles = sp[0] * ((cnt[0] + kTile - 1) / kTile) + sp[1] * ((cnt[1] + kTile - 1) / kTile);
    }
    for (int off = 32; off > 0; off >>= 1) {
        kids += __shfl_xor(kids, off);
        tiles += __shfl_xor(tiles, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        s_k[threadIdx.x >> 6] = kids;
        s_t[threadIdx.x >> 6] = tiles;
    }
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = NextCounts{s_k[0] + s_k[1] + s_k[2] + s_k[3], s_t[0] + s_t[1] + s_t[2] + s_t[3]};
}
// exclusive scan of the block sums in place (one block; n_blocks <= a few thousand) + the level totals
__global__ __launch_bounds__(1024) void k_next_scan(NextCounts *__restrict__ block_sums, uint32_t n_blocks,
                                                    LevelInfo *__restrict__ info, uint32_t *__restrict__ tree_first,
                                                    uint32_t n_trees, uint32_t *__restrict__ pend) {
    __shared__ uint32_t s_k[16], s_t[16];
    __shared__ uint32_t s_carry_k, s_carry_t;
    if (threadIdx.x == 0) s_carry_k = s_carry_t = 0;
    if (threadIdx.x < 3) pend[threadIdx.x] = 0;  // the level's attempts are over: the next level counts its own pending nodes
    for (uint32_t t = threadIdx.x; t <= n_trees; t += blockDim.x) tree_first[t] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const NextCounts v = i < n_blocks ? block_sums[i] : NextCounts{0u, 0u};
        uint32_t ik = v.kids, it = v.tiles;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t uk = __shfl_up(ik, off), ut = __shfl_up(it, off);
            if ((int)lane >= off) {
                ik += uk;
                it += ut;
            }
        }
        if (lane == 63) {
            s_k[wave] = ik;
            s_t[wave] = it;
        }
        __syncthreads();
        uint32_t wk = 0, wt = 0;
        for (uint32_t w = 0; w < wave; w++) {
            wk += s_k[w];
            wt += s_t[w];
        }
        const uint32_t ck = s_carry_k, ct = s_carry_t;
        if (i < n_blocks) block_sums[i] = NextCounts{ck + wk + ik - v.kids, ct + wt + it - v.tiles};
        __syncthreads();
        if (threadIdx.x == 1023) {
            s_carry_k = ck + wk + ik;
            s_carry_t = ct + wt + it;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        info->n_nodes = s_carry_k;
        info->n_tiles = s_carry_t;
        info->n_fix = 0;
        info->pairs = 0;
    }
}
__global__ __launch_bounds__(256) void k_next_emit(const FNode *__restrict__ nodes, uint32_t n_nodes, uint32_t split_after,
                                                   const NextCounts *__restrict__ block_sums, FNode *__restrict__ next,
                                                   uint32_t *__restrict__ child, LevelInfo *__restrict__ info,
                                                   uint32_t *__restrict__ tree_first) {
    __shared__ uint32_t s_k[4], s_t[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    FNode nd{};
    uint32_t cnt[2] = {0, 0}, sp[2] = {0, 0};
    if (i < n_nodes) {
        nd = nodes[i];
        child_counts(nd, split_after, cnt, sp);
    }
    const uint32_t kids = sp[0] + sp[1];
    const uint32_t nt[2] = {sp[0] * ((cnt[0] + kTile - 1) / kTile), sp[1] * ((cnt[1] + kTile - 1) / kTile)};
    uint32_t ik = kids, it = nt[0] + nt[1];
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t uk = __shfl_up(ik, off), ut = __shfl_up(it, off);
        if ((int)lane >= off) {
            ik += uk;
            it += ut;
        }
    }
    if (lane == 63) {
        s_k[wave] = ik;
        s_t[wave] = it;
    }
    __syncthreads();
    uint32_t idx = block_sums[blockIdx.x].kids + ik - kids, tidx = block_sums[blockIdx.x].tiles + it - (nt[0] + nt[1]);
    for (uint32_t w = 0; w < wave; w++) {
        idx += s_k[w];
        tidx += s_t[w];
    }
    if (i >= n_nodes) return;
    // sides of a first-attempt accept are the ones a row-major pass left in side_bytes; anything else was re-drawn
    const uint32_t fix = (nd.state == ST_ACCEPTED && nd.attempt == 0) ? 0u : 1u;
    unsigned long long pairs = 0;
    uint32_t n_fix = 0;
    for (uint32_t side = 0; side < 2; side++) {
        uint32_t link = 0xFFFFFFFFu;
        if (sp[side]) {
            FNode cn{};
            cn.key = ah_node_key_child(nd.key, side);
            cn.start = side ? nd.start + nd.n_left : nd.start;
            cn.tree = nd.tree;
            cn.count = cnt[side];
            cn.tile_begin = tidx;
            cn.n_tiles = nt[side];
            cn.fix = fix;
            next[idx] = cn;
            if (!fix) link = idx;
            atomicMin(&tree_first[nd.tree], idx);
            pairs += cnt[side];
            n_fix += fix;
            idx++;
            tidx += nt[side];
        }
        child[2 * (uint64_t)i + side] = link;
    }
    if (pairs) atomicAdd(&info->pairs, pairs);
    if (n_fix) atomicAdd(&info->n_fix, n_fix);
}
// A level cut into groups of trees (the tail of build_batch): per group the tiles and the items under its nodes, and its
// nodes' tile indices rebased to the group's first tile — from there on the group is a level of its own over the shared
// tile tables.  One block per group; nodes [first[g], first[g + 1]) of the level's table.
struct GroupCuts {
    uint32_t first[33];
};
struct GroupInfo {
    uint32_t n_tiles, n_fix;
    unsigned long long pairs;
};
__global__ __launch_bounds__(256) void k_fork_groups(FNode *__restrict__ nodes, GroupCuts cuts, GroupInfo *__restrict__ out) {
    __shared__ unsigned long long s_pairs[4];
    __shared__ uint32_t s_fix[4];
    const uint32_t lo = cuts.first[blockIdx.x], hi = cuts.first[blockIdx.x + 1];
    uint32_t t0 = 0, t1 = 0;
    if (hi > lo) {
        t0 = nodes[lo].tile_begin;
        t1 = nodes[hi - 1].tile_begin + nodes[hi - 1].n_tiles;
    }
    __syncthreads();  // every thread holds the first tile before the first node is rebased
    unsigned long long pairs = 0;
    uint32_t fix = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        pairs += nodes[i].count;
        fix += nodes[i].fix;
        nodes[i].tile_begin -= t0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        pairs += __shfl_xor(pairs, off);
        fix += __shfl_xor(fix, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        s_pairs[threadIdx.x >> 6] = pairs;
        s_fix[threadIdx.x >> 6] = fix;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = GroupInfo{t1 - t0, s_fix[0] + s_fix[1] + s_fix[2] + s_fix[3], s_pairs[0] + s_pairs[1] + s_pairs[2] + s_pairs[3]};
}
// tile list of a level: one wave per node
__global__ __launch_bounds__(256) void k_build_tiles(const FNode *__restrict__ nodes, uint32_t n_nodes,
                                                     FTile *__restrict__ tiles) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t node = blockIdx.x * 4 + (threadIdx.x >> 6); node < n_nodes; node += n_waves) {
        const uint32_t begin = nodes[node].tile_begin, nt = nodes[node].n_tiles;
        for (uint32_t t = lane; t < nt; t += 64) tiles[begin + t] = FTile{node, t * kTile};
    }
}

}  // namespace ah

#include "dense_device.h"

using namespace ah;

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// ---- host memory of the forests' blobs, recycled --------------------------------------------------------------------
// A 10M x 768 x 100-tree forest is 5.4 GB of normals + 4 GB of item ids.  Handing every build FRESH pageable memory means
// 2.3 M first-touch page faults per build, taken by a dozen threads of one process that contend for the address-space lock
// with each other and with the HIP runtime's own mappings — measured: 8 threads commit 4 KiB pages at 1.3 GB/s in total,
// 2 threads at 4.9, and the round-3 driver run saw the same build take 1.47 .. 1.82 s.  So: (1) blobs are mapped with
// MADV_HUGEPAGE on 2 MiB boundaries (512x fewer faults where transparent huge pages are available: 7.3 GB/s on 8 threads),
// and (2) the blobs of a destroyed forest are KEPT, committed, in a process-wide pool and handed to the next build, which
// then faults nothing at all.  AH_HOST_CACHE_MB bounds what the pool holds (0 = recycle nothing); ah_host_cache_trim
// returns it to the system.
struct HostBlob {
    uint8_t *p = nullptr;
    size_t cap = 0;        // usable bytes at p
    size_t committed = 0;  // bytes from p whose pages have been written before (a recycled blob): nothing to fault there
    bool recycled = false; // came out of the pool (HostBlobPool::take), not from a fresh mapping
    void *map = nullptr;   // the mapping p lives in (nullptr: malloc'd — small blobs)
    size_t map_len = 0;
    int node = -1;         // host NUMA node its pages prefer (-1: none asked for)
};
// the host node of the device whose build is running on this thread (build_batch sets it): what fresh blobs prefer, and what
// the pool matches recycled blobs against
static thread_local int tl_blob_node = -1;
class HostBlobPool {
    std::mutex mu;
    std::vector<HostBlob> idle;
    size_t held = 0;  // committed bytes of the idle blobs
    static constexpr size_t kSmall = 16u << 20, kHuge = 2u << 20;
    static void release(HostBlob &b) {
        if (b.map) munmap(b.map, b.map_len);
        else free(b.p);
        b = HostBlob{};
    }

  public:
    // a blob of at least `bytes` bytes; false = out of memory
    bool take(size_t bytes, HostBlob *out) {
        bytes += 16;
        if (bytes >= kSmall) {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = idle.size();
            // the smallest blob that fits, and never one more than twice the size asked for: a 20 MB request must not walk off
            // with the 5 GB blob the next big build is counting on
            // ... of those on the asking device's host node first (a blob committed on the far socket halves the rate its copy
            // threads reach), then of the others
            for (int pass = 0; pass < 2 && best == idle.size(); pass++)
                for (size_t i = 0; i < idle.size(); i++) {
                    if (pass == 0 && tl_blob_node >= 0 && idle[i].node != tl_blob_node) continue;
                    if (idle[i].cap >= bytes && idle[i].cap <= 2 * bytes + kHuge && (best == idle.size() || idle[i].cap < idle[best].cap))
                        best = i;
                }
            if (best != idle.size()) {
                *out = idle[best];
                out->recycled = true;
                held -= std::min(held, idle[best].committed);
                idle.erase(idle.begin() + (ptrdiff_t)best);
                return true;
            }
        }
        if (fail_alloc_tick()) return false;
        HostBlob b;
        if (bytes < kSmall) {
            b.p = reinterpret_cast<uint8_t *>(malloc(bytes));
            if (!b.p) return false;
            b.cap = bytes;
        } else {
            const size_t cap = (bytes + kHuge - 1) & ~(kHuge - 1);
            void *m = mmap(nullptr, cap + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (m == MAP_FAILED) return false;
            b.map = m;
            b.map_len = cap + kHuge;
            b.p = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(m) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
            b.cap = cap;
            (void)madvise(b.p, cap, MADV_HUGEPAGE);  // a hint: 4 KiB pages where it is refused
            // the pages prefer the host node of the device whose build asks (api.hip "NUMA placement"; tl_blob_node < 0: wherever
            // they are first touched)
            numa_prefer_node(b.p, cap, tl_blob_node);
            b.node = tl_blob_node;
        }
        *out = b;
        return true;
    }
    void give(HostBlob &b, size_t used) {
        if (!b.p) return;
        NoFailScope no_fail;
        b.committed = std::min(b.cap, std::max(b.committed, used));
        const size_t limit = (size_t)std::max<long long>(0, tun(TUN_HOST_CACHE_MB)) << 20;
        if (!b.map || limit == 0) {
            release(b);
            return;
        }
        std::lock_guard<std::mutex> lk(mu);
        idle.push_back(b);
        held += b.committed;
        b = HostBlob{};
        while (held > limit && !idle.empty()) {  // over the budget: the oldest go first
            held -= std::min(held, idle.front().committed);
            release(idle.front());
            idle.erase(idle.begin());
        }
    }
    size_t trim() {
        NoFailScope no_fail;
        std::lock_guard<std::mutex> lk(mu);
        const size_t was = held;
        for (HostBlob &b : idle) release(b);
        idle.clear();
        held = 0;
        return was;
    }
};
static HostBlobPool &host_pool() {
    static HostBlobPool *pool = new HostBlobPool();  // never destroyed: forests may outlive static destruction order
    return *pool;
}
static size_t records_pool_trim();  // (the node records of the last build, below)
namespace ah {
size_t host_cache_trim() { return host_pool().trim() + records_pool_trim() + pinned_spare_trim(); }
}  // namespace ah
// grow `blob` to at least `need` bytes keeping its first `keep` bytes; the old mapping goes back to the pool
static bool host_blob_reserve(HostBlob &blob, size_t need, size_t keep) {
    if (need + 16 <= blob.cap) return true;
    HostBlob grown;
    if (!host_pool().take(need, &grown)) return false;
    if (keep) memcpy(grown.p, blob.p, keep);
    grown.committed = std::max(grown.committed, keep);
    host_pool().give(blob, keep);
    blob = grown;
    return true;
}

struct ah_forest {
    std::vector<uint32_t> roots;
    std::vector<ah_node> nodes;
    HostBlob normals_blob, desc_blob;  // raw buffers: filled by D2H copies only, never repacked; recycled (HostBlobPool)
    uint8_t *normals = nullptr;        // = normals_blob.p
    uint64_t normals_len = 0;
    uint32_t *descendants = nullptr;   // = desc_blob.p
    uint64_t descendants_len = 0;
    uint64_t normal_stride = 0, normal_vector_offset = 0, normal_header_offset = 0;
    ah_build_stats stats{};
    ~ah_forest() {
        host_pool().give(normals_blob, normals_len);
        host_pool().give(desc_blob, descendants_len * 4);
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
// The records of a build and their index vector, kept for the next build of the process like the forests' blobs (and under
// the same switch, AH_HOST_CACHE_MB; ah_host_cache_trim frees them): 136 MB + 14 MB at 10M x 100 trees, whose fresh pages
// used to be faulted in under the levels by the launching thread and unmapped — 11 ms — between the last copy and the return.
struct RecordsPool {
    std::mutex mu;
    std::vector<HostRec> recs;
    std::vector<uint32_t> index;
};
static RecordsPool &records_pool() {
    static RecordsPool *pool = new RecordsPool();  // never destroyed (see host_pool)
    return *pool;
}
struct RecordsLease {  // takes the pooled vectors for one batch and gives the larger ones back
    std::vector<HostRec> &recs;
    std::vector<uint32_t> &index;
    RecordsLease(std::vector<HostRec> &r, std::vector<uint32_t> &i) : recs(r), index(i) {
        if (tun(TUN_HOST_CACHE_MB) <= 0) return;
        RecordsPool &pool = records_pool();
        std::lock_guard<std::mutex> lk(pool.mu);
        recs.swap(pool.recs);
        index.swap(pool.index);
        recs.clear();
        index.clear();
    }
    ~RecordsLease() {
        if (tun(TUN_HOST_CACHE_MB) <= 0) return;
        RecordsPool &pool = records_pool();
        std::lock_guard<std::mutex> lk(pool.mu);
        if (recs.capacity() > pool.recs.capacity()) recs.swap(pool.recs);
        if (index.capacity() > pool.index.capacity()) index.swap(pool.index);
    }
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return AH_OK;
        if (p) AH_HIP(dev_free(p));
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)1024);
        AH_HIP(dev_malloc((void **)&p, want * sizeof(T)));
        cap = want;
        return AH_OK;
    }
    void release() {
        if (p) (void)dev_free(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
};

// Device memory for the normal records of all levels of a batch: a few big blocks handed out level by level (a level's
// records are contiguous), so that a level costs no hipMalloc.  Everything is freed when the batch ends.
struct Arena {
    std::vector<void *> blocks;
    uint8_t *cur = nullptr;
    size_t left = 0, block_bytes = 0;
    int take(size_t bytes, uint8_t **out) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) {
            size_t got = std::max(bytes, block_bytes);
            void *p = nullptr;
            blocks.reserve(blocks.size() + 1);  // (may throw: BEFORE the block exists, so that it cannot be lost)
            hipError_t e = dev_malloc(&p, got);
            if (e != hipSuccess && got > bytes) {  // tight on memory: exactly this level
                (void)hipGetLastError();
                got = bytes;
                e = dev_malloc(&p, got);
            }
            if (e != hipSuccess) {
                set_error("hipMalloc of %zu bytes of normals failed: %s", bytes, hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE;
            }
            blocks.push_back(p);
            cur = reinterpret_cast<uint8_t *>(p);
            left = got;
        }
        *out = cur;
        cur += bytes;
        left -= bytes;
        return AH_OK;
    }
    ~Arena() {
        for (void *p : blocks) (void)dev_free(p);
    }
};

// Measurement and test aids: the tunables of common.h (environment variable at load time, ah_tuning_set at run time).
// The per-call knob is ah_build_options.margin_mode; the tunables only steer AH_MARGIN_AUTO and the schedule of a level.
// None of them changes a result.
constexpr size_t kLdsNormalsBytes = 128u << 10;  // LDS given to the normals of one tree group (of 160 KiB per CU)
#define AH_DBG(s, what)                                                       \
    do {                                                                      \
        if (tun(TUN_DEBUG)) {                                                 \
            hipError_t _e = hipStreamSynchronize(s);                          \
            fprintf(stderr, "[ah] %s: %s\n", what, hipGetErrorString(_e));    \
            fflush(stderr);                                                   \
        }                                                                     \
    } while (0)

struct BatchCleanup {  // events / side stream of one batch
    hipEvent_t ev_attempt[8] = {};  // begin / end of the margin pass of attempts 0..3, re-used by every level
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_level = nullptr, ev_copy[2] = {nullptr, nullptr};
    hipStream_t side = nullptr;
    int create() {
        for (hipEvent_t &e : ev_attempt) AH_HIP(hipEventCreate(&e));
        AH_HIP(hipEventCreate(&ev_begin));
        AH_HIP(hipEventCreate(&ev_end));
        AH_HIP(hipEventCreateWithFlags(&ev_level, hipEventDisableTiming));
        AH_HIP(hipEventCreateWithFlags(&ev_copy[0], hipEventDisableTiming));
        AH_HIP(hipEventCreateWithFlags(&ev_copy[1], hipEventDisableTiming));
        AH_HIP(create_copy_stream(&side));  // (node tables, the abort word, a group's rows -> ids: under the level's kernels)
        return AH_OK;
    }
    ~BatchCleanup() {
        if (side) (void)hipStreamSynchronize(side);
        for (hipEvent_t e : ev_attempt)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_begin, ev_end, ev_level, ev_copy[0], ev_copy[1]})
            if (e) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
    }
};

// Measured cost of one row-major pass in ns per row, as a function of the tree-group size and of the bytes of normals
// the group streams in the level (profiles/r02_*: 10M x 768 x 100 trees, every level forced to one group size), for
// the f32 kernels (rows of 3072 bytes) and for the screened kernels (binary16 rows of 1536 bytes).  While the normals fit
// the L2s a pass is bound by the vector-memory pipeline — one 16-byte load per 4 FMAs / 4 dot2c — and costs more as the
// octets of a wave stop sharing normals (deeper levels); beyond ~10 MB they come through the fabric and every group size
// converges to (1 + tc) rows' worth of traffic at ~8 TB/s.  Linear interpolation between the measured points.
struct CostPt {
    double mb, ns;
};
template <size_t N>
double interp(const CostPt (&t)[N], double mb) {
    if (mb <= t[0].mb) return t[0].ns;
    for (size_t i = 1; i < N; i++)
        if (mb <= t[i].mb) return t[i - 1].ns + (t[i].ns - t[i - 1].ns) * (mb - t[i - 1].mb) / (t[i].mb - t[i - 1].mb);
    return t[N - 1].ns;
}
double rows_pass_ns_per_row(uint32_t tc, double ws_mb, bool screened) {
    static const CostPt e16[] = {{0.4, 1.60}, {3.2, 1.83}, {6.3, 2.05}, {12.6, 2.33}, {25, 5.08}, {50, 6.05}, {100, 6.5}, {400, 6.8}};
    static const CostPt e8[] = {{0.2, 1.16}, {1.6, 1.22}, {3.2, 2.0}, {6.3, 2.5}, {12.6, 3.1}, {25, 3.5}, {100, 3.7}};
    static const CostPt e4[] = {{0.1, 0.73}, {0.8, 0.85}, {3.2, 0.88}, {6.3, 1.07}, {12.6, 1.45}, {25, 1.85}, {50, 2.05}, {100, 2.14}};
    static const CostPt e2[] = {{0.05, 0.49}, {0.8, 0.60}, {3.1, 0.62}, {6.3, 0.74}, {12.6, 0.99}, {25, 1.14}, {50, 1.21}};
    // screened kernels, chunk-major launches, epilogue on owner lanes (gpurun r02w: 96 trees, every level forced)
    static const CostPt s16[] = {{0.025, 1.11}, {0.2, 1.13}, {0.4, 1.20}, {0.8, 1.25}, {1.6, 1.30}, {3.2, 1.45}, {6.4, 1.99},
                                 {12.7, 2.94}, {25, 3.57}, {51, 3.92}, {102, 4.10}, {203, 4.21}};
    static const CostPt s8[] = {{0.012, 0.58}, {0.1, 0.61}, {0.2, 0.67}, {0.4, 0.71}, {0.8, 0.74}, {1.6, 0.77}, {3.2, 0.85},
                                {6.4, 1.24}, {12.7, 1.66}, {25, 1.94}, {51, 2.09}, {102, 2.15}};
    static const CostPt s4[] = {{0.025, 0.325}, {0.05, 0.353}, {0.1, 0.415}, {0.2, 0.463}, {0.4, 0.493}, {0.8, 0.511},
                                {1.7, 0.45}, {3.4, 0.50}, {6.8, 0.69}, {13.6, 0.90}, {27, 1.03}, {54, 1.10}};  // one XCD per group, nt rows > 5 MB
    static const CostPt s2[] = {{0.003, 0.235}, {0.05, 0.243}, {0.1, 0.261}, {0.2, 0.274}, {0.4, 0.284}, {0.8, 0.292},
                                {1.6, 0.314}, {3.2, 0.389}, {6.4, 0.49}, {12.7, 0.56}, {25, 0.60}};
    if (screened) return tc >= 16 ? interp(s16, ws_mb) : tc == 8 ? interp(s8, ws_mb) : tc == 4 ? interp(s4, ws_mb) : interp(s2, ws_mb);
    return tc >= 16 ? interp(e16, ws_mb) : tc == 8 ? interp(e8, ws_mb) : tc == 4 ? interp(e4, ws_mb) : interp(e2, ws_mb);
}

// index into ah_build_stats.margin_mode_launches
enum { MM_NODE = 0, MM_ROWS2 = 1, MM_ROWS4 = 2, MM_ROWS8 = 3, MM_ROWS16 = 4, MM_LDS8 = 5, MM_LDS16 = 6, MM_BQ = 7 };
inline int mm_rows(uint32_t tc) { return tc >= 16 ? MM_ROWS16 : tc == 8 ? MM_ROWS8 : tc == 4 ? MM_ROWS4 : MM_ROWS2; }

}  // namespace

// Size of a normal record [vector, row_bytes][header, 16-byte slot][padding].  For the f32 metrics it is rounded up to
// whole 128-byte lines (768-d: 3088 -> 3200 bytes; the binary16 shadow records likewise, 1552 -> 1664): the row-major
// passes gather normals as 128-byte pieces, one per octet and load instruction, and a piece that straddles two lines
// costs the L1 three 64-byte accesses instead of two — measured with TCP_TOTAL_CACHE_ACCESSES, that was 6.1e9 accesses
// per 16-tree pass against 4.1e9 useful, in a pass that runs at ~80 % of the L1's access rate.  Callers see the stride
// in ah_forest_view.normal_stride.
static size_t records_pool_trim() {
    RecordsPool &pool = records_pool();
    std::vector<HostRec> r;
    std::vector<uint32_t> i;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        r.swap(pool.recs);
        i.swap(pool.index);
    }
    return r.capacity() * sizeof(HostRec) + i.capacity() * sizeof(uint32_t);
}
static uint64_t normal_record_stride(const ah_dataset *ds) {
    const uint64_t raw = ds->row_bytes() + 16;
    return metric_is_bq(ds->metric) ? raw : (raw + 127) & ~(uint64_t)127;
}

// Binary16 shadow of an f32 dataset (screen_device.h) and the int8 copy of the first node-major stage, built by the first
// forest build that wants them.  Returns false when the screen cannot be used: never for 1-bit metrics / short vectors,
// and — NOT remembered, the next build tries again — when the memory for the copies is not available right now (another
// build's arenas may be live); ah_build_stats.screen_unavailable then says why the build ran in f32 arithmetic only.
static bool ensure_screen8(ah_dataset *ds, hipStream_t s, bool force, bool want_lo = true);
// (ensure_screen is shared with search.hip: the certified top-k screen of the re-rank uses the same copy)
namespace ah {
bool ensure_screen(ah_dataset *ds, hipStream_t s, bool want8, bool retry_failed) {
    std::lock_guard<std::mutex> lk(ds->mu);
    if (metric_is_bq(ds->metric) || ds->dims < 32 || ds->n == 0) {
        ds->screen_never = true;
        return false;
    }
    if (!ds->d_rows_h16) {
        // a reader that found no memory for the copies does not ask again on every call (each attempt would cost an
        // allocation of n x hpitch x 2 bytes): the next build does
        if (ds->screen_alloc_failed && !retry_failed) return false;
        ds->screen_alloc_failed = true;  // until the copies exist
        const uint32_t hpitch = (ds->dims + 63u) & ~63u;
        uint16_t *rows = nullptr;
        float4 *stats = nullptr;
        // (optional allocations: a full device does not make them empty the cache)
        if (dev_malloc((void **)&rows, ds->n * (size_t)hpitch * 2, true) != hipSuccess ||
            dev_malloc((void **)&stats, ds->n * sizeof(float4), true) != hipSuccess) {
            (void)hipGetLastError();
            if (rows) (void)dev_free_unused(rows);
            return false;
        }
        const DataView dv = ds->view();
        const unsigned grid = (unsigned)std::min<uint64_t>((ds->n + 31) / 32, 1u << 20);
        hipLaunchKernelGGL(k_shadow_rows, dim3(grid), dim3(kBlock), 0, s, dv, rows, hpitch, stats);
        uint32_t *d_max = nullptr;
        uint32_t h_max[4] = {0u, 0u, 0u, 0u};
        bool ok = dev_malloc((void **)&d_max, 16) == hipSuccess && hipMemsetAsync(d_max, 0, 16, s) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_stats_max, dim3(1024), dim3(256), 0, s, stats, (uint64_t)ds->n, d_max);
            ok = hipMemcpyAsync(h_max, d_max, 16, hipMemcpyDeviceToHost, s) == hipSuccess;
        }
        if (hipStreamSynchronize(s) != hipSuccess || !ok) {
            (void)hipGetLastError();
            (void)dev_free(rows);
            (void)dev_free(stats);
            if (d_max) (void)dev_free(d_max);
            return false;
        }
        (void)dev_free(d_max);
        memcpy(ds->screen_max, h_max, 12);
        ds->hpitch = hpitch;
        ds->d_screen_stats = stats;
        ds->d_rows_h16 = rows;
        ds->screen_alloc_failed = false;
        ds->screen_ready.store(true, std::memory_order_release);  // readers without `mu` look at this flag only
    }
    const long long tun8 = tun(TUN_SCREEN8);
    if (want8 && tun8 != 0 && ds->metric != AH_DOT_PRODUCT && !ds->d_rows_i8 && (!ds->screen8_decided || tun8 == 1))
        (void)ensure_screen8(ds, s, tun8 == 1);
    return true;
}
// The int8 copy for the SEARCH side (round 6: the first stage of the certified top-k screen of ah_search_batch /
// ah_rerank_batch): the copy the node-major levels of the build use — made here for DotProduct as well, whose build has no use
// for it (and therefore without the second digit of the rows, which only the build's stage 1 reads).
bool ensure_screen8_search(ah_dataset *ds, hipStream_t s) {
    if (ds->screen8_ready.load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lk(ds->mu);
    if ((ds->metric != AH_COSINE && ds->metric != AH_DOT_PRODUCT) || ds->dims < 32 || ds->n == 0) return false;
    if (ds->d_rows_i8) return ds->screen8_ready.load(std::memory_order_acquire);
    const long long tun8 = tun(TUN_SCREEN8);
    if (tun8 == 0 || (ds->screen8_decided && tun8 != 1)) return false;  // found useless (or unavailable) before
    return ensure_screen8(ds, s, tun8 == 1, ds->metric != AH_DOT_PRODUCT);
}
}  // namespace ah
// The int8 copy (rows, one scale per row, one power of two per dimension).  Kept only when it will decide most pairs: the
// margin of a row against a normal of an unrelated direction is ~ |n||x| / sqrt(dims), the bound ~ |n| |x - x~8| (the
// normal's two int8 digits make its own error negligible), so the copy is useful while quality = max|y/s - q| sqrt(dims)
// / (typical |x|/s) is small: 0.06 for uniform 768-d rows (~90 % of the pairs decided), 0.2 for N(0,1) rows (~84 %); a
// few huge entries in otherwise small rows blow it up.  `force` (AH_SCREEN8=1, a test aid) keeps it whatever the data.
static bool ensure_screen8(ah_dataset *ds, hipStream_t s, bool force, bool want_lo) {
    const DataView dv = ds->view();
    const uint32_t pitch8 = (ds->dims + 127u) & ~127u;
    const unsigned grid = (unsigned)std::min<uint64_t>((ds->n + 31) / 32, 1u << 20);
    int8_t *rows8 = nullptr, *rows8_lo = nullptr;
    float *scales = nullptr, *dimsc = nullptr;
    uint32_t *d_m = nullptr;  // [pitch8 column maxima][5 row maxima + pad]
    uint32_t h_m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // the rows' second digit is optional: without the memory for it stage 1 is simply the binary16 row
    if (want_lo && tun(TUN_SCREEN8_LO) != 0 && dev_malloc((void **)&rows8_lo, ds->n * (size_t)pitch8) != hipSuccess) {
        (void)hipGetLastError();
        rows8_lo = nullptr;
    }
    bool ok = dev_malloc((void **)&rows8, ds->n * (size_t)pitch8) == hipSuccess &&
              dev_malloc((void **)&scales, ds->n * sizeof(float)) == hipSuccess &&
              dev_malloc((void **)&dimsc, 2 * (size_t)pitch8 * sizeof(float)) == hipSuccess &&
              dev_malloc((void **)&d_m, ((size_t)pitch8 + 8) * 4) == hipSuccess;
    const bool alloc_ok = ok;
    ok = ok && hipMemsetAsync(d_m, 0, ((size_t)pitch8 + 8) * 4, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_col_maxabs, dim3((unsigned)std::min<uint64_t>(4096, (ds->n + 255) / 256)), dim3(256), 0, s, dv, d_m);
        hipLaunchKernelGGL(k_dim_scales, dim3((pitch8 + 255) / 256), dim3(256), 0, s, d_m, ds->dims, pitch8, dimsc, dimsc + pitch8);
        hipLaunchKernelGGL(k_shadow_rows8, dim3(grid), dim3(kBlock), 0, s, dv, dimsc + pitch8, rows8, rows8_lo, pitch8, scales,
                           d_m + pitch8);
        ok = hipMemcpyAsync(h_m, d_m + pitch8, 20, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    bool keep = false;
    if (ok) {
        float a8, b8, c8;
        memcpy(&a8, &h_m[0], 4);
        memcpy(&b8, &h_m[1], 4);
        memcpy(&c8, &h_m[2], 4);
        // against a TYPICAL |q| rather than the largest: rows are ~ isotropic after the column scaling, |q| ~ |x| / s
        const double quality = (double)b8 * std::sqrt((double)ds->dims) / std::max((double)a8, 1e-300);
        ds->screen8_quality = quality;
        if (tun(TUN_TIMING) != 0)
            fprintf(stderr, "[ah] int8 copy of %llu x %u rows: max |q| %.1f, max |y/s - q| %.2f, max |x|/s %.1f, quality %.3f (kept below 0.6)\n",
                    (unsigned long long)ds->n, ds->dims, (double)a8, (double)b8, (double)c8, quality);
        keep = std::isfinite(a8) && std::isfinite(b8) && std::isfinite(c8) && a8 > 0.0f && (force || quality < 0.6);
        if (keep) {
            ds->d_rows_i8 = rows8;
            ds->d_rows_i8_lo = rows8_lo;
            memcpy(&ds->screen8_max[3], &h_m[3], 8);
            ds->d_scale8_rows = scales;
            ds->d_dim_scale = dimsc;
            ds->pitch8 = pitch8;
            ds->screen8_max[0] = a8;
            ds->screen8_max[1] = b8;
            ds->screen8_max[2] = c8;
            ds->screen8_ready.store(true, std::memory_order_release);  // (readers without `mu`: the search paths)
        }
        ds->screen8_decided = true;
    } else {
        (void)hipGetLastError();
        if (alloc_ok) ds->screen8_decided = true;  // a device fault, not a lack of memory: do not loop on it
    }
    if (!keep) {
        if (rows8_lo) (void)dev_free(rows8_lo);
        if (rows8) (void)dev_free(rows8);
        if (scales) (void)dev_free(scales);
        if (dimsc) (void)dev_free(dimsc);
    }
    if (d_m) (void)dev_free(d_m);
    return keep;
}

// Device -> pageable host copies off the build's critical path.  hipMemcpy into pageable memory is staged by the
// runtime on one thread and pays the first-touch page faults of the fresh destination there (measured: 4 GB of item
// ids in 0.60 s = 6.7 GB/s, whatever the number of concurrent calls).  Instead a worker thread owns a pinned double
// buffer: the DMA engine fills one half while a few threads copy the other half to its final place, and the level
// loop never waits for it — each level's normals travel while the next levels are computed.
// One delivery of a streaming build (ah_build_forest_stream): the payloads of `nodes` lie back to back at a device address, in
// the order of the list (split planes of a level: one record per node; item ids: the leaves in (tree, position) order).  The
// read-back worker moves them through its pinned double buffer in pieces of whole nodes and calls the sink on every piece.
// (a list of 819 000 nodes is 39 MB: value-initialising it on the launching thread — page faults of fresh memory included — was
// most of the 37 ms a streaming build's deepest digest took; the threads that fill it touch its pages instead)
struct StreamNodeRaw {
    ah_stream_node v;
    StreamNodeRaw() {}  // NOLINT: no initialisation on purpose
};
static_assert(sizeof(StreamNodeRaw) == sizeof(ah_stream_node), "a plain wrapper");
struct StreamNodeList {
    std::vector<StreamNodeRaw> raw;
    void resize(size_t n) { raw.resize(n); }
    size_t size() const { return raw.size(); }
    bool empty() const { return raw.empty(); }
    ah_stream_node *data() { return reinterpret_cast<ah_stream_node *>(raw.data()); }
    ah_stream_node &operator[](size_t i) { return raw[i].v; }
    const ah_stream_node &operator[](size_t i) const { return raw[i].v; }
};
struct StreamJob {
    ah_node_batch head{};                // kind, level, record geometry
    StreamNodeList nodes;                // payload_offset: byte offset from the job's device address, ascending, contiguous
    uint64_t fixed_len = 0;              // bytes of one payload (split planes), or 0: count * 4 (item ids)
};
struct StreamTarget {
    ah_node_batch_fn sink = nullptr;
    void *user = nullptr;
    std::atomic<int> sink_rc{0};         // first non-zero return of the sink: the build stops with AH_ERR_CANCELLED
    std::atomic<int> too_big{0};         // a single payload larger than half of the pinned buffer
    uint64_t batches = 0, bytes = 0;
};

struct Readback {
    struct Job {
        void *dst;
        const void *src;
        size_t bytes;
        StreamJob *stream = nullptr;     // non-null: deliver to the sink (dst unused); owned by the job
    };
    StreamTarget *target = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    size_t pending = 0;
    bool stop = false, started = false, inline_mode = false;
    hipError_t err = hipSuccess;
    int device = 0;
    int numa_node = -1;      // host node of the device: the worker and its copy threads run there (api.hip "NUMA placement")
    uint8_t *pin = nullptr;  // 2 x half bytes of pinned memory (owned by the build's Context)
    size_t half = 0;

    void start(int dev, void *pinned, size_t pinned_bytes) {
        device = dev;
        pin = reinterpret_cast<uint8_t *>(pinned);
        half = pinned_bytes / 2;
        started = true;
        try {
            th = std::thread([this] { run(); });
        } catch (...) {  // no worker: push() copies synchronously
            started = false;
            inline_mode = true;
        }
    }
    // pinned bounce buffer -> final place, with non-temporal stores: the destination is written once and read much
    // later by the caller, so its lines need neither be read for ownership nor stay in the host caches
    __attribute__((target("avx2"))) static void copy_stream(uint8_t *dst, const uint8_t *src, size_t bytes) {
        size_t i = 0;
        const size_t head = std::min<size_t>(bytes, (32 - (reinterpret_cast<uintptr_t>(dst) & 31u)) & 31u);
        if (head) memcpy(dst, src, head);
        for (i = head; i + 128 <= bytes; i += 128) {
            const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
            const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
            const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
            const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
            _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
            _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
            _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
            _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
        }
        if (i < bytes) memcpy(dst + i, src + i, bytes - i);
        _mm_sfence();
    }
    static void copy_part(uint8_t *dst, const uint8_t *src, size_t bytes) {
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2 && bytes >= 4096) copy_stream(dst, src, bytes);
        else memcpy(dst, src, bytes);
    }
    unsigned copy_threads = 8;
    void spread(uint8_t *dst, const uint8_t *src, size_t bytes) const {  // copy on up to copy_threads cores
        const unsigned n_threads = (unsigned)std::min<size_t>(copy_threads, bytes >> 20);
        if (n_threads <= 1) {
            copy_part(dst, src, bytes);
            return;
        }
        const size_t per = ((bytes + n_threads - 1) / n_threads + 4095) & ~(size_t)4095;
        parallel_run(n_threads, [=](unsigned t) {
            const size_t lo = (size_t)t * per;
            if (lo < bytes) copy_part(dst + lo, src + lo, std::min(per, bytes - lo));
        });
    }
    hipError_t copy(const Job &job, hipStream_t cs, hipEvent_t *ev) {
        const bool direct = tun(TUN_READBACK_DIRECT) != 0;  // A/B: let the runtime stage the copy
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
    // device -> pinned half -> sink, piece by piece: the DMA of piece p + 1 is in flight while the sink looks at piece p
    hipError_t deliver(StreamJob &job, const uint8_t *src, hipStream_t cs, hipEvent_t *ev) {
        struct Piece {
            size_t a = 0, b = 0;      // nodes [a, b)
            uint64_t off = 0, len = 0;
            int buf = 0;
            bool live = false;
        } cur, prev;
        auto len_of = [&](const ah_stream_node &nd) -> uint64_t { return job.fixed_len ? job.fixed_len : (uint64_t)nd.count * 4; };
        const size_t n = job.nodes.size();
        size_t next = 0;
        int b = 0;
        hipError_t e = hipSuccess;
        while ((next < n || prev.live) && e == hipSuccess) {
            cur = Piece{};
            if (next < n && !target->sink_rc.load() && !target->too_big.load()) {
                cur.a = next;
                cur.off = job.nodes[next].payload_offset;
                size_t k = next;
                while (k < n && job.nodes[k].payload_offset + len_of(job.nodes[k]) - cur.off <= half) k++;
                if (k == next) {  // one payload alone does not fit: nothing sensible to hand over
                    target->too_big.store(1);
                    next = n;
                } else {
                    cur.b = k;
                    cur.len = job.nodes[k - 1].payload_offset + len_of(job.nodes[k - 1]) - cur.off;
                    cur.buf = b;
                    cur.live = true;
                    if (cur.len) e = hipMemcpyAsync(pin + (size_t)b * half, src + cur.off, cur.len, hipMemcpyDeviceToHost, cs);
                    if (e == hipSuccess) e = hipEventRecord(ev[b], cs);
                    next = k;
                    b ^= 1;
                }
            } else {
                next = n;
            }
            if (prev.live && e == hipSuccess) {
                e = hipEventSynchronize(ev[prev.buf]);
                if (e == hipSuccess && !target->sink_rc.load()) {
                    for (size_t i = prev.a; i < prev.b; i++) job.nodes[i].payload_offset -= prev.off;
                    ah_node_batch batch = job.head;
                    batch.n_nodes = prev.b - prev.a;
                    batch.nodes = job.nodes.data() + prev.a;
                    batch.payload = pin + (size_t)prev.buf * half;
                    batch.payload_len = prev.len;
                    const int rc = target->sink(target->user, &batch);
                    target->batches++;
                    target->bytes += prev.len;
                    if (rc != 0) target->sink_rc.store(rc);
                }
            }
            prev = cur;
        }
        return e;
    }
    void run() {
        hipStream_t cs = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        (void)numa_bind_thread_to_node(numa_node);  // (threads this one starts — the copies' spread — inherit its CPUs)
        hipError_t e = hipSetDevice(device);
        // A stream of its own PRIORITY: the runtime multiplexes a process's streams onto a few hardware queues per priority
        // (four by default), and once more than four streams are alive — a second dataset with its context is enough — this
        // stream could land on the queue of the build's compute stream: every chunk of the read-back then waited for the
        // level's kernel in front of it, and the ids of a group of trees arrived 0.1 s late (10M x 100 trees: 1.43 instead of
        // 1.31 s for every build of the second dataset; scripts/exp_second_dataset.py).  High-priority streams have queues
        // of their own.
        if (e == hipSuccess) e = create_copy_stream(&cs);
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
            if (e == hipSuccess) e = job.stream ? deliver(*job.stream, reinterpret_cast<const uint8_t *>(job.src), cs, ev) : copy(job, cs, ev);
            delete job.stream;
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
        if (inline_mode) {  // the worker thread could not be created: a plain synchronous copy
            const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess && err == hipSuccess) err = e;
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(Job{dst, src, bytes, nullptr});
            pending++;
        }
        cv.notify_all();
    }
    // hand a stream job (ownership included) to the worker; without a worker thread it is delivered right here
    void push_stream(StreamJob *job, const void *src) {
        if (job->nodes.empty()) {
            delete job;
            return;
        }
        if (inline_mode) {
            hipStream_t cs = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            hipError_t e = create_copy_stream(&cs);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            if (e == hipSuccess) e = deliver(*job, reinterpret_cast<const uint8_t *>(src), cs, ev);
            if (e != hipSuccess && err == hipSuccess) err = e;
            if (ev[0]) (void)hipEventDestroy(ev[0]);
            if (ev[1]) (void)hipEventDestroy(ev[1]);
            if (cs) (void)hipStreamDestroy(cs);
            delete job;
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(Job{nullptr, src, 1, job});
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

// Launch one instantiation of k_forest_screen_rows (metric x trees per group x LDS-resident normals).
namespace {
struct ScreenRowsArgs {
    DataView dv;
    ScreenView sv;
    const uint32_t *node_of;
    uint32_t tree_base, n_pass;
    const uint8_t *normals;
    uint64_t nstride, hdr_off;
    const uint8_t *shadow;
    uint64_t hstride;
    uint8_t *side_bytes;
    RowsSchedule sch;
    AbortFlags abort_flag;
    ScreenCounters *counters;
    uint32_t verify;
};
template <int M, int TC, bool LDS>
int launch_screen_rows_inst(const ScreenRowsArgs &a, unsigned grid, size_t sh, hipStream_t s, int device) {
    const unsigned threads = LDS ? (TC >= 16 ? 512u : 1024u) : (unsigned)kBlock;
    if (LDS) {
        static std::atomic<bool> opt_in[64];  // once per instantiation and device
        if (!opt_in[device & 63].load(std::memory_order_acquire)) {
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_screen_rows<M, TC, LDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsNormalsBytes));
            opt_in[device & 63].store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((k_forest_screen_rows<M, TC, LDS>), dim3(grid), dim3(threads), LDS ? sh : 0, s, a.dv, a.sv, a.node_of,
                       a.tree_base, a.n_pass, a.normals, a.nstride, a.hdr_off, a.shadow, a.hstride, a.side_bytes, a.sch,
                       a.abort_flag, a.counters, a.verify);
    return AH_OK;
}
template <int M>
int launch_screen_rows_metric(uint32_t tc, bool lds, const ScreenRowsArgs &a, unsigned grid, size_t sh, hipStream_t s, int device) {
    if (lds) return tc >= 16 ? launch_screen_rows_inst<M, 16, true>(a, grid, sh, s, device)
                             : launch_screen_rows_inst<M, 8, true>(a, grid, sh, s, device);
    switch (tc) {
    case 16: return launch_screen_rows_inst<M, 16, false>(a, grid, sh, s, device);
    case 8: return launch_screen_rows_inst<M, 8, false>(a, grid, sh, s, device);
    case 4: return launch_screen_rows_inst<M, 4, false>(a, grid, sh, s, device);
    default: return launch_screen_rows_inst<M, 2, false>(a, grid, sh, s, device);
    }
}
int launch_screen_rows(int metric, uint32_t tc, bool lds, const ScreenRowsArgs &a, unsigned grid, size_t sh, hipStream_t s,
                       int device) {
    switch (metric) {
    case AH_EUCLIDEAN: return launch_screen_rows_metric<AH_EUCLIDEAN>(tc, lds, a, grid, sh, s, device);
    case AH_MANHATTAN: return launch_screen_rows_metric<AH_MANHATTAN>(tc, lds, a, grid, sh, s, device);
    case AH_COSINE: return launch_screen_rows_metric<AH_COSINE>(tc, lds, a, grid, sh, s, device);
    default: return launch_screen_rows_metric<AH_DOT_PRODUCT>(tc, lds, a, grid, sh, s, device);
    }
}
}  // namespace

// Grids of the dense MFMA screen and of the exact-pairs pass behind it (also used by ah_debug_launch_coverage).
namespace {
struct DensePlan {
    uint32_t n_row_tiles, n_col_tiles, group;
    bool wide;    // k_forest_dense_screen: 256-column tiles (512 threads) unless one 128-column tile covers the level
    int narrow;   // 0: k_forest_dense_screen (256-row tiles); 2 / 4: k_forest_dense_narrow<NT> (160-row tiles, 32 NT columns)
    bool stream;  // narrow: rows read once per level (one column tile): non-temporal loads
    uint32_t tile_rows, tile_cols, threads, lds;
    uint64_t grid;
};
DensePlan dense_plan(uint64_t N, uint32_t n_cols) {
    DensePlan p{};
    // the narrow kernel while the level is bound by the rows' HBM time rather than by the matrix units
    const long long narrow_max = tun(TUN_DENSE_NARROW_MAX_COLS), narrow_force = tun(TUN_DENSE_NARROW);
    p.narrow = narrow_force == 0 || (long long)n_cols > narrow_max ? 0 : (n_cols <= 64 ? 2 : 4);
    if (p.narrow) {
        p.tile_rows = kNarrowRows;
        p.tile_cols = 32u * (uint32_t)p.narrow;
        p.threads = kNarrowThreads;
        p.lds = p.narrow == 2 ? NarrowShape<2>::kLds : NarrowShape<4>::kLds;
        p.wide = false;
    } else {
        p.wide = n_cols > 128;
        p.tile_rows = kDM;
        p.tile_cols = p.wide ? 256u : 128u;
        p.threads = p.wide ? DenseShape<4>::kThreads : DenseShape<2>::kThreads;
        p.lds = kDenseLds;
    }
    p.n_row_tiles = (uint32_t)((N + p.tile_rows - 1) / p.tile_rows);
    p.n_col_tiles = (n_cols + p.tile_cols - 1) / p.tile_cols;
    p.group = p.n_col_tiles > 1 ? (p.narrow ? 12u : kDenseGroup) : 1u;  // ~3 MB of X~ tiles per group and XCD
    const long long st = tun(TUN_DENSE_NARROW_STREAM);
    p.stream = p.narrow && (st < 0 ? p.n_col_tiles == 1 : st != 0);
    const uint64_t r8 = (p.n_row_tiles + 7) / 8;
    p.grid = 8 * ((r8 + p.group - 1) / p.group) * p.group * p.n_col_tiles;
    return p;
}
unsigned exact_pairs_grid(uint64_t N, uint32_t n_trees) {  // a multiple of 8: block b and b + grid land on the same XCD
    return (unsigned)std::min<uint64_t>((((N + 1023) / 1024 + 7) / 8) * 8 * ((n_trees + 3) / 4), 1u << 16);
}
}  // namespace

// Launch plan of one screened row-major pass over `groups` groups of `tcv` trees (RowsSchedule): chunk size, one XCD per
// (chunk, group) or not, non-temporal rows or not, and the launches the pass is cut into (a launch carries at most
// LAUNCH_MAX_ITEMS work-items).  The build and ah_debug_launch_coverage both launch exactly what this returns.
namespace {
struct RowsPlan {
    RowsSchedule sch{};
    unsigned threads = kBlock;
    bool xcd = false, nt = false;
    std::vector<std::pair<uint32_t, unsigned>> launches;  // (first chunk, grid)
};
int plan_rows_launches(uint64_t N, uint32_t hpitch, uint32_t tcv, uint32_t groups, bool lds, uint64_t group_normal_bytes,
                       RowsPlan *plan) {
    const uint32_t env_rpb = (uint32_t)std::max<long long>(0, tun(TUN_ROWS_PER_BLOCK)) / 32u * 32u;
    const uint32_t rpb = lds ? 1024u : env_rpb ? env_rpb : 32u;  // rows per block
    const uint64_t hrow = (uint64_t)hpitch * 2;
    const uint64_t want_rows = tun(TUN_ROWS_CHUNK_ROWS) > 0 ? (uint64_t)tun(TUN_ROWS_CHUNK_ROWS)
                                                             : ((uint64_t)std::max<long long>(1, tun(TUN_ROWS_CHUNK_MB)) << 20) / hrow;
    uint32_t chunk_rows = (uint32_t)std::max<uint64_t>(rpb, want_rows / rpb * rpb);
    if (chunk_rows > N) chunk_rows = (uint32_t)((N + rpb - 1) / rpb * rpb);
    const uint32_t n_chunks = (uint32_t)((N + chunk_rows - 1) / chunk_rows);
    RowsSchedule &sch = plan->sch;
    sch.n_groups = groups;
    sch.tiles = chunk_rows / rpb;
    sch.chunk_rows = chunk_rows;
    sch.rows_per_block = rpb;
    // one XCD per (chunk, group) pays for small groups of many trees (measured at 10M x 100 trees: TC=4 level 9 139 -> 124 ms,
    // level 10 199 -> 188; TC=2 level 10 211 -> 193; TC=8 with its 12 groups on 8 XCDs: level 8 111 -> 120, so not there)
    plan->xcd = !lds && tun(TUN_ROWS_XCD) != 0 && tcv <= 4 && groups >= (uint32_t)std::max<long long>(1, tun(TUN_ROWS_XCD_MIN_GROUPS));
    sch.xcd_slots = plan->xcd ? (groups + 7) / 8 : 0u;
    const uint64_t per_chunk = plan->xcd ? 8ull * sch.xcd_slots * sch.tiles : (uint64_t)groups * sch.tiles;
    // One XCD per group: the rows of a (chunk, group) are read once by that XCD, so in its L2 they only compete with the
    // group's normals.  When those no longer fit (> ~5 MB) the rows are streamed with non-temporal loads (level 10 at 10M x
    // 100 trees: 188 -> 171 ms); while they fit, cached row loads are better (the chunk then stays in the Infinity Cache
    // for the other groups: level 9 125 vs 142 ms).
    const long long nt_force = tun(TUN_ROWS_NT);
    plan->nt = plan->xcd && (nt_force >= 0 ? nt_force != 0 : group_normal_bytes > (uint64_t)std::max<long long>(0, tun(TUN_ROWS_NT_BYTES)));
    if (plan->nt) sch.xcd_slots |= 0x80000000u;
    plan->threads = lds ? (tcv >= 16 ? 512u : 1024u) : (unsigned)kBlock;
    // a launch carries at most 2^32 - 1 work-items: big levels go out in several launches over chunk ranges
    const uint64_t max_items = (uint64_t)std::min<long long>(0xFFFFFFFFll, std::max<long long>(1, tun(TUN_LAUNCH_MAX_ITEMS)));
    AH_REQUIRE(per_chunk * plan->threads <= max_items, AH_ERR_INVALID_ARGUMENT, "forest build: too many trees for one row-major launch");
    const uint32_t chunks_per_launch = (uint32_t)std::min<uint64_t>(n_chunks, (max_items / plan->threads) / per_chunk);
    plan->launches.clear();
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += chunks_per_launch)
        plan->launches.emplace_back(c0, (unsigned)((uint64_t)std::min<uint32_t>(chunks_per_launch, n_chunks - c0) * per_chunk));
    return AH_OK;
}
}  // namespace

// `subset_ids` == nullptr: every tree covers all items (Writer::build with missing trees).  Otherwise tree t covers
// the ascending id list subset_ids[subset_offsets[first_tree + t] .. subset_offsets[first_tree + t + 1]) — the
// "descendants that became too large" of an incremental build (src/writer.rs:660-739).
//
// One level = one set of launches, and ONE host wait: the host needs the next level's sizes (LevelInfo, read back
// through pinned memory) before it can launch it; everything else about a level — its node table, from which the final
// node list is assembled — follows on a side stream and is digested by the host while the GPU runs the next level.
// host threads one build may keep busy at a time for its output path (page commits, bounce copies, digests, node list):
// ah_build_options.max_host_threads, else AH_HOST_THREADS (8).  Eight concurrent builds of an 8-GPU node in a 16-CPU
// container want 2 each.
static unsigned host_thread_budget(const ah_build_options *opt) {
    const long long v = opt->max_host_threads ? (long long)opt->max_host_threads : tun(TUN_HOST_THREADS);
    return (unsigned)std::min<long long>(64, std::max<long long>(1, v));
}

// State of one ah_build_forest_stream call across its batches.
struct StreamBuild {
    StreamTarget target;
    uint32_t id_base = 0;       // node ids handed out by earlier batches
    uint32_t *roots = nullptr;  // the caller's out_roots
};
struct LeafRec {  // a Descendants node of a streaming build: its ids are final_perm[start, start + count)
    uint64_t start;
    uint32_t count, id, tree, depth;
};

static int build_batch(ah_dataset *ds, const ah_build_options *opt, uint32_t first_tree, uint32_t n_trees,
                       uint32_t split_after, ah_forest *forest, Context *ctx, const uint32_t *subset_ids,
                       const uint64_t *subset_offsets, StreamBuild *sb = nullptr) {
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
    const uint64_t nstride = normal_record_stride(ds);
    // the tunables (common.h), read once per batch
    const bool g_screen = tun(TUN_SCREEN) != 0, g_screen_verify = tun(TUN_SCREEN_VERIFY) != 0;
    const int g_rows_force = (int)tun(TUN_ROWMAJOR), g_dense = (int)tun(TUN_DENSE);
    const uint32_t g_tile_blocks = (uint32_t)std::max<long long>(1, tun(TUN_FOREST_TILE_BLOCKS));
    const uint32_t g_node_blocks = (uint32_t)std::max<long long>(0, tun(TUN_FOREST_NODE_BLOCKS));
    const uint32_t g_split_blocks = (uint32_t)std::max<long long>(1, tun(TUN_FOREST_SPLIT_BLOCKS));
    const uint32_t g_row_blocks = (uint32_t)std::max<long long>(1, tun(TUN_FOREST_ROW_BLOCKS));
    const bool g_rows_advance = tun(TUN_ROWMAJOR_ADVANCE) != 0, g_rows_lds = tun(TUN_ROWMAJOR_LDS) != 0;
    const uint32_t g_rows_max_tc = (uint32_t)std::max<long long>(2, tun(TUN_ROWMAJOR_MAX_TC));
    const double g_rows_cache_mb = (double)tun(TUN_ROWMAJOR_CACHE_MB);
    const uint32_t g_dense_max_cols = (uint32_t)std::max<long long>(0, tun(TUN_DENSE_MAX_COLS));
    const double g_dense_gmacs = (double)std::max<long long>(1, tun(TUN_DENSE_GMACS));
    const int timing = (int)tun(TUN_TIMING);
    const bool g_retry_gate = tun(TUN_RETRY_GATE) != 0;
    // AH_MARGIN_MODE (measurement aid): the kernel family for callers that leave the choice to the library
    const uint32_t env_mode = (uint32_t)tun(TUN_MARGIN_MODE) & 0xFFFu;
    const uint32_t mode_req = (opt->margin_mode & 0xFFFu) ? (opt->margin_mode & 0xFFFu) : env_mode;
    const bool exact_only = (opt->margin_mode & AH_MARGIN_EXACT_ONLY) != 0 || !g_screen;

    // Upper bounds known up front (every split node owns > split_after items), so nothing is reallocated
    // between levels: nodes per level <= n_trees * N / (split_after + 1), tiles <= items / kTile + nodes.
    const uint64_t max_nodes = M / ((uint64_t)split_after + 1) + n_trees;
    const uint64_t max_tiles = M / kTile + n_trees + max_nodes;
    DevBuf<uint32_t> perm_a, perm_b, final_perm, tile_left, tile_left_off, d_child, d_small;
    DevBuf<FNode> d_nodes_a, d_nodes_b, d_nodes_c;  // (c: the tail in groups of trees)
    DevBuf<GroupInfo> d_groups;
    DevBuf<FTile> d_tiles;
    DevBuf<uint64_t> masks;
    DevBuf<NextCounts> d_block_sums;
    AH_TRY(perm_a.ensure(M));
    AH_TRY(perm_b.ensure(M));
    AH_TRY(final_perm.ensure(M));
    AH_TRY(d_nodes_a.ensure(max_nodes));
    AH_TRY(d_nodes_b.ensure(max_nodes));
    AH_TRY(d_tiles.ensure(max_tiles));
    AH_TRY(masks.ensure(max_tiles * 32));
    AH_TRY(tile_left.ensure(max_tiles));
    AH_TRY(tile_left_off.ensure(max_tiles));
    AH_TRY(d_child.ensure(2 * max_nodes));
    AH_TRY(d_block_sums.ensure(max_nodes / 256 + 2));
    // small device block: [abort flag, 3 pad][ScreenCounters][LevelInfo + tree_first[n_trees + 1]]
    const size_t info_words = (sizeof(LevelInfo) + ((size_t)n_trees + 1) * 4 + 3) / 4;
    AH_TRY(d_small.ensure(4 + 12 + info_words));
    const AbortFlags d_abort{d_small.p, nullptr};
    ScreenCounters *d_counters = reinterpret_cast<ScreenCounters *>(d_small.p + 4);
    LevelInfo *d_info = reinterpret_cast<LevelInfo *>(d_small.p + 16);
    uint32_t *d_tree_first = reinterpret_cast<uint32_t *>(d_info + 1);
    DevBuf<uint32_t> d_tree_first_buf;  // first node of every tree of the level, gaps closed (LDS variant of the row pass)
    AH_TRY(d_tree_first_buf.ensure((size_t)n_trees + 2));
    // The control structures start from zeros.  Blocks come from the caching allocator with whatever their last user left in
    // them, and a CANCELLED level lets its bookkeeping kernels run over tables its drained margin kernels never wrote: with
    // zeros (what fresh device memory used to hold) those kernels find empty nodes and tiles; with the item indices of
    // another build in a tile table they would scatter out of bounds.  ~0.7 GB of memsets: 0.2 ms.
    AH_HIP(hipMemsetAsync(d_nodes_a.p, 0, max_nodes * sizeof(FNode), s));
    AH_HIP(hipMemsetAsync(d_nodes_b.p, 0, max_nodes * sizeof(FNode), s));
    AH_HIP(hipMemsetAsync(d_tiles.p, 0, max_tiles * sizeof(FTile), s));
    AH_HIP(hipMemsetAsync(masks.p, 0, max_tiles * 32 * sizeof(uint64_t), s));
    AH_HIP(hipMemsetAsync(tile_left.p, 0, max_tiles * 4, s));
    AH_HIP(hipMemsetAsync(tile_left_off.p, 0, max_tiles * 4, s));
    AH_HIP(hipMemsetAsync(d_child.p, 0, 2 * max_nodes * 4, s));
    AH_HIP(hipMemsetAsync(d_block_sums.p, 0, (max_nodes / 256 + 2) * sizeof(NextCounts), s));
    AH_HIP(hipMemsetAsync(d_tree_first_buf.p, 0, ((size_t)n_trees + 2) * 4, s));
    const auto t_setup_alloc = std::chrono::steady_clock::now();
    uint32_t *d_tree_first_fixed = d_tree_first_buf.p;
    AH_HIP(hipMemsetAsync(d_small.p, 0, (16 + info_words) * 4, s));

    // pinned host memory: [2 x LevelInfo block][one word for the abort flag][2 x node table][read-back bounce]
    // pinned double buffer of the read-back worker (a streaming build hands a sink at most one half at a time)
    const size_t kBounce = (size_t)std::min<long long>(4096, std::max<long long>(2, tun(TUN_READBACK_MB))) << 20;
    const size_t pin_info = (info_words * 4 + 255) & ~(size_t)255;
    const size_t pin_nodes = (max_nodes * sizeof(FNode) + 4095) & ~(size_t)4095;
    const size_t pin_head = (3 * pin_info + 256 + 4095) & ~(size_t)4095;
    AH_TRY(ctx->ensure_pinned(pin_head + 2 * pin_nodes + kBounce));
    const auto t_setup_pin = std::chrono::steady_clock::now();
    uint8_t *pin = reinterpret_cast<uint8_t *>(ctx->h_pinned);
    LevelInfo *h_info[2] = {reinterpret_cast<LevelInfo *>(pin), reinterpret_cast<LevelInfo *>(pin + pin_info)};
    uint32_t *h_one = reinterpret_cast<uint32_t *>(pin + 2 * pin_info);
    *h_one = 1u;
    uint32_t *h_tree_first = reinterpret_cast<uint32_t *>(pin + 2 * pin_info + 256);  // staging of d_tree_first_fixed
    FNode *h_nodes[2] = {reinterpret_cast<FNode *>(pin + pin_head), reinterpret_cast<FNode *>(pin + pin_head + pin_nodes)};

    // row-major margin mode (full-dataset trees, f32 metrics): node index and side byte per (tree, row)
    const bool rows_allowed = !subset_ids && !bq && ds->dims >= 32 && n_trees >= 2 &&
                              (mode_req != AH_MARGIN_AUTO ? mode_req != AH_MARGIN_NODE_MAJOR : g_rows_force != 0);
    DevBuf<uint32_t> node_of;
    DevBuf<uint8_t> side_bytes;
    DevBuf<uint32_t> side_bits;
    if (rows_allowed) {
        AH_TRY(node_of.ensure((size_t)n_trees * N + 4));
        // padded to whole 1 KiB windows and zeroed once: k_forest_exact_pairs scans it 16 bytes per lane for marks
        AH_TRY(side_bytes.ensure((size_t)n_trees * N + 1024 + 16));
        AH_HIP(hipMemsetAsync(side_bytes.p, 0, (size_t)n_trees * N + 1024 + 16, s));
        // the same sides one bit each for the permutation-order gather of k_forest_masks_from_bytes (worth its pass from
        // the size on at which a tree's bytes no longer sit in an L2)
        if ((tun(TUN_MASK_BITS) > 0 || (tun(TUN_MASK_BITS) < 0 && N >= (1u << 20))) &&
            side_bits.ensure(((size_t)n_trees * N + 31) / 32 + 4) != AH_OK)
            (void)hipGetLastError();  // no room: the tiles gather the bytes
    }
    // certified binary16 screen: f32 metrics with AVX-tier rows, unless the caller (or AH_SCREEN=0) asks for f32 only
    const auto t_setup_rows = std::chrono::steady_clock::now();
    const bool screen = !exact_only && !bq && ds->dims >= 32 && ensure_screen(ds, s, true, true);
    const auto t_setup_screen = std::chrono::steady_clock::now();
    if (!exact_only && !screen && ds->screen_alloc_failed) forest->stats.screen_unavailable = 1;
    ScreenView sv{};
    uint64_t hstride = 0;
    if (screen) {
        sv.rows = ds->d_rows_h16;
        sv.stats = ds->d_screen_stats;
        sv.max_stats = make_float4(ds->screen_max[0], ds->screen_max[1], ds->screen_max[2], 0.0f);
        sv.rows8 = tun(TUN_SCREEN8) != 0 ? ds->d_rows_i8 : nullptr;  // nullptr: no int8 first stage
        sv.pitch8 = ds->pitch8;
        sv.scale8_rows = ds->d_scale8_rows;
        sv.max8 = make_float4(ds->screen8_max[0], ds->screen8_max[1], ds->screen8_max[2], 0.0f);
        sv.rows8_lo = tun(TUN_SCREEN8_LO) != 0 ? ds->d_rows_i8_lo : nullptr;  // nullptr: stage 1 is the binary16 row
        sv.max8b = make_float4(ds->screen8_max[3], ds->screen8_max[4], ds->screen8_max[2], 0.0f);
        sv.hpitch = ds->hpitch;
        // accumulation-error factors (screen_device.h), each with a 4x safety factor over the standard model:
        //   screen: hpitch/16 dot2c per lane (2 roundings each) + 4 adds;  reference: dims/32 FMAs per chain, 6 adds of
        //   the hsum tree, and a scalar tail of up to 31 multiply-adds (2 roundings each)
        sv.gamma_s = (float)(4.0 * (2.0 * (ds->hpitch / 16) + 8.0) * 5.9604645e-8);
        sv.gamma_r = (float)(4.0 * ((double)(ds->dims / 32) + 6.0 + 62.0) * 5.9604645e-8);
        hstride = ((uint64_t)ds->hpitch * 2 + 16 + 127) & ~(uint64_t)127;  // whole lines, see normal_record_stride
    }
    const uint32_t verify = screen && g_screen_verify ? 1u : 0u;
    const bool screen8 = screen && sv.rows8 != nullptr;
    const uint64_t stride8 = screen8 ? ((2 * (uint64_t)sv.pitch8 + sizeof(NormalStats8) + 127) & ~(uint64_t)127) : 0;
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
    const int numa_node = numa_node_of_device(ds->device);
    struct BlobNodeScope {  // the blobs this thread asks the pool for prefer the device's host node
        int prev;
        explicit BlobNodeScope(int n) : prev(tl_blob_node) { tl_blob_node = n; }
        ~BlobNodeScope() { tl_blob_node = prev; }
    } blob_node_scope(numa_node);
    if (!sb) {  // (a streaming build hands the ids to the sink from the pinned ring: no blob)
        AH_REQUIRE(host_blob_reserve(forest->desc_blob, (desc_base + M) * 4, desc_base * 4), AH_ERR_OUT_OF_MEMORY,
                   "host allocation of the descendants failed");
        forest->descendants = reinterpret_cast<uint32_t *>(forest->desc_blob.p);
    }
    struct Toucher {  // commits the pages of a fresh (still unwritten) host range in the background
        std::thread th;
        unsigned max_threads = 8;
        int numa_node = -1;  // the pages are committed from the CPUs of the device's host node: first touch decides where they live
        // [off, off + bytes) of `blob`; what a recycled blob already has committed needs no touch
        void start(const HostBlob &blob, size_t off, size_t bytes) {
            join();
            const size_t lo_off = std::max(off, blob.committed), hi_off = std::min(off + bytes, blob.cap);
            if (hi_off <= lo_off || hi_off - lo_off < (8u << 20)) return;
            uint8_t *lo = blob.p + lo_off;
            const size_t len = hi_off - lo_off;
            const unsigned budget = max_threads;
            const int node = numa_node;
            try {
                th = std::thread([lo, len, budget, node] {
                    (void)numa_bind_thread_to_node(node);
                    // (several threads: the 2.6 GB of the deepest level's normals must be committed within that level's
                    // ~140 ms, or the level loop waits for page faults before it can hand the chunk to the read-back worker)
                    const unsigned parts = (unsigned)std::min<size_t>(budget, std::max<size_t>(1, len >> 26));
                    parallel_run(parts, [=](unsigned p) {
                        const size_t a = len * p / parts, b = len * (p + 1) / parts;
                        for (size_t o = a; o < b; o += 4096) reinterpret_cast<volatile uint8_t *>(lo)[o] = 0;
                    });
                });
            } catch (...) {  // no thread to be had: the read-back worker faults the pages in itself
            }
        }
        void join() {
            if (th.joinable()) th.join();
        }
        ~Toucher() { join(); }
    } prefault, touch_normals[2];  // [level & 1]: commits the level's normals, started one level ahead
    const unsigned host_threads = host_thread_budget(opt);
    prefault.max_threads = touch_normals[0].max_threads = touch_normals[1].max_threads = host_threads;
    prefault.numa_node = touch_normals[0].numa_node = touch_normals[1].numa_node = numa_node;
    if (!sb) prefault.start(forest->desc_blob, desc_base * 4, M * 4);
    Arena arena, shadow_arena, shadow8_arena;  // declared before the read-back worker: it is joined before the chunks it reads are freed
    arena.block_bytes = std::max<uint64_t>(32ull << 20, std::min<uint64_t>(2 * max_nodes * nstride, 16ull << 30));
    shadow_arena.block_bytes = std::max<uint64_t>(16ull << 20, std::min<uint64_t>(max_nodes * std::max<uint64_t>(hstride, 16), 8ull << 30));
    shadow8_arena.block_bytes = std::max<uint64_t>(16ull << 20, std::min<uint64_t>(max_nodes * std::max<uint64_t>(stride8, 16), 4ull << 30));
    BatchCleanup bc;
    AH_TRY(bc.create());
    Readback rb;
    rb.target = sb ? &sb->target : nullptr;
    rb.copy_threads = host_threads;
    rb.numa_node = numa_node;
    rb.start(ds->device, pin + pin_head + 2 * pin_nodes, kBounce);

    // ---- level 0 on the host: the roots -----------------------------------------------------------------------------
    std::vector<HostRec> recs;
    std::vector<uint32_t> new_index;  // (emit_range's: record -> forest-local node index)
    RecordsLease records_lease(recs, new_index);
    std::vector<uint32_t> tree_root(n_trees);
    std::vector<uint64_t> tree_count(n_trees, 1);  // nodes of every tree so far: its root, then two per digested split
    std::vector<uint32_t> level_rec, next_rec;  // HostRec index of every node of the level being digested / of the next
    // (room for every node of a balanced forest — leaves of split_after / 2 .. split_after items, as many split nodes — so that
    // the vector never moves: at 10M x 100 trees the 1.7 M + 1.7 M records outgrew the former 4 / 3 x max_nodes at the last big
    // level, and copying 136 MB of records held the launching thread for 38 ms with the device idle; untouched pages cost nothing)
    try {
        recs.reserve(4 * max_nodes + n_trees + 16);
    } catch (const std::bad_alloc &) {  // (a tiny split_after on a huge dataset: grow as needed, as before)
        recs.reserve(4 * max_nodes / 3 + 16);
    }
    size_t n_recs = 0;  // records in use; the vector itself is grown AHEAD of the digest (value-initialising 80 MB of fresh
                        // pages on the thread that digests the deepest level was most of that digest's 35 ms)
    LevelInfo info{};
    std::vector<uint32_t> tree_first(n_trees + 1, 0xFFFFFFFFu);
    // streaming build: the Descendants nodes, one list per level of creation (each ascending in (tree, position))
    std::vector<std::vector<LeafRec>> stream_leaves;
    const uint32_t id_base = sb ? sb->id_base : 0u;
    if (sb) stream_leaves.emplace_back();
    {
        FNode *lv = h_nodes[0];
        for (uint32_t t = 0; t < n_trees; t++) {
            const uint32_t cnt = (uint32_t)(tree_base[t + 1] - tree_base[t]);
            HostRec r{};
            r.tree = t;
            r.start = tree_base[t];
            r.count = cnt;
            r.depth = 0;
            tree_root[t] = (uint32_t)n_recs;
            if (cnt <= split_after) {  // fit_in_descendant at the root: the tree is one Descendants node
                r.kind = AH_NODE_DESCENDANTS;
                if (sb) stream_leaves[0].push_back(LeafRec{r.start, cnt, id_base + (uint32_t)n_recs, first_tree + t, 0u});
                recs.push_back(r);
                n_recs++;
                continue;
            }
            r.kind = AH_NODE_SPLIT;
            FNode nd{};
            nd.key = ah_node_key_root(opt->tree_seeds[first_tree + t]);
            nd.start = tree_base[t];
            nd.tree = t;
            nd.count = cnt;
            nd.tile_begin = info.n_tiles;
            nd.n_tiles = (cnt + kTile - 1) / kTile;
            tree_first[t] = info.n_nodes;
            level_rec.push_back((uint32_t)n_recs);
            recs.push_back(r);
            n_recs++;
            lv[info.n_nodes++] = nd;
            info.n_tiles += nd.n_tiles;
            info.pairs += cnt;
        }
        if (info.n_nodes) AH_HIP(hipMemcpyAsync(d_nodes_a.p, lv, info.n_nodes * sizeof(FNode), hipMemcpyHostToDevice, s));
    }

    uint32_t *cur = perm_a.p, *nxt = perm_b.p;
    FNode *d_cur = d_nodes_a.p, *d_next = d_nodes_b.p;
    AH_HIP(hipEventRecord(bc.ev_begin, s));
    forest->stats.seconds_setup += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_batch).count();
    if (timing) {
        auto ms_of = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[ah] batch setup: device buffers %.1f ms, pinned %.1f ms, row-major buffers %.1f ms, screen copies %.1f ms, "
                        "host blobs + workers + roots %.1f ms\n",
                ms_of(t_batch, t_setup_alloc), ms_of(t_setup_alloc, t_setup_pin), ms_of(t_setup_pin, t_setup_rows),
                ms_of(t_setup_rows, t_setup_screen), ms_of(t_setup_screen, std::chrono::steady_clock::now()));
    }
    const size_t cs_shared = (size_t)f32_space_pitch(ds->metric, ds->dims) * 4 * 3;
    AH_REQUIRE(cs_shared <= 150 * 1024, AH_ERR_INVALID_DIMENSION, "dimensions %u too large for the LDS-resident two-means",
               ds->dims);
    const int split_metric = f32_space_metric(ds->metric);
#define AH_SPLIT_KERNEL(DO)                                              \
    switch (split_metric) {                                              \
    case AH_EUCLIDEAN: DO(k_forest_create_split<AH_EUCLIDEAN>); break;   \
    case AH_MANHATTAN: DO(k_forest_create_split<AH_MANHATTAN>); break;   \
    case AH_COSINE: DO(k_forest_create_split<AH_COSINE>); break;         \
    default: DO(k_forest_create_split<AH_DOT_PRODUCT>); break;           \
    }
    if (cs_shared > 48 * 1024) {
#define AH_SPLIT_OPT_IN(K) AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs_shared))
        AH_SPLIT_KERNEL(AH_SPLIT_OPT_IN)
#undef AH_SPLIT_OPT_IN
    }
    const uint64_t normals_base = forest->normals_len;  // this batch appends its levels after earlier batches
    uint64_t normals_bytes = 0;
    // The host blob gets lazily committed head-room (splits in total ~1.3-1.6 x items / split_after) so that finished
    // levels can land in their final place while the build continues; it only moves when no copy is in flight.
    uint64_t normals_cap = forest->normals_blob.cap >= 16 ? forest->normals_blob.cap - 16 : 0;
    auto reserve_normals = [&](uint64_t need, uint64_t landed) -> int {  // landed: bytes of earlier levels / batches to keep
        if (need <= normals_cap) return AH_OK;
        touch_normals[0].join();  // the blob may move: nothing may be touching it
        touch_normals[1].join();
        AH_REQUIRE(rb.drain() == hipSuccess, AH_ERR_DEVICE, "device -> host copy of the normals failed");
        uint64_t want = std::max<uint64_t>(need + need / 2, normals_base + 2 * max_nodes * nstride);
        if (!host_blob_reserve(forest->normals_blob, want, landed)) {
            want = need;
            AH_REQUIRE(host_blob_reserve(forest->normals_blob, want, landed), AH_ERR_OUT_OF_MEMORY,
                       "host allocation of %llu bytes of normals failed", (unsigned long long)want);
        }
        forest->normals = forest->normals_blob.p;
        normals_cap = forest->normals_blob.cap - 16;
        return AH_OK;
    };

    // Digest the node table of a finished level (it arrived on the side stream): split records, children, statistics,
    // and which HostRec every node of the next level belongs to — the same walk the device did in k_next_emit.
    uint64_t items_routed = 0;
    // `rec_of`: HostRec index of every node of the level (level_rec of the level-by-level loop; a slice of the parent
    // level's list for the first level of a tree group of the tail)
    auto digest_level = [&](uint32_t depth, uint32_t n_nodes, const FNode *tbl, uint64_t chunk_host_off, const uint8_t *chunk_dev,
                            const uint32_t *rec_of) -> int {
        // Children get the records base + 2 i (left) and base + 2 i + 1 (right) of node i, so the walk splits over a few
        // threads (the deepest level of the 10M x 100-tree build has 819 000 nodes: 37 ms on one thread, more than the GPU
        // needs for the level after it); the list of children that split again is concatenated in node order afterwards.
        const size_t base = n_recs;
        if (recs.size() < base + 2 * (size_t)n_nodes) recs.resize(base + 2 * (size_t)n_nodes);
        n_recs = base + 2 * (size_t)n_nodes;
        // (a streaming build also writes the level's 48-byte stream nodes and collects its leaves: twice the threads)
        const unsigned n_threads = n_nodes >= 65536 ? std::min(sb ? 8u : 4u, host_threads) : 1u;
        struct Part {
            uint64_t evals = 0, retries = 0, routed = 0, dummies = 0;
            uint32_t bad = 0xFFFFFFFFu;
            std::vector<uint32_t> splits;
            std::vector<LeafRec> leaves;  // streaming build: the children that are Descendants nodes, in node order
            std::vector<std::pair<uint32_t, uint32_t>> runs;  // (tree, nodes of it in this part): the table is ordered by tree
        };
        std::vector<Part> parts(n_threads);
        // streaming build: this level's split planes as one job for the read-back worker (record i <-> node i)
        StreamJob *job = nullptr;
        if (sb) {
            job = new (std::nothrow) StreamJob();
            AH_REQUIRE(job, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
            job->head.kind = AH_NODE_SPLIT;
            job->head.level = depth;
            job->head.normal_stride = nstride;
            job->head.normal_vector_offset = 0;
            job->head.normal_header_offset = hdr_off;
            job->fixed_len = nstride;
            job->nodes.resize(n_nodes);
        }
        auto walk = [&](unsigned t) {
            Part &pt = parts[t];
            const uint32_t lo = (uint32_t)((uint64_t)n_nodes * t / n_threads), hi = (uint32_t)((uint64_t)n_nodes * (t + 1) / n_threads);
            pt.splits.reserve(2 * (size_t)(hi - lo));
            if (sb) pt.leaves.reserve(2 * (size_t)(hi - lo));
            for (uint32_t i = lo; i < hi; i++) {
                const FNode nd = tbl[i];
                if (nd.state == ST_PENDING || nd.n_left > nd.count) {
                    pt.bad = std::min(pt.bad, i);
                    continue;
                }
                if (pt.runs.empty() || pt.runs.back().first != nd.tree) pt.runs.push_back({nd.tree, 0u});
                pt.runs.back().second++;
                pt.evals += (uint64_t)(nd.attempt + 1) * nd.count;
                pt.retries += nd.attempt;
                pt.routed += nd.count;
                const uint32_t rec_idx = rec_of[i];
                recs[rec_idx].has_normal = nd.state == ST_ACCEPTED;
                recs[rec_idx].normal_off = chunk_host_off + (uint64_t)i * nstride;
                if (nd.state != ST_ACCEPTED) pt.dummies++;
                const uint32_t child_cnt[2] = {nd.n_left, nd.count - nd.n_left};
                const uint64_t child_start[2] = {nd.start, nd.start + nd.n_left};
                for (uint32_t side = 0; side < 2; side++) {
                    HostRec c{};
                    c.tree = nd.tree;
                    c.start = child_start[side];
                    c.count = child_cnt[side];
                    c.depth = depth + 1;
                    const uint32_t cidx = (uint32_t)(base + 2 * (size_t)i + side);
                    if (c.count <= split_after) {
                        c.kind = AH_NODE_DESCENDANTS;
                        if (sb) pt.leaves.push_back(LeafRec{c.start, c.count, id_base + cidx, first_tree + nd.tree, depth + 1});
                    } else {
                        c.kind = AH_NODE_SPLIT;
                        pt.splits.push_back(cidx);
                    }
                    recs[cidx] = c;
                    if (side == 0) recs[rec_idx].left = cidx;
                    else recs[rec_idx].right = cidx;
                }
                if (sb) {
                    ah_stream_node sn{};
                    sn.id = id_base + rec_idx;
                    sn.tree = first_tree + nd.tree;
                    sn.kind = AH_NODE_SPLIT;
                    sn.has_normal = nd.state == ST_ACCEPTED;
                    sn.left = id_base + (uint32_t)(base + 2 * (size_t)i);
                    sn.right = sn.left + 1;
                    sn.count = nd.count;
                    sn.depth = depth;
                    sn.payload_offset = (uint64_t)i * nstride;
                    job->nodes[i] = sn;
                }
            }
        };
        parallel_run(n_threads, walk);
        if (sb) {
            bool bad = false;
            for (const Part &pt : parts) bad = bad || pt.bad != 0xFFFFFFFFu;
            if (bad) {
                delete job;
            } else {
                stream_leaves.emplace_back();
                for (Part &pt : parts) stream_leaves.back().insert(stream_leaves.back().end(), pt.leaves.begin(), pt.leaves.end());
                rb.push_stream(job, chunk_dev);  // the level's planes travel while the next levels are computed
            }
        }
        next_rec.clear();
        for (const Part &pt : parts) {
            AH_REQUIRE(pt.bad == 0xFFFFFFFFu, AH_ERR_DEVICE, "forest build: node %u of level %u left pending (internal error)",
                       pt.bad, depth);
            forest->stats.margin_evaluations += pt.evals;
            forest->stats.retries += pt.retries;
            forest->stats.dummy_normals += pt.dummies;
            items_routed += pt.routed;
            next_rec.insert(next_rec.end(), pt.splits.begin(), pt.splits.end());
            for (const auto &run : pt.runs) tree_count[run.first] += 2ull * run.second;
        }
        level_rec.swap(next_rec);
        forest->stats.levels = std::max(forest->stats.levels, depth + 1);
        if (opt->progress) opt->progress(opt->progress_user, depth + 1, n_recs, items_routed);
        return AH_OK;
    };

    // Wait for the level on the stream.  With a cancel flag the wait polls it: a cancelled build raises the device-side
    // abort word (on the side stream), the margin kernels still queued or running drain without work, and the call
    // returns AH_ERR_CANCELLED as soon as the stream is idle (src/writer.rs:1178,1196 poll per node / per item).
    bool abort_sent = false;
    auto wait_level = [&]() -> int {
        if (!opt->cancel) {
            AH_HIP(hipEventSynchronize(bc.ev_level));
            return AH_OK;
        }
        for (uint32_t spins = 0;; spins++) {
            const hipError_t e = hipEventQuery(bc.ev_level);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) {
                set_error("forest build: %s", hipGetErrorString(e));
                return AH_ERR_DEVICE;
            }
            if (*opt->cancel && !abort_sent) {
                AH_HIP(hipMemcpyAsync(d_small.p, h_one, 4, hipMemcpyHostToDevice, bc.side));
                abort_sent = true;
            }
            if (spins < 4096) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (abort_sent || *opt->cancel) {
            set_error("build cancelled");
            return AH_ERR_CANCELLED;
        }
        return AH_OK;
    };

    uint32_t depth = 0;
    uint32_t step = 0;  // levels launched so far: the parity of the double buffers (== depth until the tail runs in groups)
    auto t_prev_waited = std::chrono::steady_clock::now();
    bool prev_rows = false;     // the previous level ran row-major: node_of / side_bytes describe it, d_child links it
    // Node tables that arrived on the side stream and are still to be digested, oldest first.  A table is digested under a
    // level that runs long enough to cover it (~45 ns per node on four threads against ~0.15 ns per pair on the device): the
    // table of the last big level — 819 000 nodes, 35 ms — used to be digested under the 9 ms remnant level after it, and
    // the loop, the hand-over of the ids and the next group of trees waited for it.  The tables live in a ring over the two
    // pinned table buffers; a table that finds no room there forces the oldest digests first.
    struct PendingTable {
        uint32_t depth, step, n_nodes;
        size_t ring_off;
        uint64_t host_off;
        const uint8_t *chunk_dev;
        int64_t rec_off;      // >= 0: its record indices are level_rec_full[off ...] (first level of a tree group)
        bool fork_parent;     // it is the level the groups were cut from: its children's list is kept whole
        int group_done;       // >= 0: the last level of that tree group — its node list (its leaves' job) can follow
    };
    std::deque<PendingTable> tables;
    uint8_t *const ring_base = reinterpret_cast<uint8_t *>(h_nodes[0]);
    const size_t ring_cap = 2 * pin_nodes;
    size_t ring_head = 0;
    auto ring_alloc = [&](size_t bytes, size_t *off) -> bool {  // FIFO ring; head == tail only when nothing is live
        if (tables.empty()) ring_head = 0;
        const size_t tail = tables.empty() ? 0 : tables.front().ring_off;
        if (tables.empty() || ring_head > tail) {
            if (ring_head + bytes <= ring_cap) {
                *off = ring_head;
                ring_head += bytes;
                return true;
            }
            if (!tables.empty() && bytes < tail) {
                *off = 0;
                ring_head = bytes;
                return true;
            }
            return false;
        }
        if (ring_head + bytes < tail) {
            *off = ring_head;
            ring_head += bytes;
            return true;
        }
        return false;
    };
    int lvl_status = AH_OK;

    // ---- the tail in groups of trees (AH_BUILD_TAIL_GROUPS) ----------------------------------------------------------
    // The item ids of a tree are final with its last scatter, and 95 % of them (10M x 768 x 100 trees) are still under
    // splitting nodes when the last big level starts: level by level, the 4 GB of ids and the 2.6 GB of that level's
    // normals leave the device after the last launch — 0.10 s of 1.43.  So the last big level and whatever follows it
    // run GROUP BY GROUP (contiguous ranges of trees, each to its end): the ids and normals of group g travel while
    // group g + 1 computes, and only the last group's are left for the tail.  A group is a level-by-level build of fewer
    // trees over the same tables: its nodes are a slice of the level's table (tile indices rebased by k_fork_groups), its
    // levels always node-major (they are by then), its records digested like any level's.  The forest is the same node
    // for node — trees never interact — only the order of the normals in the blob (and the ids of a streaming build)
    // follows the order of the launches.
    const uint32_t g_tail_groups = (uint32_t)std::min<long long>(32, std::max<long long>(0, tun(TUN_TAIL_GROUPS)));
    const double g_tail_node_items = (double)std::max<long long>(1, tun(TUN_TAIL_NODE_ITEMS));
    const uint64_t g_tail_min_items = (uint64_t)std::max<long long>(0, tun(TUN_TAIL_MIN_MB)) * (1u << 20) / 4;
    bool grouped = false, fork_now = false, group_first_level = false;
    uint32_t group_n0 = 0;
    std::vector<uint32_t> level_rec_full;
    std::vector<uint32_t> group_tree;        // tree range of group g: [group_tree[g], group_tree[g + 1])
    int pending_hand = -1;                   // group whose item ids are final and still to be handed to the read-back worker
    FNode *group_spare = nullptr;            // third node table of the grouped tail
    std::function<int(uint32_t, uint32_t)> hand_over_ids;  // (defined below, once the leaves' merge exists)
    std::function<int(uint32_t, uint32_t)> group_complete;  // node list / leaves' job of a finished group (defined below)
    std::deque<int> done_groups;  // groups whose tables are all digested and whose node list is still to be emitted
    auto digest_front = [&]() -> int {
        const PendingTable e = tables.front();
        // (copies on the side stream complete in order: the event of this parity was recorded by this table's copy or a later one)
        AH_HIP(hipEventSynchronize(bc.ev_copy[e.step & 1]));
        const uint32_t *rec_of = e.rec_off >= 0 ? level_rec_full.data() + e.rec_off : level_rec.data();
        AH_TRY(digest_level(e.depth, e.n_nodes, reinterpret_cast<const FNode *>(ring_base + e.ring_off), e.host_off, e.chunk_dev, rec_of));
        if (e.fork_parent) level_rec_full.swap(level_rec);
        tables.pop_front();
        if (e.group_done >= 0) {
            // a streaming build's leaves' job starts the transfer of the group's ids: at once; a node list only costs host time
            if (sb) AH_TRY(group_complete(group_tree[e.group_done], group_tree[e.group_done + 1]));
            else done_groups.push_back(e.group_done);
        }
        return AH_OK;
    };
    // the tables a level of `pairs` margin evaluations covers (all of them: pairs = ~0)
    auto digest_under = [&](uint64_t pairs) -> int {
        // host nanoseconds the level covers: 0.14 ns per pair on the device, three quarters of it at most; a table costs ~45 ns
        // per node (four threads), a node list ~10 ns per node (eight) — both next to the read-back worker's copy threads
        double budget = (double)pairs * 0.105;
        // (a table of a few thousand nodes is a fraction of a millisecond: always — progress reports stay level by level)
        while (!tables.empty() && (45.0 * tables.front().n_nodes <= budget || tables.front().n_nodes < 16384u)) {
            budget -= 45.0 * tables.front().n_nodes;
            AH_TRY(digest_front());
        }
        while (!done_groups.empty()) {
            const uint32_t tA = group_tree[done_groups.front()], tB = group_tree[done_groups.front() + 1];
            double nodes = 0.0;
            for (uint32_t t = tA; t < tB; t++) nodes += (double)tree_count[t];
            if (10.0 * nodes > budget) break;
            budget -= 10.0 * nodes;
            done_groups.pop_front();
            AH_TRY(group_complete(tA, tB));
        }
        return AH_OK;
    };
    auto digest_pending = [&]() -> int { return digest_under(~0ull); };
    // Streaming build: the Descendants nodes of trees [tA, tB), ascending in (tree, position) — i.e. in the order their ids lie
    // in the final permutation — as one job.  Every level's list is already in that order; the levels of one tree are merged
    // per tree (a few threads: 1.7 M leaves at 10M x 100 trees).
    auto push_leaves_job = [&](uint32_t tA, uint32_t tB) -> int {
        StreamJob *job = new (std::nothrow) StreamJob();
        AH_REQUIRE(job, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
        job->head.kind = AH_NODE_DESCENDANTS;
        job->head.normal_stride = nstride;
        job->head.normal_header_offset = hdr_off;
        // per tree: where its leaves start in every level's list, and in the output
        const size_t n_lv = stream_leaves.size();
        const uint32_t nt = tB - tA;
        std::vector<size_t> cut((size_t)(nt + 1) * n_lv, 0), out_at(nt + 1, 0);
        for (size_t l = 0; l < n_lv; l++) {
            const auto &lv = stream_leaves[l];
            // (by tree, not by position: an EMPTY child at the very end of a tree starts where the next tree does)
            for (uint32_t t = 0; t <= nt; t++)
                cut[(size_t)t * n_lv + l] = (size_t)(std::lower_bound(lv.begin(), lv.end(), first_tree + tA + t, [](const LeafRec &r, uint32_t v) { return r.tree < v; }) - lv.begin());
        }
        for (uint32_t t = 0; t < nt; t++) {
            size_t c = 0;
            for (size_t l = 0; l < n_lv; l++) c += cut[(size_t)(t + 1) * n_lv + l] - cut[(size_t)t * n_lv + l];
            out_at[t + 1] = out_at[t] + c;
        }
        const size_t total = out_at[nt];
        job->nodes.resize(total);
        std::atomic<uint32_t> next_tree{0};
        parallel_run(total < 100000 ? 1u : std::min(nt, host_threads), [&](unsigned) {
            std::vector<LeafRec> mine;
            for (;;) {
                const uint32_t t = next_tree.fetch_add(1, std::memory_order_relaxed);
                if (t >= nt) break;
                mine.clear();
                for (size_t l = 0; l < n_lv; l++)
                    mine.insert(mine.end(), stream_leaves[l].begin() + (ptrdiff_t)cut[(size_t)t * n_lv + l],
                                stream_leaves[l].begin() + (ptrdiff_t)cut[(size_t)(t + 1) * n_lv + l]);
                // (an empty leaf shares its position with its sibling: it goes first, so that offsets never step back)
                std::sort(mine.begin(), mine.end(), [](const LeafRec &a, const LeafRec &b) {
                    return a.start != b.start ? a.start < b.start : a.count < b.count;
                });
                ah_stream_node *dst = job->nodes.data() + out_at[t];
                for (const LeafRec &r : mine) {
                    ah_stream_node sn{};
                    sn.id = r.id;
                    sn.tree = r.tree;
                    sn.kind = AH_NODE_DESCENDANTS;
                    sn.count = r.count;
                    sn.depth = r.depth;
                    sn.payload_offset = r.start * 4;
                    *dst++ = sn;
                }
            }
        });
        rb.push_stream(job, final_perm.p);
        return AH_OK;
    };
    // The item ids of trees [tA, tB) are final (every kernel that writes them has been waited for): rows -> item ids where the
    // two differ (on the side stream: the main one is busy with the next group), then — the blob's slice of a materialised
    // forest — to the worker.  (A streaming build hands them over as the leaves' job, once the group's tables are digested.)
    hand_over_ids = [&](uint32_t tA, uint32_t tB) -> int {
        const uint64_t a = tree_base[tA], b = tree_base[tB];
        if (b > a && !ds->identity_ids) {
            hipLaunchKernelGGL(k_rows_to_ids, dim3(2048), dim3(256), 0, bc.side, final_perm.p + a, b - a, ds->d_ids);
            AH_HIP(hipStreamSynchronize(bc.side));
        }
        if (sb) return AH_OK;
        prefault.join();
        rb.push(forest->descendants + desc_base + a, final_perm.p + a, (b - a) * 4);
        return AH_OK;
    };
    // Emit per tree in post-order (children before parents: the order TmpNodes::put receives them,
    // src/writer.rs:1235-1258), with forest-local indices.  Trees are independent: the number of nodes of every tree
    // is known (counted while the levels were digested), so each tree is written into its own slice by a few threads —
    // all trees after the last level, or, when the tail runs in groups, a group's trees under the next group's kernels.
    const size_t node_base = forest->nodes.size(), roots_base = forest->roots.size();
    size_t emit_cursor = node_base;  // forest->nodes index of the next tree to be emitted
    uint32_t emit_tree = 0;          // trees [0, emit_tree) are emitted
    std::atomic<uint64_t> n_split{0}, n_desc{0};
    auto emit_range = [&](uint32_t tA, uint32_t tB) -> int {
        if (sb || tB <= tA) return AH_OK;
        AH_REQUIRE(tA == emit_tree, AH_ERR_DEVICE, "forest build: trees emitted out of order (internal error)");
        std::vector<uint64_t> off(tB - tA + 1, 0);
        for (uint32_t t = tA; t < tB; t++) off[t - tA + 1] = off[t - tA] + tree_count[t];
        const uint64_t total = off[tB - tA];
        forest->nodes.resize(emit_cursor + total);
        if (forest->roots.size() < roots_base + n_trees) forest->roots.resize(roots_base + n_trees);
        new_index.resize(n_recs, 0xFFFFFFFFu);
        uint32_t *roots_out = forest->roots.data() + roots_base;
        std::atomic<uint32_t> next_tree{tA};
        auto emit_trees = [&](unsigned) {
            std::vector<std::pair<uint32_t, int>> stack;
            uint64_t splits = 0, descs = 0;
            for (;;) {
                const uint32_t t = next_tree.fetch_add(1, std::memory_order_relaxed);
                if (t >= tB) break;
                uint64_t out = emit_cursor + off[t - tA];
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
                    nd.tree = first_tree + t;
                    nd.count = r.count;
                    nd.depth = r.depth;
                    if (r.kind == AH_NODE_SPLIT) {
                        nd.left = new_index[r.left];
                        nd.right = new_index[r.right];
                        nd.offset = r.normal_off;
                        splits++;
                    } else {
                        nd.offset = desc_base + r.start;
                        descs++;
                    }
                    new_index[ri] = (uint32_t)out;
                    forest->nodes[out++] = nd;
                    stack.pop_back();
                }
                roots_out[t] = new_index[tree_root[t]];
            }
            n_split.fetch_add(splits, std::memory_order_relaxed);
            n_desc.fetch_add(descs, std::memory_order_relaxed);
        };
        parallel_run(total < 200000 ? 1u : std::min({tB - tA, host_threads, std::max(1u, std::thread::hardware_concurrency())}),
                     emit_trees);
        emit_cursor += total;
        emit_tree = tB;
        return AH_OK;
    };
    group_complete = [&](uint32_t tA, uint32_t tB) -> int { return sb ? push_leaves_job(tA, tB) : emit_range(tA, tB); };
    auto run_levels = [&]() -> int {
    while (info.n_nodes) {
        if (opt->cancel && *opt->cancel) {
            set_error("build cancelled");
            return AH_ERR_CANCELLED;  // Error::BuildCancelled
        }
        AH_REQUIRE(depth < 100000, AH_ERR_DEVICE, "forest build: depth %u exceeded (internal error)", depth);
        if (sb && (sb->target.sink_rc.load() || sb->target.too_big.load())) break;  // reported after the loop
        const auto t_level_top = std::chrono::steady_clock::now();
        const double ms_tail_prev = std::chrono::duration<double, std::milli>(t_level_top - t_prev_waited).count();
        const uint32_t n_nodes = info.n_nodes, n_tiles = info.n_tiles;
        AH_REQUIRE(n_nodes <= max_nodes && n_tiles <= max_tiles, AH_ERR_DEVICE, "forest build: node / tile bound exceeded");
        const unsigned tile_grid = std::min<uint32_t>(n_tiles, g_tile_blocks);
        // The node-major margin kernels walk their tiles with a persistent grid: at the deep levels a tile is ~1200 items
        // (~2 MB of rows) and launching one workgroup per tile — 819 000 of them at level 13 of the 10M x 100 build — cost
        // 20 % of the level (measured: 306 -> 236 ms, the same 1.70 TB of fabric reads, TCC_EA0_RDREQ).  The small
        // bookkeeping kernels keep one tile per block (their random 1-byte / 4-byte gathers want every block in flight),
        // except the byte -> mask conversion of the top levels, whose reads are still nearly sequential.
        const unsigned node_grid = g_node_blocks ? std::min<uint32_t>(n_tiles, g_node_blocks)
                                                 : std::min<uint32_t>(n_tiles, std::max<uint32_t>(2048u, std::min<uint32_t>(65536u, n_tiles / 16u)));
        const unsigned masks_grid = n_nodes <= 4 * n_trees ? std::min<uint32_t>(tile_grid, 32768u) : tile_grid;

        // ---- margin mode of the first attempt ------------------------------------------------------------------------
        // Row-major streams all N rows once per group of row_tc trees; node-major reads only the still-active items, once
        // per tree.  A forced mode (ah_build_options.margin_mode) pins the kernel family wherever it is legal.
        const uint64_t rec_bytes = screen ? hstride : nstride;  // bytes of one normal as the margin pass streams it
        const uint64_t nodes_per_tree = (n_nodes + n_trees - 1) / n_trees;
        uint32_t row_tc = 0, lds_tc = 0, lds_worst = 0;
        double best_cost = -1.0;  // AUTO: modelled cost in ns of the cheapest of node-major / row-major for this level
        if (grouped) {
            // a tree group of the tail: node-major (what the level it was cut from had chosen)
        } else if (rows_allowed && mode_req == AH_MARGIN_AUTO) {
            // Cost model in ns per row of 3072 bytes, fitted to per-level rocprofv3 traces of the 10M x 768 x 100-tree
            // build (profiles/): node-major = one HBM read of the row per (item, tree) pair; row-major = per pass and
            // row the HBM read of the row plus row_tc normals through the vector memory pipeline (L2 while the group's
            // normals of the level fit, the fabric beyond), scaled by the share of (row, tree) pairs still splitting,
            // plus the node_of / mask conversion of the level.
            // costs in ns, for rows of this dataset's size: node-major = one HBM read of the row per (item, tree) pair
            // (0.47 ns per 3072-byte row at 6.7 TB/s; the screen reads half the bytes)
            const double scale = screen ? (double)ds->hpitch * 2 / 1536.0 : (double)ds->row_bytes() / 3072.0;
            // (with the int8 first stage: 768 bytes + ~10 % x 768 (second digit) per pair on data that quantises like the
            // benchmark's; worse data decide less there: scaled by the quality figure ensure_screen8 measured)
            // (measured, 10M x 768 x 100 trees: 0.130-0.142 ns per pair on uniform rows, quality 0.12, 0.140-0.151 on ~N(0,1)
            // rows, quality 0.18, with the rows' second digit as stage 1; 0.152-0.170 / 0.170-0.186 with the binary16 row)
            // (+0.02: the row-major entries of the tables below carry their mask conversion twice — level 9 of that build,
            // measured both ways, takes 130.7 ms row-major and 136.9 ms node-major, and this keeps it row-major)
            const double node8_ns = sv.rows8_lo ? std::min(0.24, 0.135 + 0.15 * ds->screen8_quality)
                                                : std::min(0.24, 0.155 + 0.125 * ds->screen8_quality);
            const double node_ns = screen ? (screen8 && ds->metric != AH_DOT_PRODUCT ? node8_ns : 0.24) : 0.47;
            const double active = (double)info.pairs / ((double)n_trees * (double)N);
            const double cost_node = (double)info.pairs * node_ns * scale;
            const double convert = (double)n_trees * (double)N *
                                   ((prev_rows && g_rows_advance ? 0.007 : 0.01 + 0.03 * std::min(1.0, (double)nodes_per_tree / 256.0)) +
                                    0.012 * std::min(1.0, (double)nodes_per_tree / 512.0));
            // A tc-slot schedule costs what its launches cost: the full groups in one chunk-major launch, then the groups the
            // launch loop below forms for the last trees (16 + ... + 4, or one more full group when only a few of its slots
            // would idle: idle slots cost what busy ones cost), each a launch of its own and priced as the pass of ITS size.
            // The table of pass costs was fitted on 96-tree builds, six or more groups in flight; a launch of one or two
            // 16-slot groups has the L2s to itself and behaves like one with half the normals (13-tree share, levels 6-9
            // forced both ways: 13.1 / 13.5 / 15.5 / 21.7 ms as one 16-slot pass — the table says 12.5 / 13.1 / 14.7 / 20.5
            // that way — and 16.9 / 17.2 / 17.8 / 20.7 as 8 + 4 + 1, which the table prices as it stands).
            // Counting 13 trees as 13/16 of a 16-slot pass and 13/8 of an 8-slot one kept level 5 of that share row-major
            // (13 ms; the dense product takes 11), level 8 on 8 + 4 + 1 and level 9 row-major (19.5; node-major 18.3).
            double best = 0.97 * cost_node;
            const double floor_ns = screen ? 0.15 : 0.45;  // the read of the row alone (screened: mostly Infinity Cache)
            auto launch_ns_per_row = [&](uint32_t tcv, uint32_t groups) {
                double ws_mb = (double)((uint64_t)tcv * nodes_per_tree * rec_bytes) / 1e6;
                if (groups <= 2 && tcv >= 16) ws_mb *= 0.5;
                return groups * (floor_ns + (rows_pass_ns_per_row(tcv, ws_mb, screen) - floor_ns) * std::min(1.0, active * 1.05));
            };
            for (uint32_t tc = std::min(16u, g_rows_max_tc); tc >= 2; tc >>= 1) {
                if (tc > 2 && tc / 2 >= n_trees) continue;  // do not instantiate more slots than trees
                const double ws_mb = (double)((uint64_t)tc * nodes_per_tree * rec_bytes) / 1e6;
                if (g_rows_cache_mb > 0 && ws_mb > g_rows_cache_mb) continue;
                double ns_per_row = n_trees / tc ? launch_ns_per_row(tc, n_trees / tc) : 0.0;
                for (uint32_t t0 = n_trees / tc * tc; t0 < n_trees;) {
                    const uint32_t rem = n_trees - t0;
                    uint32_t tcv = tc;
                    while (tcv > 2 && tcv > rem) tcv >>= 1;
                    for (uint32_t up = tcv << 1; up <= tc && tcv < rem; up <<= 1)
                        if (up >= rem && up - rem <= (up >= 16 ? 3u : 1u)) tcv = up;
                    ns_per_row += launch_ns_per_row(tcv, 1);
                    t0 += std::min(tcv, rem);
                }
                const double cost_rows = (double)N * ns_per_row * scale + convert;
                if (cost_rows < best || (g_rows_force == 1 && row_tc == 0)) {
                    best = std::min(best, cost_rows);
                    row_tc = tc;
                }
            }
            best_cost = best;
        } else if (rows_allowed && mode_req != AH_MARGIN_DENSE_MFMA) {
            row_tc = mode_req & 0xFFu;
        }
        // Dense screen of the level on the matrix units (dense_device.h): one product rows x normals^T instead of
        // n_trees / tc passes.  Cost: the rows leave HBM once (binary16), hpitch multiply-adds per (row, column) at the
        // sustained MFMA rate, an epilogue per (row, tree), and the reference arithmetic for the pairs left open.
        bool dense = false;
        // (a launch carries at most 2^32 - 1 work-items: threads x row tiles x column tiles of the shape the level would take)
        const DensePlan dp_legal = dense_plan(N, std::max(n_nodes, 1u));
        const bool dense_legal = !grouped && rows_allowed && screen && g_dense != 0 && n_nodes <= g_dense_max_cols &&
                                 (dp_legal.grid + 64) * (uint64_t)dp_legal.threads < 0xFFFFFFFFull && dp_legal.grid < 0x7FFFFFFFull;
        if (dense_legal && mode_req == AH_MARGIN_DENSE_MFMA) {
            dense = true;
        } else if (dense_legal && mode_req == AH_MARGIN_AUTO) {
            // measured (10M x 768, 100 and 13 trees, gpurun_out/dense): the product itself 0.17 ns per row + 0.0023 ns per
            // (row, tree) of epilogue + hpitch multiply-adds per (row, padded column) at 495e3 MAC/ns (990 TF), never below
            // the 0.40 ns per row of the narrow kernel; 0.008 ns per pair for k_forest_exact_pairs
            const uint32_t bn = n_nodes > 128 ? 256u : 128u;
            const double cols = (double)(((uint64_t)n_nodes + bn - 1) / bn * bn);
            const double hscale = (double)ds->hpitch / 768.0;
            // (short rows: a tile's k-loop is a few latency-bound steps and the epilogue does not shrink with the row — never
            // below 0.25 ns per row; the exact pass scans a byte per pair whatever the row length)
            const double per_row = std::max(std::max(0.40 * hscale, 0.25), 0.17 * hscale + 0.0023 * n_trees + cols * (double)ds->hpitch / g_dense_gmacs);
            const double convert = (double)n_trees * (double)N * (prev_rows && g_rows_advance ? 0.007 : 0.04);
            const double cost_dense = (double)N * per_row + (double)info.pairs * (0.002 + 0.006 * (double)ds->row_bytes() / 3072.0) + convert;
            dense = g_dense == 1 || best_cost < 0 || cost_dense < best_cost;
        }
        if (dense) row_tc = 16;  // the level is row-major as far as node_of / side_bytes / the next level are concerned
        // Top levels: all normals of a group of >= 8 trees fit in LDS -> the LDS-resident variant of the row-major pass.
        const bool want_lds = dense ? false : mode_req == AH_MARGIN_AUTO ? g_rows_lds && row_tc >= 2 : (mode_req & 0x100u) != 0;
        if (!grouped && rows_allowed && want_lds && (rec_bytes & 15) == 0) {
            tree_first[n_trees] = n_nodes;  // trees without a node in this level start where the next tree starts
            for (uint32_t t = n_trees; t-- > 0;) tree_first[t] = std::min(tree_first[t], tree_first[t + 1]);
            // measured per tree and row: screened 0.039 ns for 16-tree groups, 0.040 for 8-tree groups; f32 0.061 / 0.070
            const uint32_t tc_hi = mode_req == AH_MARGIN_AUTO ? std::min<uint32_t>(16, g_rows_max_tc) : (mode_req & 0xFFu);
            const uint32_t tc_lo = mode_req == AH_MARGIN_AUTO ? 8u : tc_hi;
            for (uint32_t tc = tc_hi; tc >= tc_lo && tc >= 8; tc >>= 1) {
                uint32_t worst = 0;
                for (uint32_t t0 = 0; t0 < n_trees; t0 += tc)
                    worst = std::max(worst, tree_first[std::min(n_trees, t0 + tc)] - tree_first[t0]);
                if ((uint64_t)worst * rec_bytes <= kLdsNormalsBytes) {
                    lds_tc = tc;
                    lds_worst = worst;
                    break;
                }
            }
            if (lds_tc) row_tc = lds_tc;
            else if (mode_req != AH_MARGIN_AUTO) row_tc = 0;  // a forced LDS mode that does not fit: node-major
        }

        // ---- the tail in groups? (see above) ---------------------------------------------------------------------------
        // This level would run node-major, most of its children will be Descendants nodes (fewer than 2 x split_after
        // items per node on average), every tree still has nodes in it, and the ids under them are worth the trouble:
        // hand the level back unlaunched; the caller cuts it into groups of trees.
        // (Skewed data — clusters of duplicates that keep a few nodes large for another thirty levels while most of the
        // forest is done — never meet the first condition and stay level by level: cut at the level where a tenth of the
        // ids is final, the 10M x 100-tree build on AH_SYNTH_CLUSTERED rows ran 29 levels five times over, 2.56 s against
        // 2.46: every level carries four attempts' launches whatever its size.)
        if (!grouped && !fork_now && g_tail_groups >= 2 && n_trees >= 2 && !subset_ids && row_tc < 2 && !dense &&
            info.pairs >= g_tail_min_items && (double)info.pairs <= g_tail_node_items * (double)split_after * (double)n_nodes) {
            bool all = true;
            for (uint32_t t = 0; t < n_trees && all; t++)
                all = tree_first[t] != 0xFFFFFFFFu && (t == 0 ? tree_first[0] == 0 : tree_first[t] > tree_first[t - 1]);
            if (all) {
                fork_now = true;
                return AH_OK;
            }
        }
        hipLaunchKernelGGL(k_build_tiles, dim3(std::min<uint32_t>((n_nodes + 3) / 4, kMaxBlocks)), dim3(256), 0, s, d_cur,
                           n_nodes, d_tiles.p);
        uint8_t *chunk_d = nullptr, *shadow_d = nullptr, *shadow8_d = nullptr;
        const uint64_t chunk_bytes = (uint64_t)n_nodes * nstride;
        const uint64_t chunk_host_off = normals_base + normals_bytes;
        AH_TRY(arena.take(chunk_bytes, &chunk_d));
        if (screen) AH_TRY(shadow_arena.take((uint64_t)n_nodes * hstride, &shadow_d));
        if (screen8) AH_TRY(shadow8_arena.take((uint64_t)n_nodes * stride8, &shadow8_d));
        normals_bytes += chunk_bytes;
        // Host side of the normals: reserved now; the pages of THIS level's records were started one level ago (a level has
        // about twice the nodes of the one before), those of the NEXT level start now — committing the 2.6 GB of the deepest
        // level takes longer than the ~160 ms that level runs, and the loop must not wait for page faults before it can
        // hand a chunk to the read-back worker.  Whatever the prediction missed is faulted in by the worker itself.
        if (!sb) AH_TRY(reserve_normals(chunk_host_off + chunk_bytes, chunk_host_off));
        if (!sb && depth == 0) touch_normals[0].start(forest->normals_blob, chunk_host_off, chunk_bytes);
        if (!sb) {
            // next level: twice the nodes while the nodes are large; once they hold fewer than 2 x split_after items on
            // average most children are Descendants and the next level is a remnant (an eighth); nothing below that
            const uint64_t next_begin = chunk_host_off + chunk_bytes;
            const double avg_items = n_nodes ? (double)info.pairs / (double)n_nodes : 0.0;
            const uint64_t predicted = avg_items > 2.0 * split_after ? 2 * chunk_bytes : avg_items > (double)split_after ? chunk_bytes / 8 : 0;
            const uint64_t next_len = normals_cap > next_begin ? std::min<uint64_t>(normals_cap - next_begin, predicted) : 0;
            touch_normals[(step + 1) & 1].start(forest->normals_blob, next_begin, (size_t)next_len);
        }
        for (int attempt = 0; attempt < 4; attempt++) {
            // (attempts 1-3: gated by the number of nodes the attempt before left pending, d_small[attempt])
            const uint32_t *gate = attempt && g_retry_gate ? d_small.p + attempt : nullptr;
            const AbortFlags abort_a{d_small.p, gate};
#define AH_SPLIT_LAUNCH(K)                                                                                             \
    hipLaunchKernelGGL(K, dim3(std::min<uint32_t>(n_nodes, g_split_blocks)), dim3(64), cs_shared, s, dv, d_cur, n_nodes, cur, N, \
                       chunk_d, nstride, hdr_off, gate)
            AH_SPLIT_KERNEL(AH_SPLIT_LAUNCH)
#undef AH_SPLIT_LAUNCH
            AH_DBG(s, "create_split");
            if (screen)
                hipLaunchKernelGGL(k_forest_shadow_normals, dim3(std::min<uint32_t>(n_nodes, attempt ? 65536u : 1u << 20)), dim3(64), 0, s,
                                   dv, d_cur, chunk_d, nstride, hdr_off, shadow_d, hstride, sv.hpitch, n_nodes, gate);
            // the int8 records are only read by the node-major screen: the first attempt of a row-order level skips them
            if (screen8 && !(attempt == 0 && row_tc >= 2))
                hipLaunchKernelGGL(k_forest_shadow_normals8, dim3(std::min<uint32_t>(n_nodes, 65536u)), dim3(64), 0, s, dv, d_cur,
                                   n_nodes, chunk_d, nstride, hdr_off, ds->d_dim_scale, shadow8_d, stride8, sv.pitch8, gate);
            AH_HIP(hipEventRecord(bc.ev_attempt[2 * attempt], s));
            if (attempt == 0 && row_tc >= 2) {
                // one pass over the rows serves up to row_tc trees (see k_forest_margin_rows)
                if (prev_rows && g_rows_advance) {
                    hipLaunchKernelGGL(k_forest_advance_node_of, dim3(kMaxBlocks), dim3(kBlock), 0, s, node_of.p, side_bytes.p,
                                       d_child.p, (uint64_t)n_trees * N);
                    if (info.n_fix)
                        hipLaunchKernelGGL(k_forest_assign_node_of, dim3(std::min<uint32_t>(n_tiles, kMaxBlocks)), dim3(kBlock), 0,
                                           s, d_cur, d_tiles.p, n_tiles, cur, N, node_of.p, 1u);
                } else {
                    AH_HIP(hipMemsetAsync(node_of.p, 0xFF, (size_t)n_trees * N * 4, s));
                    hipLaunchKernelGGL(k_forest_assign_node_of, dim3(tile_grid), dim3(kBlock), 0, s, d_cur, d_tiles.p, n_tiles,
                                       cur, N, node_of.p, 0u);
                }
                uint32_t passes = 0;
                if (dense) {
                    DenseArgs da{};
                    da.rows = sv.rows;
                    da.stats = sv.stats;
                    da.headers = dv.headers;
                    da.n = N;
                    da.hpitch = sv.hpitch;
                    da.shadow = shadow_d;
                    da.hstride = hstride;
                    da.n_cols = n_nodes;
                    da.nodes = d_cur;
                    da.node_of = node_of.p;
                    da.side_bytes = side_bytes.p;
                    // accumulation error of the product: every one of the hpitch exact binary16 products enters one f32
                    // chain; whatever the order and the rounding mode of the matrix unit's adders (nearest or truncating,
                    // aligned to the largest addend of a 16-wide step or not), the sum is off by less than
                    // (hpitch + hpitch / 16) * 2^-23 * sum |x~_i n~_i|; taken twice over
                    da.gamma_s = (float)(2.0 * ((double)sv.hpitch + (double)sv.hpitch / 16 + 16.0) * 1.1920929e-7);
                    da.gamma_r = sv.gamma_r;
                    const DensePlan dp = dense_plan(N, n_nodes);
                    da.n_row_tiles = dp.n_row_tiles;
                    da.n_col_tiles = dp.n_col_tiles;
                    da.group = dp.group;
                    da.verify = verify;
                    const bool wide = dp.wide;
                    const uint64_t dgrid = dp.grid;
                    AH_REQUIRE(dgrid < 0x7FFFFFFFull, AH_ERR_INVALID_ARGUMENT, "forest build: too many tiles for one launch");
                    const unsigned egrid = exact_pairs_grid(N, n_trees);
#define AH_DENSE_WN(M, WNV)                                                                                              \
    do {                                                                                                                 \
        static std::atomic<bool> dense_opt_in[64]; /* once per instantiation and device */                              \
        if (!dense_opt_in[ds->device & 63].load(std::memory_order_acquire)) {                                            \
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_dense_screen<M, WNV>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDenseLds));                     \
            dense_opt_in[ds->device & 63].store(true, std::memory_order_release);                                        \
        }                                                                                                                \
        hipLaunchKernelGGL((k_forest_dense_screen<M, WNV>), dim3((unsigned)dgrid), dim3(DenseShape<WNV>::kThreads),      \
                           kDenseLds, s, da, d_abort);                                                                   \
    } while (0)
#define AH_DENSE_NARROW(M, NTV)                                                                                          \
    do {                                                                                                                 \
        if (dp.stream)                                                                                                   \
            hipLaunchKernelGGL((k_forest_dense_narrow<M, NTV, true>), dim3((unsigned)dgrid), dim3(kNarrowThreads),       \
                               NarrowShape<NTV>::kLds, s, da, d_abort);                                                  \
        else                                                                                                             \
            hipLaunchKernelGGL((k_forest_dense_narrow<M, NTV, false>), dim3((unsigned)dgrid), dim3(kNarrowThreads),      \
                               NarrowShape<NTV>::kLds, s, da, d_abort);                                                  \
    } while (0)
#define AH_DENSE(M)                                                                                                      \
    do {                                                                                                                 \
        if (dp.narrow == 2) AH_DENSE_NARROW(M, 2);                                                                       \
        else if (dp.narrow == 4) AH_DENSE_NARROW(M, 4);                                                                  \
        else if (wide) AH_DENSE_WN(M, 4);                                                                                \
        else AH_DENSE_WN(M, 2);                                                                                          \
        if (tun(TUN_EXACT_WIDE) != 0)                                                                                    \
            hipLaunchKernelGGL((k_forest_exact_pairs<M, true>), dim3(egrid), dim3(256), 0, s, dv, node_of.p, side_bytes.p, \
                               n_trees, chunk_d, nstride, hdr_off, d_counters, d_abort);                                 \
        else                                                                                                             \
            hipLaunchKernelGGL((k_forest_exact_pairs<M>), dim3(egrid), dim3(256), 0, s, dv, node_of.p, side_bytes.p,     \
                               n_trees, chunk_d, nstride, hdr_off, d_counters, d_abort);                                 \
    } while (0)
                    switch (ds->metric) {
                    case AH_EUCLIDEAN: AH_DENSE(AH_EUCLIDEAN); break;
                    case AH_MANHATTAN: AH_DENSE(AH_MANHATTAN); break;
                    case AH_COSINE: AH_DENSE(AH_COSINE); break;
                    default: AH_DENSE(AH_DOT_PRODUCT); break;
                    }
#undef AH_DENSE_WN
#undef AH_DENSE_NARROW
#undef AH_DENSE
                    forest->stats.dense_launches++;
                    forest->stats.dense_columns += n_nodes;
                    passes = 1;
                } else if (screen) {
                    // Chunk-major launches (RowsSchedule): all full groups of the level in ONE launch, so that the passes over a
                    // chunk of rows run back to back and find it in the Infinity Cache; the last trees (fewer than a group) in
                    // launches of their own.  The LDS variant reads the first node of every tree from a device copy.
                    const uint32_t gtc = lds_tc ? lds_tc : row_tc;
                    ScreenRowsArgs ra{dv, sv, node_of.p, 0, 0, chunk_d, nstride, hdr_off, shadow_d, hstride, side_bytes.p,
                                      RowsSchedule{}, d_abort, d_counters, verify};
                    if (lds_tc) {
                        memcpy(h_tree_first, tree_first.data(), ((size_t)n_trees + 1) * 4);
                        AH_HIP(hipMemcpyAsync(d_tree_first_fixed, h_tree_first, ((size_t)n_trees + 1) * 4, hipMemcpyHostToDevice, s));
                    }
                    const size_t lds_sh = (size_t)lds_worst * rec_bytes;
                    auto launch = [&](uint32_t tcv, uint32_t t0, uint32_t groups, uint32_t np) -> int {
                        RowsPlan plan;
                        AH_TRY(plan_rows_launches(N, sv.hpitch, tcv, groups, lds_tc != 0, (uint64_t)tcv * nodes_per_tree * rec_bytes, &plan));
                        ra.tree_base = t0;
                        ra.n_pass = np;
                        ra.sch = plan.sch;
                        ra.sch.tree_first = d_tree_first_fixed;
                        for (const auto &l : plan.launches) {
                            ra.sch.chunk0 = l.first;
                            AH_TRY(launch_screen_rows(ds->metric, tcv, lds_tc != 0, ra, l.second, lds_sh, s, ds->device));
                        }
                        forest->stats.margin_mode_launches[lds_tc ? (tcv >= 16 ? MM_LDS16 : MM_LDS8) : mm_rows(tcv)]++;
                        forest->stats.rows_xcd_launches += plan.xcd ? plan.launches.size() : 0;
                        forest->stats.rows_nt_launches += plan.nt ? plan.launches.size() : 0;
                        forest->stats.rows_split_launches += plan.launches.size() - 1;
                        passes += groups;
                        return AH_OK;
                    };
                    const uint32_t full = n_trees / gtc;
                    if (full) AH_TRY(launch(gtc, 0, full, gtc));
                    for (uint32_t t0 = full * gtc; t0 < n_trees;) {
                        // The last trees take the largest instantiation that they fill, or the next larger one when only a few
                        // of its slots would idle (13 trees: one 16-slot pass, measured 15 ms, instead of 8 + 4 + 1: 18-23 ms;
                        // spare slots repeat the last tree).  LDS groups exist for 8 and 16 trees only.
                        const uint32_t rem = n_trees - t0;
                        uint32_t tcv = gtc;
                        if (!lds_tc) {
                            while (tcv > 2 && tcv > rem) tcv >>= 1;
                            for (uint32_t up = tcv << 1; up <= gtc && tcv < rem; up <<= 1)
                                if (up >= rem && up - rem <= (up >= 16 ? 3u : 1u)) tcv = up;
                        }
                        const uint32_t np = std::min<uint32_t>(tcv, rem);
                        AH_TRY(launch(tcv, t0, 1, np));
                        t0 += np;
                    }
                }
                const unsigned row_grid = (unsigned)std::min<uint64_t>((N + 31) / 32, g_row_blocks);
                for (uint32_t t0 = 0; t0 < n_trees && lds_tc && !screen; t0 += lds_tc) {
                    const uint32_t np = std::min<uint32_t>(lds_tc, n_trees - t0);
                    const uint32_t first = tree_first[t0], cnt = tree_first[t0 + np] - first;
                    if (cnt == 0) continue;
                    const size_t sh = (size_t)cnt * rec_bytes;
                    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(2, (150u << 10) / std::max<size_t>(sh, 1)));
                    const unsigned lthreads = lds_tc >= 16 ? 512u : 1024u;
                    const unsigned lgrid = (unsigned)std::min<uint64_t>((N * 8 + lthreads - 1) / lthreads, 256u * per_cu);
#define AH_LDS_OPT_IN(KERNEL)                                                                                          \
    do {                                                                                                               \
        static std::atomic<bool> lds_opt_in[64]; /* once per instantiation and device */                               \
        if (!lds_opt_in[ds->device & 63].load(std::memory_order_acquire)) {                                            \
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)kLdsNormalsBytes));                                                        \
            lds_opt_in[ds->device & 63].store(true, std::memory_order_release);                                        \
        }                                                                                                              \
    } while (0)
#define AH_ROWS_LDS(M, TCV)                                                                                              \
    do {                                                                                                                 \
        AH_LDS_OPT_IN((k_forest_margin_rows_lds<M, TCV>));                                                               \
        hipLaunchKernelGGL((k_forest_margin_rows_lds<M, TCV>), dim3(lgrid), dim3(lthreads), sh, s, dv, node_of.p, t0, np,  \
                           chunk_d, nstride, hdr_off, side_bytes.p, first, cnt, d_abort);                                \
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
                    forest->stats.margin_mode_launches[lds_tc == 16 ? MM_LDS16 : MM_LDS8]++;
                }
                for (uint32_t t0 = 0; t0 < n_trees && !lds_tc && !screen;) {
                    // The last trees of the forest take the largest instantiation that they fill (16 + 16 + ... + 4), or the
                    // next larger one when only a few of its slots would idle (13 trees: one 16-slot pass, measured 15 ms,
                    // instead of 8 + 4 + 1: 18-23 ms; spare slots repeat the last tree).
                    const uint32_t rem = n_trees - t0;
                    uint32_t tcv = row_tc;
                    while (tcv > 2 && tcv > rem) tcv >>= 1;
                    for (uint32_t up = tcv << 1; up <= row_tc && tcv < rem; up <<= 1)
                        if (up >= rem && up - rem <= (up >= 16 ? 3u : 1u)) tcv = up;
                    const uint32_t np = std::min<uint32_t>(tcv, rem);
#define AH_ROWS(M, TCV)                                                                                                    \
    do {                                                                                                                   \
        hipLaunchKernelGGL((k_forest_margin_rows<M, TCV>), dim3(row_grid), dim3(kBlock), 0, s, dv, node_of.p, t0, np,      \
                           chunk_d, nstride, hdr_off, side_bytes.p, d_abort);                                              \
    } while (0)
#define AH_ROWS_TC(M)                       \
    switch (tcv) {                          \
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
                    forest->stats.margin_mode_launches[mm_rows(tcv)]++;
                    passes++;
                    t0 += np;
                }
                if (side_bits.p) {
                    const uint64_t n_words = ((uint64_t)n_trees * N + 31) / 32;
                    hipLaunchKernelGGL(k_forest_pack_sides, dim3((uint32_t)std::min<uint64_t>((n_words + 255) / 256, 1u << 16)),
                                       dim3(256), 0, s, side_bytes.p, n_words, side_bits.p);
                }
                hipLaunchKernelGGL(k_forest_masks_from_bytes, dim3(masks_grid), dim3(kBlock), 0, s, d_cur, d_tiles.p, n_tiles,
                                   cur, N, side_bytes.p, side_bits.p, masks.p, tile_left.p);
                if (lds_tc && !screen) passes = (n_trees + lds_tc - 1) / lds_tc;
                forest->stats.margin_row_passes += passes;
                if (screen) forest->stats.screened_launches += passes;
            } else if (bq) {
                hipLaunchKernelGGL(k_forest_margin_bq, dim3(node_grid), dim3(kBlock), dv.pitch * 8, s, dv, d_cur, d_tiles.p,
                                   n_tiles, cur, N, chunk_d, nstride, hdr_off, masks.p, tile_left.p, abort_a);
                forest->stats.margin_mode_launches[MM_BQ]++;
            } else if (screen) {
                const size_t sh = (size_t)dv.pitch * 4 + (size_t)sv.hpitch * 2 + (screen8 ? 2 * (size_t)sv.pitch8 : 0);
                // rows of at most six 128-byte int8 steps (768 dimensions): the pipelined variant of the int8 stage
                const bool node_pre = screen8 && ds->metric != AH_DOT_PRODUCT && sv.pitch8 <= 6 * 128 && tun(TUN_NODE_PREFETCH) != 0;
#define AH_LAUNCH_PRE(M, PRE)                                                                                            \
    do {                                                                                                                  \
        if (sh > 48 * 1024) /* very long vectors: opt in to more dynamic LDS than the default limit */                    \
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_screen_node<M, PRE>),                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                             \
        hipLaunchKernelGGL((k_forest_screen_node<M, PRE>), dim3(node_grid), dim3(kBlock), sh, s, dv, sv, d_cur, d_tiles.p, \
                           n_tiles, cur, chunk_d, nstride, hdr_off, shadow_d, hstride, shadow8_d, stride8, masks.p,       \
                           tile_left.p, abort_a, d_counters, verify);                                                     \
    } while (0)
#define AH_LAUNCH(M)                          \
    do {                                      \
        if (node_pre) AH_LAUNCH_PRE(M, 6);    \
        else AH_LAUNCH_PRE(M, 0);             \
    } while (0)
                switch (ds->metric) {
                case AH_EUCLIDEAN: AH_LAUNCH(AH_EUCLIDEAN); break;
                case AH_MANHATTAN: AH_LAUNCH(AH_MANHATTAN); break;
                case AH_COSINE: AH_LAUNCH(AH_COSINE); break;
                default: AH_LAUNCH(AH_DOT_PRODUCT); break;
                }
#undef AH_LAUNCH
#undef AH_LAUNCH_PRE
                forest->stats.margin_mode_launches[MM_NODE]++;
                forest->stats.screened_launches++;
            } else {
                const size_t sh = (size_t)dv.pitch * 4;
#define AH_LAUNCH(M)                                                                                                \
    do {                                                                                                            \
        if (sh > 48 * 1024)                                                                                         \
            AH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_forest_margin_f32<M>),                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                       \
        hipLaunchKernelGGL((k_forest_margin_f32<M>), dim3(node_grid), dim3(kBlock), sh, s, dv, d_cur, d_tiles.p,    \
                           n_tiles, cur, N, chunk_d, nstride, hdr_off, masks.p, tile_left.p, abort_a);              \
    } while (0)
                switch (ds->metric) {
                case AH_EUCLIDEAN: AH_LAUNCH(AH_EUCLIDEAN); break;
                case AH_MANHATTAN: AH_LAUNCH(AH_MANHATTAN); break;
                case AH_COSINE: AH_LAUNCH(AH_COSINE); break;
                default: AH_LAUNCH(AH_DOT_PRODUCT); break;
                }
#undef AH_LAUNCH
                forest->stats.margin_mode_launches[MM_NODE]++;
            }
            AH_HIP(hipEventRecord(bc.ev_attempt[2 * attempt + 1], s));
            AH_DBG(s, "margin");
            hipLaunchKernelGGL(k_forest_decide, dim3((n_nodes + 255) / 256), dim3(256), 0, s, d_cur, n_nodes,
                               attempt < 3 ? d_small.p + attempt + 1 : nullptr, gate);
            AH_DBG(s, "decide");
        }
        hipLaunchKernelGGL(k_forest_random_sides, dim3(std::min<uint32_t>(n_tiles, kMaxBlocks)), dim3(64), 0, s, d_cur,
                           d_tiles.p, n_tiles, masks.p, tile_left.p);
        AH_DBG(s, "random_sides");
        hipLaunchKernelGGL(k_forest_tile_offsets, dim3(std::min<uint32_t>(n_nodes, kMaxBlocks)), dim3(64), 0, s, d_cur,
                           n_nodes, tile_left.p, tile_left_off.p);
        AH_DBG(s, "tile_offsets");
        hipLaunchKernelGGL(k_forest_scatter, dim3(tile_grid), dim3(kBlock), 0, s, d_cur, d_tiles.p, n_tiles, cur, nxt,
                           final_perm.p, N, masks.p, tile_left_off.p, split_after);
        AH_DBG(s, "scatter");
        // the next level, on the device (its node table lands in the buffer the level before this one used: the side-stream
        // copy of that table must be over)
        if (step >= 1) AH_HIP(hipStreamWaitEvent(s, bc.ev_copy[(step + 1) & 1], 0));
        // (a group's first level emits into the table the group before it may have finished in: that copy is one step old)
        if (grouped && step >= 1) AH_HIP(hipStreamWaitEvent(s, bc.ev_copy[step & 1], 0));
        const uint32_t n_blocks = (n_nodes + 255) / 256;
        hipLaunchKernelGGL(k_next_count, dim3(n_blocks), dim3(256), 0, s, d_cur, n_nodes, split_after, d_block_sums.p);
        hipLaunchKernelGGL(k_next_scan, dim3(1), dim3(1024), 0, s, d_block_sums.p, n_blocks, d_info, d_tree_first, n_trees,
                           d_small.p + 1);
        hipLaunchKernelGGL(k_next_emit, dim3(n_blocks), dim3(256), 0, s, d_cur, n_nodes, split_after, d_block_sums.p, d_next,
                           d_child.p, d_info, d_tree_first);
        AH_DBG(s, "next_level");
        AH_HIP(hipGetLastError());
        LevelInfo *hi = h_info[step & 1];
        AH_HIP(hipMemcpyAsync(hi, d_info, info_words * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipEventRecord(bc.ev_level, s));

        // while the level runs: room for the records of the two digests to come (the level before this one, now; this one,
        // under the next level), then digest the node table of the level before it
        {
            size_t want = n_recs + 2 * (size_t)n_nodes;
            for (const PendingTable &e : tables) want += 2 * (size_t)e.n_nodes;
            if (recs.size() < want) recs.resize(want);
        }
        const auto t_launched = std::chrono::steady_clock::now();
        // the group before this one is complete (its last level was waited for): its ids travel under this group's kernels
        if (pending_hand >= 0) {
            AH_TRY(hand_over_ids(group_tree[pending_hand], group_tree[pending_hand + 1]));
            pending_hand = -1;
        }
        AH_TRY(digest_under(info.pairs));
        const auto t_digested = std::chrono::steady_clock::now();
        lvl_status = wait_level();
        if (lvl_status != AH_OK) return lvl_status;
        const auto t_waited = std::chrono::steady_clock::now();
        t_prev_waited = t_waited;
        float attempt_ms[4] = {0.f, 0.f, 0.f, 0.f};
        for (int attempt = 0; attempt < 4; attempt++) {
            float m = 0.0f;
            if (hipEventElapsedTime(&m, bc.ev_attempt[2 * attempt], bc.ev_attempt[2 * attempt + 1]) == hipSuccess) {
                forest->stats.seconds_margin += m * 1e-3;
                attempt_ms[attempt] = m;
            }
        }
        if (timing >= 2)
            fprintf(stderr, "[ah] level %2u: %8u nodes %12llu pairs  mode %s  margin pass %.2f ms (+ retries %.2f) = %.3f ns per pair\n",
                    depth, n_nodes, (unsigned long long)info.pairs,
                    dense ? "dense-mfma" : lds_tc ? "rows-lds" : row_tc >= 2 ? "rows" : "node-major", attempt_ms[0],
                    attempt_ms[1] + attempt_ms[2] + attempt_ms[3], info.pairs ? attempt_ms[0] * 1e6 / (double)info.pairs : 0.0);
        if (timing >= 3) {
            auto ms_of = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[ah]          host: after the previous wait %.2f ms, launch %.2f ms, under the level (ids handed over, tables digested, node lists) %.2f ms, wait %.2f ms\n",
                    ms_tail_prev, ms_of(t_level_top, t_launched), ms_of(t_launched, t_digested), ms_of(t_digested, t_waited));
        }
        forest->stats.margin_launches += 4;
        // this level's node table follows on the side stream (the level is complete: no stream dependency needed)
        size_t table_off = 0;
        const size_t table_bytes = ((size_t)n_nodes * sizeof(FNode) + 4095) & ~(size_t)4095;
        while (!ring_alloc(table_bytes, &table_off)) AH_TRY(digest_front());  // (no room: the oldest tables first)
        AH_HIP(hipMemcpyAsync(ring_base + table_off, d_cur, n_nodes * sizeof(FNode), hipMemcpyDeviceToHost, bc.side));
        AH_HIP(hipEventRecord(bc.ev_copy[step & 1], bc.side));
        // (a millisecond at most — and it must not queue behind the gigabytes of normals the worker is about to request: the
        // digest of this table runs under the NEXT level, which at the bottom of the forest is a short one)
        if (n_nodes >= 65536) AH_HIP(hipEventSynchronize(bc.ev_copy[step & 1]));
        tables.push_back(PendingTable{depth, step, n_nodes, table_off, chunk_host_off, chunk_d,
                                      group_first_level ? (int64_t)group_n0 : -1, false, -1});
        // this level's normals are final: the worker copies them while the next level runs
        touch_normals[step & 1].join();  // never touch a page the worker may already have filled
        if (!sb) rb.push(forest->normals + chunk_host_off, chunk_d, chunk_bytes);  // (streaming: pushed by the level's digest)

        info = *hi;
        const uint32_t *tf = reinterpret_cast<const uint32_t *>(hi + 1);
        for (uint32_t t = 0; t <= n_trees; t++) tree_first[t] = tf[t];
        prev_rows = row_tc >= 2;
        std::swap(cur, nxt);
        std::swap(d_cur, d_next);
        if (group_first_level) {  // the group's first table was a slice of the level's: from here on it has two of its own
            d_next = group_spare;
            group_first_level = false;
        }
        depth++;
        step++;
    }
    return AH_OK;
    };
    auto run_tail_groups = [&]() -> int {
        if (!fork_now) return AH_OK;
        const uint32_t K = std::min(g_tail_groups, n_trees);
        group_tree.resize(K + 1);
        GroupCuts cuts{};
        // shrinking groups: group g + 1 computes while group g's ids and normals travel (~0.76 of the time its kernels took at
        // PCIe rate), and only the last group's are left after the last launch — so the last one is the smallest
        {
            const double ratio = (double)std::min<long long>(100, std::max<long long>(10, tun(TUN_TAIL_RATIO))) / 100.0;
            double total_w = 0.0, w = 1.0, cum = 0.0;
            for (uint32_t g = 0; g < K; g++, w *= ratio) total_w += w;
            group_tree[0] = 0;
            w = 1.0;
            for (uint32_t g = 1; g < K; g++, w *= ratio) {
                cum += w;
                const uint32_t at = (uint32_t)std::llround((double)n_trees * cum / total_w);
                group_tree[g] = std::min(std::max(at, group_tree[g - 1] + 1), n_trees - (K - g));  // at least one tree each
            }
            group_tree[K] = n_trees;
        }
        for (uint32_t g = 0; g <= K; g++) cuts.first[g] = g < K ? tree_first[group_tree[g]] : info.n_nodes;
        AH_TRY(d_nodes_c.ensure(max_nodes));
        AH_TRY(d_groups.ensure(32));
        AH_HIP(hipMemsetAsync(d_nodes_c.p, 0, max_nodes * sizeof(FNode), s));  // (zeros: see the tables above)
        hipLaunchKernelGGL(k_fork_groups, dim3(K), dim3(256), 0, s, d_cur, cuts, d_groups.p);
        GroupInfo gi[32];
        AH_HIP(hipMemcpyAsync(gi, d_groups.p, K * sizeof(GroupInfo), hipMemcpyDeviceToHost, s));
        AH_HIP(hipStreamSynchronize(s));
        // the records of the level's nodes: the children's list of the level before it, digested under the first group
        if (!tables.empty()) tables.back().fork_parent = true;
        else level_rec_full.swap(level_rec);
        FNode *const table = d_cur, *const other = d_next;
        uint32_t *const cur_f = cur, *const nxt_f = nxt;
        const uint32_t depth_f = depth;
        grouped = true;
        group_spare = d_nodes_c.p;
        forest->stats.tail_groups += K;
        if (timing)
            fprintf(stderr, "[ah] tail: level %u on (%u nodes, %llu pairs) in %u groups of trees\n", depth, info.n_nodes,
                    (unsigned long long)info.pairs, K);
        for (uint32_t g = 0; g < K; g++) {
            const uint32_t n0 = cuts.first[g], n1 = cuts.first[g + 1];
            info = LevelInfo{};
            info.n_nodes = n1 - n0;
            info.n_tiles = gi[g].n_tiles;
            info.n_fix = gi[g].n_fix;
            info.pairs = gi[g].pairs;
            d_cur = table + n0;
            d_next = other;
            cur = cur_f;
            nxt = nxt_f;
            depth = depth_f;
            prev_rows = false;
            group_first_level = true;
            group_n0 = n0;
            AH_TRY(run_levels());
            if (sb && (sb->target.sink_rc.load() || sb->target.too_big.load())) break;  // reported after the loop
            // its ids: handed over under the next group's first level (the last group's: after the loop); its node list (the
            // leaves' job of a streaming build): when its last table has been digested
            pending_hand = (int)g;
            if (!tables.empty()) tables.back().group_done = (int)g;
        }
        return AH_OK;
    };
    AH_TRY(run_levels());
    AH_TRY(run_tail_groups());
    AH_HIP(hipEventRecord(bc.ev_end, s));
    const auto t_loop_end = std::chrono::steady_clock::now();

    // Results come back with plain D2H copies straight into their final place — no host-side repacking:
    //   normals      one copy per level chunk (device record layout == caller-visible layout), already under way
    //   descendants  the final permutations themselves (rows -> item ids on device first)
    auto t_tail_prefault = t_loop_end, t_tail_sync = t_loop_end;
    {
        prefault.join();
        t_tail_prefault = std::chrono::steady_clock::now();
        if (!sb) forest->descendants_len = desc_base + M;
        // trees that are a single Descendants node never went through a scatter: their list is the input itself
        // (a tail in groups has no such tree, and its ids were converted group by group)
        for (uint32_t t = 0; t < n_trees && !fork_now; t++)
            if (recs[tree_root[t]].kind == AH_NODE_DESCENDANTS && tree_base[t + 1] > tree_base[t])
                AH_HIP(hipMemcpyAsync(final_perm.p + tree_base[t], perm_a.p + tree_base[t],
                                      (tree_base[t + 1] - tree_base[t]) * 4, hipMemcpyDeviceToDevice, s));
        if (!ds->identity_ids && M && !fork_now)
            hipLaunchKernelGGL(k_rows_to_ids, dim3(2048), dim3(256), 0, s, final_perm.p, M, ds->d_ids);
        ScreenCounters sc{};
        AH_HIP(hipMemcpyAsync(&sc, d_counters, sizeof sc, hipMemcpyDeviceToHost, s));
        AH_HIP(hipStreamSynchronize(s));
        t_tail_sync = std::chrono::steady_clock::now();
        forest->stats.screen_fallbacks += sc.fallbacks;
        forest->stats.screen_violations += sc.violations;
        forest->stats.screen8_pairs += sc.stage8_pairs;
        forest->stats.screen8_decided += sc.stage8_decided;
        forest->stats.screen8b_decided += sc.stage8b_decided;
        if (!sb && !fork_now) rb.push(forest->descendants + desc_base, final_perm.p, M * 4);  // lands while the host emits the node list
        if (fork_now && pending_hand >= 0) {  // (the last group's)
            AH_TRY(hand_over_ids(group_tree[pending_hand], group_tree[pending_hand + 1]));
            pending_hand = -1;
        }
    }
    // the last level's node table is digested only now: the 4 GB of item ids are already on their way
    AH_TRY(digest_pending());
    if (sb && !fork_now) AH_TRY(push_leaves_job(0, n_trees));  // (in groups: every group's job followed its last table)
    const auto t_levels = std::chrono::steady_clock::now();
    float ms = 0.0f;
    AH_HIP(hipEventElapsedTime(&ms, bc.ev_begin, bc.ev_end));
    forest->stats.seconds_device += ms * 1e-3;
    if (sb) {
        // no node list, no blobs: the roots, the counts, and the worker's verdict
        recs.resize(n_recs);
        uint64_t splits = 0, descs = 0;
        for (const HostRec &r : recs) (r.kind == AH_NODE_SPLIT ? splits : descs)++;
        forest->stats.split_nodes += splits;
        forest->stats.descendant_nodes += descs;
        for (uint32_t t = 0; t < n_trees; t++) sb->roots[first_tree + t] = id_base + tree_root[t];
        AH_REQUIRE((uint64_t)id_base + n_recs < 0xFFFFFFFFull, AH_ERR_INVALID_ARGUMENT, "more than 2^32 tree nodes in one build");
        sb->id_base = id_base + (uint32_t)n_recs;
        AH_REQUIRE(rb.drain() == hipSuccess, AH_ERR_DEVICE, "device -> host copy of the forest failed");
        AH_REQUIRE(!sb->target.too_big.load(), AH_ERR_INVALID_ARGUMENT,
                   "a Descendants node holds more item ids than the streaming buffer (split_after too large for ah_build_forest_stream)");
        AH_REQUIRE(!sb->target.sink_rc.load(), AH_ERR_CANCELLED, "node sink asked to stop (code %d)", sb->target.sink_rc.load());
        forest->stats.seconds_after_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop_end).count();
        if (timing)
            fprintf(stderr, "[ah] streamed batch of %u trees: %.3f s (device %.3f), %llu sink calls, %.2f GB handed over\n", n_trees,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_batch).count(), ms * 1e-3,
                    (unsigned long long)sb->target.batches, sb->target.bytes / 1e9);
        return AH_OK;
    }

    // the node list of the trees not emitted yet (all of them, unless the tail ran in groups)
    AH_TRY(emit_range(emit_tree, n_trees));
    AH_REQUIRE(emit_cursor - node_base == n_recs, AH_ERR_DEVICE, "forest build: %zu nodes emitted of %zu (internal error)",
               emit_cursor - node_base, n_recs);
    forest->stats.split_nodes += n_split.load();
    forest->stats.descendant_nodes += n_desc.load();
    const auto t_emitted = std::chrono::steady_clock::now();
    AH_REQUIRE(rb.drain() == hipSuccess, AH_ERR_DEVICE, "device -> host copy of the forest failed");
    forest->normals_len = normals_base + normals_bytes;
    touch_normals[0].join();  // (the last level started the commit of a level that never came)
    touch_normals[1].join();
    // (the head-room of the blob stays mapped: untouched pages cost nothing, and the pool hands the whole blob to the next build)
    forest->stats.host_blob_recycled += (forest->normals_blob.recycled ? 1u : 0u) + (forest->desc_blob.recycled ? 1u : 0u);
    forest->stats.seconds_after_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop_end).count();
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        fprintf(stderr, "[ah] batch of %u trees: levels %.3f s (device %.3f; after the last launch: ids' pages %.3f, stream %.3f, last "
                        "digest %.3f), emit %.3f s, read-back still in flight after it %.3f s (%.2f GB normals%s, %.2f GB ids%s)\n",
                n_trees, sec(t_batch, t_levels), ms * 1e-3, sec(t_loop_end, t_tail_prefault), sec(t_tail_prefault, t_tail_sync),
                sec(t_tail_sync, t_levels), sec(t_levels, t_emitted), sec(t_emitted, t_end), normals_bytes / 1e9,
                forest->normals_blob.committed ? ", recycled pages" : "", M * 4 / 1e9,
                forest->desc_blob.committed ? ", recycled pages" : "");
    }
    return AH_OK;
}

extern "C" {

static int build_forest_impl(ah_dataset *ds, const ah_build_options *options, const uint32_t *subset_ids,
                             const uint64_t *subset_offsets, ah_forest **out, StreamBuild *sb = nullptr) {
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(ds && options, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE(ds->finalized, AH_ERR_NOT_FINALIZED, "dataset not finalized (call ah_dataset_finalize)");
    AH_REQUIRE(options->n_trees == 0 || options->tree_seeds, AH_ERR_INVALID_ARGUMENT, "tree_seeds is NULL");
    {
        const uint32_t m = options->margin_mode & 0xFFFu;
        AH_REQUIRE((options->margin_mode & ~(0xFFFu | AH_MARGIN_EXACT_ONLY)) == 0 &&
                       (m == AH_MARGIN_AUTO || m == AH_MARGIN_NODE_MAJOR || m == AH_MARGIN_ROWS_2 || m == AH_MARGIN_ROWS_4 ||
                        m == AH_MARGIN_ROWS_8 || m == AH_MARGIN_ROWS_16 || m == AH_MARGIN_ROWS_LDS_8 || m == AH_MARGIN_ROWS_LDS_16 ||
                        m == AH_MARGIN_DENSE_MFMA),
                   AH_ERR_INVALID_ARGUMENT, "unknown margin_mode 0x%x", options->margin_mode);
    }
    AH_REQUIRE(ds->metric != AH_DOT_PRODUCT || ds->dot_preprocessed, AH_ERR_NEED_PREPROCESS,
               "DotProduct needs ah_preprocess_dot before the build (src/writer.rs:964-976)");
    AH_HIP(hipSetDevice(ds->device));
    {
        const auto tw = std::chrono::steady_clock::now();
        ds->join_reserve();  // (ah_dataset_reserve_build still filling the cache)
        ds->reserve_wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    }
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t split_after = options->split_after ? options->split_after : ds->dims;  // src/writer.rs:474-477
    ah_forest *forest = new (std::nothrow) ah_forest();
    AH_REQUIRE(forest, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
    forest->normal_vector_offset = 0;
    forest->normal_header_offset = ds->row_bytes();
    forest->normal_stride = normal_record_stride(ds);
    int st = AH_OK;
    if (!subset_ids && ds->n <= split_after) {
        // fit_in_descendant at the root (src/writer.rs:1183-1188): every tree is one Descendants node
        const uint64_t total = sb ? ds->n : ds->n * options->n_trees;  // (streaming: one copy of the id list serves every tree)
        if (!host_blob_reserve(forest->desc_blob, total * 4, 0)) {
            delete forest;
            set_error("host allocation failed");
            return AH_ERR_OUT_OF_MEMORY;
        }
        forest->descendants = reinterpret_cast<uint32_t *>(forest->desc_blob.p);
        forest->descendants_len = total;
        if (sb) {
            for (uint64_t i = 0; i < ds->n; i++) forest->descendants[i] = ds->identity_ids ? (uint32_t)i : ds->h_ids[i];
            for (uint32_t t = 0; t < options->n_trees && st == AH_OK; t++) {
                ah_stream_node sn{};
                sn.id = t;
                sn.tree = t;
                sn.kind = AH_NODE_DESCENDANTS;
                sn.count = (uint32_t)ds->n;
                ah_node_batch batch{};
                batch.kind = AH_NODE_DESCENDANTS;
                batch.n_nodes = 1;
                batch.nodes = &sn;
                batch.payload = reinterpret_cast<const uint8_t *>(forest->descendants);
                batch.payload_len = ds->n * 4;
                batch.normal_stride = forest->normal_stride;
                batch.normal_header_offset = forest->normal_header_offset;
                const int rc = sb->target.sink(sb->target.user, &batch);
                if (rc != 0) {
                    set_error("node sink asked to stop (code %d)", rc);
                    st = AH_ERR_CANCELLED;
                }
                sb->roots[t] = t;
                forest->stats.descendant_nodes++;
            }
        }
        for (uint32_t t = 0; t < options->n_trees && !sb; t++) {
            for (uint64_t i = 0; i < ds->n; i++)
                forest->descendants[t * ds->n + i] = ds->identity_ids ? (uint32_t)i : ds->h_ids[i];
            ah_node nd{};
            nd.kind = AH_NODE_DESCENDANTS;
            nd.tree = t;
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
            // the binary16 shadow of the rows is made (once per dataset) before the batch is sized against free memory
            if (!(options->margin_mode & AH_MARGIN_EXACT_ONLY) && tun(TUN_SCREEN)) (void)ensure_screen(ds, lease.c->stream, true, true);
            // Trees in flight: bounded by HBM (per item and tree: 3 permutations + node index + side byte + masks = 18
            // bytes, plus the normals of all levels and their shadow) or by the caller.
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            free_b += dev_cache_idle_bytes(ds->device);  // idle blocks of the caching allocator are ours to use
            uint32_t batch = options->n_trees;
            if (!subset_ids) {
                const uint64_t per_tree =
                    ds->n * 18 + ((ds->n / ((uint64_t)split_after + 1)) + 2) * 3 * (ds->row_bytes() + 192) + (1u << 20);
                uint64_t fit = (uint64_t)(free_b * 0.8) / per_tree;
                if (fit < 1) fit = 1;
                batch = (uint32_t)std::min<uint64_t>(fit, options->n_trees);
            }
            if (options->max_trees_in_flight) batch = std::min(batch, options->max_trees_in_flight);
            try {
                for (uint32_t first = 0; first < options->n_trees && st == AH_OK; first += batch) {
                    const auto tb = std::chrono::steady_clock::now();
                    st = build_batch(ds, options, first, std::min(batch, options->n_trees - first), split_after, forest,
                                     lease.c, subset_ids, subset_offsets, sb);
                    if (tun(TUN_TIMING))
                        fprintf(stderr, "[ah] build: %.1f ms before the batch, the batch and the teardown of its buffers %.1f ms\n",
                                std::chrono::duration<double, std::milli>(tb - t0).count(),
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count());
                }
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
    {  // the reserve helper's figures go to the first build that joined it
        std::lock_guard<std::mutex> lk(ds->mu);
        forest->stats.seconds_reserve = ds->reserve_seconds;
        forest->stats.seconds_reserve_wait = ds->reserve_wait_seconds;
        ds->reserve_seconds = ds->reserve_wait_seconds = 0.0;
    }
    *out = forest;
    return AH_OK;
}

// ah_dataset_reserve_build: the device memory the first build of this dataset will ask for — the binary16 / int8 copies of the
// rows and the scratch of an `n_trees`-tree batch — obtained NOW, on a helper thread, and parked in the caching allocator.
// Meant to be called right after ah_dataset_create, while the records are still being staged over PCIe: fresh HBM costs the
// driver 20+ ms per GB on a box whose memory was recently released (r04: 59 GB of fresh blocks put 1.3 s in front of a cold
// 10M x 100-tree build's first launch and stretched its kernels from 1.33 to 1.8 s); under the 0.6 s of staging it is free.
// Sizes mirror build_batch / ensure_screen; a mismatch only costs the cache hit.
int ah_dataset_reserve_build(ah_dataset *ds, uint32_t n_trees, uint32_t split_after_opt) {
    AH_GUARDED("ah_dataset_reserve_build")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    ds->join_reserve();
    const uint64_t N = std::max<uint64_t>(ds->n, ds->capacity);
    if (N == 0 || n_trees == 0 || tun(TUN_DEVICE_CACHE_MB) <= 0) return AH_OK;
    const uint32_t split_after = split_after_opt ? split_after_opt : ds->dims;
    std::vector<size_t> sizes;
    const bool f32 = !metric_is_bq(ds->metric) && ds->dims >= 32;
    const bool want_screen = f32 && tun(TUN_SCREEN) != 0 && !ds->d_rows_h16;
    const uint32_t hpitch = (ds->dims + 63u) & ~63u, pitch8 = (ds->dims + 127u) & ~127u;
    if (want_screen) {  // ensure_screen / ensure_screen8
        sizes.push_back(N * (size_t)hpitch * 2);
        sizes.push_back(N * sizeof(float4));
        if (ds->metric != AH_DOT_PRODUCT && tun(TUN_SCREEN8) != 0) {
            if (tun(TUN_SCREEN8_LO) != 0) sizes.push_back(N * (size_t)pitch8);
            sizes.push_back(N * (size_t)pitch8);
            sizes.push_back(N * sizeof(float));
        }
    }
    {  // build_batch, one batch of n_trees full-dataset trees
        const uint64_t M = (uint64_t)n_trees * N;
        const uint64_t max_nodes = M / ((uint64_t)split_after + 1) + n_trees, max_tiles = M / kTile + n_trees + max_nodes;
        auto buf = [&](uint64_t n, size_t elem) { sizes.push_back((size_t)std::max<uint64_t>(n, 1024) * elem); };
        buf(M, 4); buf(M, 4); buf(M, 4);                                 // the three permutations
        buf(max_nodes, sizeof(FNode)); buf(max_nodes, sizeof(FNode));
        buf(max_tiles, sizeof(FTile));
        buf(max_tiles * 32, 8);                                          // side masks
        buf(max_tiles, 4); buf(max_tiles, 4);
        buf(2 * max_nodes, 4);
        buf(max_nodes / 256 + 2, sizeof(NextCounts));
        if (f32 && n_trees >= 2) {
            buf((uint64_t)n_trees * N + 4, 4);                           // node_of
            buf((uint64_t)n_trees * N + 1024 + 16, 1);                   // side bytes
            if (tun(TUN_MASK_BITS) > 0 || (tun(TUN_MASK_BITS) < 0 && N >= (1u << 20)))
                buf(((uint64_t)n_trees * N + 31) / 32 + 4, 4);           // ... and bits
        }
        const uint64_t nstride = normal_record_stride(ds);
        sizes.push_back((size_t)std::max<uint64_t>(32ull << 20, std::min<uint64_t>(2 * max_nodes * nstride, 16ull << 30)));
        if (f32 && tun(TUN_SCREEN) != 0) {
            const uint64_t hstride = ((uint64_t)hpitch * 2 + 16 + 127) & ~(uint64_t)127;
            const uint64_t stride8 = (2 * (uint64_t)pitch8 + sizeof(NormalStats8) + 127) & ~(uint64_t)127;
            const size_t sh = (size_t)std::max<uint64_t>(16ull << 20, std::min<uint64_t>(max_nodes * hstride, 8ull << 30));
            sizes.push_back(sh);
            sizes.push_back(sh);  // (a 100-tree 10M build takes a second block for its deepest levels)
            if (ds->metric != AH_DOT_PRODUCT && tun(TUN_SCREEN8) != 0)
                sizes.push_back((size_t)std::max<uint64_t>(16ull << 20, std::min<uint64_t>(max_nodes * stride8, 4ull << 30)));
        }
    }
    // ... and the pinned host memory of that batch (node tables, level info, the read-back worker's double buffer)
    size_t pinned_bytes = 0;
    {
        const uint64_t M = (uint64_t)n_trees * N;
        const uint64_t max_nodes = M / ((uint64_t)split_after + 1) + n_trees;
        const size_t info_words = (sizeof(LevelInfo) + ((size_t)n_trees + 1) * 4 + 3) / 4;
        const size_t pin_info = (info_words * 4 + 255) & ~(size_t)255;
        const size_t pin_nodes = (max_nodes * sizeof(FNode) + 4095) & ~(size_t)4095;
        const size_t pin_head = (3 * pin_info + 256 + 4095) & ~(size_t)4095;
        pinned_bytes = pin_head + 2 * pin_nodes + ((size_t)std::min<long long>(4096, std::max<long long>(2, tun(TUN_READBACK_MB))) << 20);
    }
    const int device = ds->device;
    double *const took = &ds->reserve_seconds;  // (the handle outlives the helper: every build and the destroy join it)
    try {
        std::thread helper([device, sizes, took, pinned_bytes] {
            const auto th = std::chrono::steady_clock::now();
            if (hipSetDevice(device) != hipSuccess) return;
            pinned_spare_fill(device, pinned_bytes);
            std::vector<void *> got;
            for (size_t b : sizes) {
                void *p = nullptr;
                if (dev_malloc(&p, b) != hipSuccess) {  // no room: the build will see for itself
                    (void)hipGetLastError();
                    break;
                }
                got.push_back(p);
            }
            for (void *p : got) (void)dev_free_unused(p);
            *took += std::chrono::duration<double>(std::chrono::steady_clock::now() - th).count();
        });
        std::thread stale;  // (a helper another thread started since the join above: wait for that one too)
        {
            std::lock_guard<std::mutex> lk(ds->mu);
            stale = std::move(ds->reserve_thread);
            ds->reserve_thread = std::move(helper);
        }
        if (stale.joinable()) stale.join();
    } catch (...) {  // no thread: the first build allocates as before
    }
    return AH_OK;
    AH_GUARDED_END
}

int ah_build_forest(ah_dataset *ds, const ah_build_options *options, ah_forest **out) {
    AH_GUARDED("ah_build_forest")
    return build_forest_impl(ds, options, nullptr, nullptr, out);
    AH_GUARDED_END
}

// The same build with the node sink inside it (include/arroy_hip.h, "Streaming build"): nothing is materialised.
int ah_build_forest_stream(ah_dataset *ds, const ah_build_options *options, ah_node_batch_fn sink, void *user,
                           uint32_t *out_roots, ah_build_stats *out_stats) {
    AH_GUARDED("ah_build_forest_stream")
    AH_REQUIRE(sink, AH_ERR_INVALID_ARGUMENT, "sink is NULL");
    AH_REQUIRE(options && (out_roots || options->n_trees == 0), AH_ERR_INVALID_ARGUMENT, "NULL argument");
    StreamBuild sb;
    sb.target.sink = sink;
    sb.target.user = user;
    sb.roots = out_roots;
    ah_forest *forest = nullptr;
    const int st = build_forest_impl(ds, options, nullptr, nullptr, &forest, &sb);
    if (st == AH_OK && out_stats) *out_stats = forest->stats;
    delete forest;
    return st;
    AH_GUARDED_END
}

// `incremental_index_large_descendant` (src/writer.rs:660-739) for many descendants at once: tree t of the result is
// `make_tree_in_file` over the ascending id list item_ids[offsets[t] .. offsets[t+1]).
int ah_build_subtrees(ah_dataset *ds, const ah_build_options *options, const uint32_t *item_ids, const uint64_t *offsets,
                      ah_forest **out) {
    AH_GUARDED("ah_build_subtrees")
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
    AH_GUARDED_END
}

int ah_forest_view_get(const ah_forest *forest, ah_forest_view *out) {
    AH_GUARDED("ah_forest_view_get")
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
    AH_GUARDED_END
}

// ---- content digest ---------------------------------------------------------------------------------------------------
namespace {
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
// 64-bit hash of a byte range: four multiply-rotate lanes over 32-byte blocks (a plain word-at-a-time chain would bound the
// digest of a 10 GB forest by the multiplier's latency), folded with the splitmix finaliser of arroy_hip_policy.h
uint64_t hash_bytes(const void *data, size_t len, uint64_t seed) {
    const uint8_t *p = reinterpret_cast<const uint8_t *>(data);
    uint64_t h0 = seed ^ 0x9E3779B97F4A7C15ull, h1 = seed + 0xC2B2AE3D27D4EB4Full, h2 = ~seed, h3 = seed * 0x165667B19E3779F9ull + len;
    size_t i = 0;
    for (; i + 32 <= len; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        h0 = rotl64((h0 ^ w[0]) * 0x9FB21C651E98DF25ull, 29);
        h1 = rotl64((h1 ^ w[1]) * 0xD6E8FEB86659FD93ull, 31);
        h2 = rotl64((h2 ^ w[2]) * 0xBF58476D1CE4E5B9ull, 27);
        h3 = rotl64((h3 ^ w[3]) * 0x94D049BB133111EBull, 33);
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    if (i < len) memcpy(tail, p + i, len - i);
    h0 = (h0 ^ tail[0]) * 0x9FB21C651E98DF25ull;
    h1 = (h1 ^ tail[1]) * 0xD6E8FEB86659FD93ull;
    h2 = (h2 ^ tail[2]) * 0xBF58476D1CE4E5B9ull;
    h3 = (h3 ^ tail[3]) * 0x94D049BB133111EBull;
    return ah_mix64(ah_mix64(h0 ^ rotl64(h1, 17)) + ah_mix64(h2 ^ rotl64(h3, 41)) + len);
}
}  // namespace

static int forest_digest_impl(const ah_forest *forest, const uint64_t *tree_keys, uint64_t *out_per_tree, uint64_t *out_total);
int ah_forest_digest(const ah_forest *forest, uint64_t *out_per_tree, uint64_t *out_total) {
    AH_GUARDED("ah_forest_digest")
    return forest_digest_impl(forest, nullptr, out_per_tree, out_total);
    AH_GUARDED_END
}
int ah_forest_digest_keyed(const ah_forest *forest, const uint64_t *tree_keys, uint64_t *out_per_tree) {
    AH_GUARDED("ah_forest_digest_keyed")
    AH_REQUIRE(tree_keys, AH_ERR_INVALID_ARGUMENT, "tree_keys is NULL");
    return forest_digest_impl(forest, tree_keys, out_per_tree, nullptr);
    AH_GUARDED_END
}
static int forest_digest_impl(const ah_forest *forest, const uint64_t *tree_keys, uint64_t *out_per_tree, uint64_t *out_total) {
    AH_REQUIRE(forest && (out_per_tree || out_total), AH_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint32_t n_trees = (uint32_t)forest->roots.size();
    const size_t n_nodes = forest->nodes.size();
    // the nodes of a tree are contiguous and the trees ascend (build_batch emits tree by tree)
    std::vector<size_t> first(n_trees + 1, n_nodes);
    for (size_t i = n_nodes; i-- > 0;) {
        const uint32_t t = forest->nodes[i].tree;
        AH_REQUIRE(t < n_trees, AH_ERR_INVALID_ARGUMENT, "forest digest: node %zu names tree %u of %u", i, t, n_trees);
        first[t] = i;
    }
    for (uint32_t t = n_trees; t-- > 0;) first[t] = std::min(first[t], first[t + 1]);
    // header bytes of a record: up to the 16-byte slot, but only the metric's own floats are content (the rest is zero
    // padding either way); vector bytes: everything before the header
    const size_t vec_off = forest->normal_vector_offset, hdr_off = forest->normal_header_offset;
    const size_t vec_len = hdr_off > vec_off ? hdr_off - vec_off : 0;
    const size_t hdr_len = std::min<size_t>(8, forest->normal_stride - hdr_off);
    std::vector<uint64_t> per(n_trees, 0);
    std::atomic<uint32_t> next{0};
    auto work = [&](unsigned) {
        for (;;) {
            const uint32_t t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= n_trees) break;
            // the tree's identity is part of the content (seeds are per tree): its index in this forest, or the caller's key
            uint64_t h = ah_mix64(0x61727279ull + (tree_keys ? tree_keys[t] : (uint64_t)t));
            const size_t base = first[t];
            for (size_t i = first[t]; i < first[t + 1]; i++) {
                const ah_node &nd = forest->nodes[i];
                if (nd.tree != t) continue;
                uint64_t f[4] = {(uint64_t)nd.kind | (uint64_t)nd.has_normal << 8 | (uint64_t)nd.depth << 32, nd.count, 0, 0};
                if (nd.kind == AH_NODE_SPLIT) {
                    f[2] = nd.left - base;  // children by their position inside the tree
                    f[3] = nd.right - base;
                }
                h = ah_mix64(h ^ hash_bytes(f, sizeof f, i - base));
                if (nd.kind == AH_NODE_SPLIT) {
                    if (nd.has_normal) {
                        const uint8_t *rec = forest->normals + nd.offset;
                        h = ah_mix64(h ^ hash_bytes(rec + vec_off, vec_len, 1));
                        h = ah_mix64(h ^ hash_bytes(rec + hdr_off, hdr_len, 2));
                    }
                } else {
                    h = ah_mix64(h ^ hash_bytes(forest->descendants + nd.offset, (size_t)nd.count * 4, 3));
                }
            }
            per[t] = h;
        }
    };
    parallel_run(n_nodes < 100000 ? 1u : std::min({n_trees, (uint32_t)std::max<long long>(1, tun(TUN_HOST_THREADS)),
                                                    std::max(1u, std::thread::hardware_concurrency())}),
                 work);
    uint64_t total = ah_mix64(n_trees);
    for (uint32_t t = 0; t < n_trees; t++) {
        total = ah_mix64(total ^ per[t]);
        if (out_per_tree) out_per_tree[t] = per[t];
    }
    if (out_total) *out_total = total;
    return AH_OK;
}

// Test aid: the block -> work-item maps of the build's launches, run on the device (see include/arroy_hip.h).
int ah_debug_launch_coverage(int device, int kind, uint64_t n_rows, uint32_t dims, uint32_t a, uint32_t b, uint32_t *out_counts,
                             uint64_t out_len) {
    AH_GUARDED("ah_debug_launch_coverage")
    AH_REQUIRE(out_counts && n_rows && dims, AH_ERR_INVALID_ARGUMENT, "NULL / empty argument");
    AH_HIP(hipSetDevice(device));
    const uint32_t hpitch = (dims + 63u) & ~63u;
    uint64_t need = 0;
    DensePlan dp{};
    if (kind == 0) {
        AH_REQUIRE((a == 2 || a == 4 || a == 8 || a == 16) && b >= 1, AH_ERR_INVALID_ARGUMENT, "kind 0: a = trees per group, b = groups");
        need = (uint64_t)b * n_rows;
    } else if (kind == 1) {
        AH_REQUIRE(a >= 1, AH_ERR_INVALID_ARGUMENT, "kind 1: a = columns");
        dp = dense_plan(n_rows, a);
        AH_REQUIRE(dp.grid < 0x7FFFFFFFull, AH_ERR_INVALID_ARGUMENT, "too many tiles for one launch");
        need = (uint64_t)dp.n_row_tiles * dp.n_col_tiles;
    } else if (kind == 2) {
        AH_REQUIRE(a >= 1, AH_ERR_INVALID_ARGUMENT, "kind 2: a = trees");
        need = (uint64_t)a * ((n_rows + 1023) >> 10);
    } else {
        AH_REQUIRE(false, AH_ERR_INVALID_ARGUMENT, "unknown coverage kind %d", kind);
    }
    AH_REQUIRE(out_len >= need, AH_ERR_INVALID_ARGUMENT, "out_counts holds %llu counters, %llu needed", (unsigned long long)out_len,
               (unsigned long long)need);
    DevMem d;
    AH_HIP(dev_malloc(&d.p, need * 4));
    AH_HIP(hipMemset(d.p, 0, need * 4));
    if (kind == 0) {
        RowsPlan plan;
        // normals of a group as the build would size them for a level with 64 nodes per tree (only the nt policy reads it)
        const uint64_t hstride = ((uint64_t)hpitch * 2 + 16 + 127) & ~(uint64_t)127;
        AH_TRY(plan_rows_launches(n_rows, hpitch, a, b, false, (uint64_t)a * 64 * hstride, &plan));
        for (const auto &l : plan.launches) {
            plan.sch.chunk0 = l.first;
            hipLaunchKernelGGL(k_rows_schedule_coverage, dim3(l.second), dim3(plan.threads), 0, 0, plan.sch, n_rows, d.as<uint32_t>());
        }
    } else if (kind == 1) {
        hipLaunchKernelGGL(k_dense_coverage, dim3((unsigned)dp.grid), dim3(64), 0, 0, dp.group, dp.n_col_tiles, dp.n_row_tiles, d.as<uint32_t>());
    } else {
        hipLaunchKernelGGL(k_exact_coverage, dim3(exact_pairs_grid(n_rows, a)), dim3(256), 0, 0, n_rows, a, d.as<uint32_t>());
    }
    AH_HIP(hipGetLastError());
    AH_HIP(hipDeviceSynchronize());
    AH_HIP(hipMemcpy(out_counts, d.p, need * 4, hipMemcpyDeviceToHost));
    return AH_OK;
    AH_GUARDED_END
}

int ah_debug_dense_tiles(uint64_t n_rows, uint32_t n_cols, uint32_t *out_tile_rows, uint32_t *out_tile_cols) {
    AH_GUARDED("ah_debug_dense_tiles")
    AH_REQUIRE(n_rows && n_cols && out_tile_rows && out_tile_cols, AH_ERR_INVALID_ARGUMENT, "NULL / empty argument");
    const DensePlan dp = dense_plan(n_rows, n_cols);
    *out_tile_rows = dp.tile_rows;
    *out_tile_cols = dp.tile_cols;
    return AH_OK;
    AH_GUARDED_END
}

int ah_forest_stats(const ah_forest *forest, ah_build_stats *out) {
    AH_GUARDED("ah_forest_stats")
    AH_REQUIRE(forest && out, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = forest->stats;
    return AH_OK;
    AH_GUARDED_END
}

// payload: SPLIT -> the normal record (vector at normal_vector_offset, header at normal_header_offset) or
// NULL for `normal: None`; DESCENDANTS -> u32 item ids.
int ah_forest_visit(const ah_forest *forest, ah_node_sink_fn sink, void *user) {
    AH_GUARDED("ah_forest_visit")
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
    AH_GUARDED_END
}

int ah_forest_destroy(ah_forest *forest) {
    AH_GUARDED("ah_forest_destroy")
    NoFailScope no_fail;
    delete forest;
    return AH_OK;
    AH_GUARDED_END
}

int ah_device_cache_trim(int device, uint64_t *out_bytes) {
    AH_GUARDED("ah_device_cache_trim")
    const size_t was = dev_cache_trim(device);
    if (out_bytes) *out_bytes = was;
    return AH_OK;
    AH_GUARDED_END
}

int ah_device_cache_stats(int device, uint64_t *out_live_bytes, uint64_t *out_idle_bytes) {
    AH_GUARDED("ah_device_cache_stats")
    if (out_live_bytes) *out_live_bytes = dev_cache_live_bytes(device);
    if (out_idle_bytes) {
        uint64_t idle = 0;
        if (device >= 0) {
            idle = dev_cache_idle_bytes(device);
        } else {
            int n_dev = 0;
            if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
            for (int d = 0; d < n_dev; d++) idle += dev_cache_idle_bytes(d);
        }
        *out_idle_bytes = idle;
    }
    return AH_OK;
    AH_GUARDED_END
}

int ah_host_cache_trim(uint64_t *out_bytes) {
    AH_GUARDED("ah_host_cache_trim")
    const size_t was = host_pool().trim();
    if (out_bytes) *out_bytes = was;
    return AH_OK;
    AH_GUARDED_END
}

// Benchmark harness only: the policy header's generator on the host cores (the rows ah_dataset_fill_synthetic makes in HBM).
int ah_synth_rows_host(uint64_t seed, int distribution, uint64_t first_item, uint64_t n, uint32_t dims, float *out) {
    AH_GUARDED("ah_synth_rows_host")
    AH_REQUIRE(out || n == 0, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    AH_REQUIRE(distribution >= AH_SYNTH_UNIFORM_01 && distribution <= AH_SYNTH_LAST, AH_ERR_INVALID_ARGUMENT,
               "unknown distribution %d", distribution);
    // the per-dataset part of the structured distributions (cluster centres / factor loadings), computed once
    std::vector<int32_t> table(ah_synth_table_len(dims, distribution));
    if (!table.empty()) ah_synth_table_fill(seed, dims, distribution, table.data());
    const int32_t *tab = table.empty() ? nullptr : table.data();
    const unsigned n_threads = (unsigned)std::min<uint64_t>(std::max(1u, std::thread::hardware_concurrency()), std::max<uint64_t>(1, n / 1024));
    parallel_run(std::min(n_threads, 64u), [=](unsigned t) {
        const unsigned parts = std::min(n_threads, 64u);
        const uint64_t lo = n * t / parts, hi = n * (t + 1) / parts;
        for (uint64_t i = lo; i < hi; i++) ah_synth_row(seed, first_item + i, dims, distribution, tab, out + i * dims);
    });
    return AH_OK;
    AH_GUARDED_END
}

}  // extern "C"
