// distance.hip — batched distance kernels, query preparation, top-k and dataset helper kernels.
//
// Replaces the re-rank loop of arroy's search (src/reader.rs:381-399) and the per-pair SIMD kernels it
// calls (src/spaces/simple_avx.rs, simple_sse.rs, simple.rs).  All kernels are HBM-bound vector
// contractions (0.5 flop/byte): no MFMA.  See device_math.h for the lane mapping that makes every f32
// result bit-identical to the reference's AVX+FMA tier.
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace ah {

static constexpr int kBlock = 256;       // 4 waves = 32 octets
static constexpr int kMaxBlocks = 32768;  // one tile per wave up to ~1M f32 rows / 8M 1-bit rows (measured: 1-bit scan 90 -> 80 us vs a 2048-block persistent grid), grid-stride beyond that
// AH_SCAN_BLOCKS overrides the grid cap of the grid-stride kernels (tuning experiments only)
#define g_max_blocks ((int)(tun(TUN_SCAN_BLOCKS) > 0 ? tun(TUN_SCAN_BLOCKS) : kMaxBlocks))
// AH_MANHATTAN_ROWS=0: the octet kernel for Manhattan too (A/B switch)
#define g_manhattan_rows (tun(TUN_MANHATTAN_ROWS) != 0)

static inline unsigned grid_for(uint64_t work_items, int items_per_block) {
    uint64_t b = (work_items + items_per_block - 1) / items_per_block;
    if (b < 1) b = 1;
    if (b > (uint64_t)g_max_blocks) b = g_max_blocks;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// Query preparation: `QueryBuilder::by_vector` (src/reader.rs:64-75) = codec + D::new_header.
// One 64-thread block.  qvec gets `pitch` elements (zero padded).
// ------------------------------------------------------------------------------------------------
__global__ void k_prepare_query(DataView dv, const float *__restrict__ q_f32, void *qvec, float *qhdr) {
    const uint32_t t = threadIdx.x;
    if (!metric_is_bq_dev(dv.metric)) {
        float *dst = reinterpret_cast<float *>(qvec);
        for (uint32_t i = t; i < dv.pitch; i += blockDim.x) dst[i] = i < dv.dims ? q_f32[i] : 0.0f;
        __syncthreads();
        if (t < 8) {
            float hdr0 = 0.0f;
            if (dv.metric == AH_COSINE) {  // cosine.rs:39-41: norm = sqrt(dot(v,v))
                float d = octet_reduce_any<OP_DOT>(dst, dst, dv.dims, t);
                hdr0 = f_sqrt(d);
            }
            if (t == 0) {
                qhdr[0] = hdr0;
                qhdr[1] = 0.0f;
            }
        }
    } else {
        uint64_t *dst = reinterpret_cast<uint64_t *>(qvec);
        for (uint32_t w = t; w < dv.pitch; w += blockDim.x) {
            uint64_t word = 0;
            if (w < dv.words) {  // binary_quantized.rs:80-91: bit i = is_sign_positive(x[64w+i])
                for (uint32_t i = 0; i < 64; i++) {
                    uint32_t e = 64 * w + i;
                    if (e < dv.dims) word |= (uint64_t)((__float_as_uint(q_f32[e]) >> 31) == 0u) << i;
                }
            }
            dst[w] = word;
        }
        if (t == 0) {
            // bq_cosine.rs:45-47: norm = sqrt(bqdot(v,v)) = sqrt(64*words); others: bias 0
            qhdr[0] = dv.metric == AH_BQ_COSINE ? f_sqrt((float)(int32_t)(64u * dv.words)) : 0.0f;
            qhdr[1] = 0.0f;
        }
    }
}

__global__ void k_load_item_as_query(DataView dv, uint32_t row, void *qvec, float *qhdr) {
    const uint32_t t = threadIdx.x;
    if (!metric_is_bq_dev(dv.metric)) {
        float *dst = reinterpret_cast<float *>(qvec);
        const float *src = dv.rows_f32 + (uint64_t)row * dv.pitch;
        for (uint32_t i = t; i < dv.pitch; i += blockDim.x) dst[i] = src[i];
    } else {
        uint64_t *dst = reinterpret_cast<uint64_t *>(qvec);
        const uint64_t *src = dv.rows_bq + (uint64_t)row * dv.pitch;
        for (uint32_t i = t; i < dv.pitch; i += blockDim.x) dst[i] = src[i];
    }
    if (t == 0) {
        const uint32_t hf = dv.metric == AH_DOT_PRODUCT ? 2u : 1u;
        qhdr[0] = dv.headers[(uint64_t)row * hf];
        qhdr[1] = hf == 2 ? dv.headers[(uint64_t)row * hf + 1] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// f32 distance scan / gather, dims >= 32.  One octet per row, grid-stride over rows.
//   METRIC in {EUCLIDEAN, MANHATTAN, COSINE, DOT_PRODUCT};  GATHER: rows addressed through item ids.
// Algorithmic traffic per distance: 4*dims (+4 header for cosine) read + 4 written (+4 id if GATHER).
// ------------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float f32_epilogue(float r, const float *qhdr, const DataView &dv, uint64_t row) {
    if (METRIC == AH_COSINE) return cosine_from_dot(r, qhdr[0], dv.headers[row]);
    if (METRIC == AH_DOT_PRODUCT) return -r;  // dot_product.rs:52-56
    return r;                                 // euclidean.rs:45-47 / manhattan.rs:44-46
}

template <int METRIC, bool GATHER>
__global__ __launch_bounds__(kBlock) void k_distances_f32(DataView dv, const float *__restrict__ qvec,
                                                          const float *__restrict__ qhdr,
                                                          const uint32_t *__restrict__ ids, uint64_t n,
                                                          float *__restrict__ out, uint32_t *err) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    extern __shared__ float4 s_q4[];
    const uint32_t nq4 = dv.pitch >> 2;
    for (uint32_t i = threadIdx.x; i < nq4; i += blockDim.x) s_q4[i] = reinterpret_cast<const float4 *>(qvec)[i];
    __shared__ float s_hdr[2];
    if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdr[threadIdx.x];
    __syncthreads();
    const float *s_q = reinterpret_cast<const float *>(s_q4);

    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += n_octets) {
        uint64_t row = i;
        if (GATHER) {
            row = row_of_id(dv, ids[i]);
            if (row == ~0ull) {  // Error::MissingKey (src/reader.rs:383-384)
                if (j == 0) {
                    atomicOr(err, 1u);
                    out[i] = __uint_as_float(0x7FC00000u);
                }
                continue;
            }
            if (i > 0 && ids[i] <= ids[i - 1] && j == 0) atomicOr(err, 2u);  // contract: ascending, unique
        }
        const float *rp = dv.rows_f32 + row * dv.pitch;
        float r;
        if (METRIC == AH_MANHATTAN) {
            r = octet_manhattan(s_q, rp, dv.dims, j);
        } else {
            r = octet_reduce_stream<OP>(s_q4, rp, dv.dims, j);
        }
        if (j == 0) out[i] = f32_epilogue<METRIC>(r, s_hdr, dv, row);
    }
}

// Manhattan scan / gather, dims >= 32.  `built_distance` is a strictly sequential sum (src/distance/manhattan.rs:44-46):
// inside one row nothing can be reassociated, so the octet mapping above spends 8 VALU issue slots per element (every
// lane of the octet steps through every hand-off) and becomes VALU-bound.  Here the parallelism is across rows instead:
// a block takes 256 rows; per 32-element chunk the octets load 128-byte lines (coalesced, non-temporal) and park them in
// LDS, then thread t walks row t's 32 elements sequentially — one subtract and one add-with-|.| per element, no
// redundant work.  Rows are padded to 33 words in LDS so that the 64 lanes of a wave hit 64 different banks.
static constexpr uint32_t kManRows = 256;
template <bool GATHER>
__global__ __launch_bounds__(kBlock) void k_distances_manhattan(DataView dv, const float *__restrict__ qvec,
                                                                const uint32_t *__restrict__ ids, uint64_t n,
                                                                float *__restrict__ out, uint32_t *err) {
    __shared__ float s_x[kManRows * 33];
    __shared__ uint32_t s_row[kManRows];
    const uint32_t t = threadIdx.x, o = t >> 3, j = t & 7u;
    const uint32_t chunks = (dv.dims + 31) >> 5;
    const uint64_t n_tiles = (n + kManRows - 1) / kManRows;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * kManRows;
        const uint32_t rows_here = (uint32_t)min((uint64_t)kManRows, n - base);
        __syncthreads();  // the previous tile's readers are done with s_row / s_x
        uint32_t my_row = 0xFFFFFFFFu;
        if (t < rows_here) {
            uint64_t row = base + t;
            if (GATHER) {
                const uint32_t id = ids[base + t];
                row = row_of_id(dv, id);
                if (row == ~0ull) atomicOr(err, 1u);                                        // Error::MissingKey
                if (base + t > 0 && id <= ids[base + t - 1]) atomicOr(err, 2u);           // contract: ascending, unique
            }
            my_row = row == ~0ull ? 0xFFFFFFFFu : (uint32_t)row;
        }
        s_row[t] = my_row;
        __syncthreads();
        // first chunk's lines into registers
        float4 x[kManRows / 32];
#pragma unroll
        for (uint32_t p = 0; p < kManRows / 32; p++) {
            const uint32_t r = s_row[o + 32 * p];
            x[p] = r != 0xFFFFFFFFu ? ld_stream(reinterpret_cast<const float4 *>(dv.rows_f32 + (uint64_t)r * dv.pitch) + j)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float sum = 0.0f;
        for (uint32_t c = 0; c < chunks; c++) {
#pragma unroll
            for (uint32_t p = 0; p < kManRows / 32; p++) {
                float *dst = s_x + (o + 32 * p) * 33 + 4 * j;
                dst[0] = x[p].x;
                dst[1] = x[p].y;
                dst[2] = x[p].z;
                dst[3] = x[p].w;
            }
            __syncthreads();
            if (c + 1 < chunks) {  // next chunk's lines travel while this one is summed
#pragma unroll
                for (uint32_t p = 0; p < kManRows / 32; p++) {
                    const uint32_t r = s_row[o + 32 * p];
                    x[p] = r != 0xFFFFFFFFu ? ld_stream(reinterpret_cast<const float4 *>(dv.rows_f32 + (uint64_t)r * dv.pitch) +
                                                        (c + 1) * 8 + j)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const uint32_t lim = min(32u, dv.dims - c * 32);
            const float *mine = s_x + t * 33;
            const float *q = qvec + c * 32;  // uniform: scalar loads
            if (lim == 32) {
#pragma unroll
                for (uint32_t e = 0; e < 32; e++) sum = f_add(sum, fabsf(f_sub(q[e], mine[e])));
            } else {
                for (uint32_t e = 0; e < lim; e++) sum = f_add(sum, fabsf(f_sub(q[e], mine[e])));
            }
            __syncthreads();
        }
        if (t < rows_here) out[base + t] = my_row != 0xFFFFFFFFu ? sum : __uint_as_float(0x7FC00000u);
    }
}

// dims < 32: SSE / scalar tiers, one thread per row (test-sized inputs; not a performance path).
template <int METRIC, bool GATHER>
__global__ __launch_bounds__(kBlock) void k_distances_f32_small(DataView dv, const float *__restrict__ qvec,
                                                                const float *__restrict__ qhdr,
                                                                const uint32_t *__restrict__ ids, uint64_t n,
                                                                float *__restrict__ out, uint32_t *err) {
    constexpr int OP = METRIC == AH_EUCLIDEAN ? OP_EUCLID : OP_DOT;
    __shared__ float s_q[32];
    __shared__ float s_hdr[2];
    if (threadIdx.x < 32) s_q[threadIdx.x] = threadIdx.x < dv.dims ? qvec[threadIdx.x] : 0.0f;
    if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdr[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t row = i;
        if (GATHER) {
            row = row_of_id(dv, ids[i]);
            if (row == ~0ull) {
                atomicOr(err, 1u);
                out[i] = __uint_as_float(0x7FC00000u);
                continue;
            }
            if (i > 0 && ids[i] <= ids[i - 1]) atomicOr(err, 2u);
        }
        const float *rp = dv.rows_f32 + row * dv.pitch;
        float r;
        if (METRIC == AH_MANHATTAN) {
            r = 0.0f;
            for (uint32_t e = 0; e < dv.dims; e++) r = f_add(r, fabsf(f_sub(s_q[e], rp[e])));
        } else {
            r = thread_reduce_small<OP>(s_q, rp, dv.dims);
        }
        out[i] = f32_epilogue<METRIC>(r, s_hdr, dv, row);
    }
}

// 1-bit metrics, cooperative version (the fast path).  A 768-d row is only 96 bytes, so one-thread-per-row
// loads would touch 64 different lines per wave instruction.  Instead a wave takes 64 consecutive rows and
// reads them as `C` fully coalesced 1 KiB instructions (C = 16-byte chunks per row): lane L of load c owns
// chunk g = 64c + L -> (row g / C, part g % C), xor-popcounts it against the matching 16 bytes of the query
// (LDS), and parks the partial count in LDS; lane r then adds the C partials of row r (integer adds: exact in
// any order) and writes 64 contiguous distances.
// Algorithmic traffic per distance: 8*words (+4 header for BQ-cosine) read + 4 written.
static constexpr uint32_t kBqMaxChunks = 32;  // rows up to 512 bytes (dims <= 4096); larger rows use the fallback
static constexpr uint32_t kBqUnroll = 8;      // chunk loads in flight per lane

typedef unsigned long long bq_chunk_raw __attribute__((ext_vector_type(2)));
struct bq_chunk {
    unsigned long long x, y;
};
__device__ __forceinline__ bq_chunk ld_stream_bq(const uint64_t *p) {
    const bq_chunk_raw t = __builtin_nontemporal_load(reinterpret_cast<const bq_chunk_raw *>(p));
    return bq_chunk{t.x, t.y};
}

template <bool GATHER>
__global__ __launch_bounds__(kBlock) void k_distances_bq(DataView dv, const uint64_t *__restrict__ qvec,
                                                         const float *__restrict__ qhdr,
                                                         const uint32_t *__restrict__ ids, uint64_t n,
                                                         float *__restrict__ out, uint32_t *err) {
    extern __shared__ uint64_t s_qw[];  // [pitch] query words, then 4 x [64 * C] u32 partial counts
    const uint32_t C = dv.pitch >> 1;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *s_part = reinterpret_cast<uint32_t *>(s_qw + dv.pitch) + wave * 64 * C;
    for (uint32_t i = threadIdx.x; i < dv.pitch; i += blockDim.x) s_qw[i] = qvec[i];
    __shared__ float s_hdr[2];
    if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdr[threadIdx.x];
    __syncthreads();
    // Each wave owns a private slice of LDS and works on its own tiles: producer and consumer of s_part are lanes
    // of the SAME wave and LDS operations of one wave complete in issue order, so no workgroup barrier is needed —
    // only a compiler fence so the reads are not hoisted above the writes.
    const uint64_t n_tiles = (n + 63) >> 6;
    const uint64_t n_waves = (uint64_t)gridDim.x * (kBlock / 64);
    // chunk g = 64c + lane of a tile belongs to part g % C of row g / C; both advance by a constant per c
    const uint32_t step_r = 64u / C, step_p = 64u - step_r * C;
    for (uint64_t tile = (uint64_t)blockIdx.x * (kBlock / 64) + wave; tile < n_tiles; tile += n_waves) {
        const uint64_t base = tile << 6;
        const uint32_t rows_here = (uint32_t)min((uint64_t)64, n - base);
        uint32_t r = lane / C, part = lane - r * C;
        // the loads of kBqUnroll chunks are issued back to back (kBqUnroll KiB in flight per wave) before the
        // first popcount waits on them
        for (uint32_t c0 = 0; c0 < C; c0 += kBqUnroll) {
            bq_chunk v[kBqUnroll];
            uint32_t parts[kBqUnroll];
            bool ok[kBqUnroll];
#pragma unroll
            for (uint32_t u = 0; u < kBqUnroll; u++) {
                parts[u] = part;
                ok[u] = (c0 + u < C) && (r < rows_here);
                v[u] = bq_chunk{0ull, 0ull};
                if (ok[u]) {
                    uint64_t row = base + r;
                    if (GATHER) row = row_of_id(dv, ids[base + r]);
                    ok[u] = row != ~0ull;
                    if (ok[u]) v[u] = ld_stream_bq(dv.rows_bq + row * dv.pitch + 2 * part);
                }
                r += step_r;
                part += step_p;
                if (part >= C) {
                    part -= C;
                    r += 1;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kBqUnroll; u++) {
                if (c0 + u < C) {
                    const uint32_t pc = ok[u] ? (uint32_t)__popcll(v[u].x ^ s_qw[2 * parts[u]]) +
                                                    (uint32_t)__popcll(v[u].y ^ s_qw[2 * parts[u] + 1])
                                              : 0u;
                    s_part[(c0 + u) * 64 + lane] = pc;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < rows_here) {
            const uint64_t i = base + lane;
            uint64_t row = i;
            if (GATHER) {
                row = row_of_id(dv, ids[i]);
                if (i > 0 && ids[i] <= ids[i - 1]) atomicOr(err, 2u);
            }
            if (row == ~0ull) {
                atomicOr(err, 1u);
                out[i] = __uint_as_float(0x7FC00000u);
            } else {
                uint32_t ham = 0;
                for (uint32_t p = 0; p < C; p++) ham += s_part[lane * C + p];
                float d;
                if (dv.metric == AH_BQ_EUCLIDEAN) d = (float)(ham * 4u);       // bq_euclidean.rs:117-124
                else if (dv.metric == AH_BQ_MANHATTAN) d = (float)(ham * 2u);  // bq_manhattan.rs:113-120
                else d = bq_cosine_from_dot((float)bq_dot_from_hamming(ham, dv.words), s_hdr[0], dv.headers[row]);
                out[i] = d;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Fallback for very wide rows (> 512 bytes): one thread per row.
template <bool GATHER>
__global__ __launch_bounds__(kBlock) void k_distances_bq_wide(DataView dv, const uint64_t *__restrict__ qvec,
                                                              const float *__restrict__ qhdr,
                                                              const uint32_t *__restrict__ ids, uint64_t n,
                                                              float *__restrict__ out, uint32_t *err) {
    extern __shared__ uint64_t s_qw[];
    for (uint32_t i = threadIdx.x; i < dv.pitch; i += blockDim.x) s_qw[i] = qvec[i];
    __shared__ float s_hdr[2];
    if (threadIdx.x < 2) s_hdr[threadIdx.x] = qhdr[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t pairs = dv.pitch >> 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t row = i;
        if (GATHER) {
            row = row_of_id(dv, ids[i]);
            if (row == ~0ull) {
                atomicOr(err, 1u);
                out[i] = __uint_as_float(0x7FC00000u);
                continue;
            }
            if (i > 0 && ids[i] <= ids[i - 1]) atomicOr(err, 2u);
        }
        const ulonglong2 *rp = reinterpret_cast<const ulonglong2 *>(dv.rows_bq + row * dv.pitch);
        uint32_t ham = 0;
        for (uint32_t p = 0; p < pairs; p++) {
            ulonglong2 v = rp[p];
            ham += (uint32_t)__popcll(v.x ^ s_qw[2 * p]) + (uint32_t)__popcll(v.y ^ s_qw[2 * p + 1]);
        }
        float d;
        if (dv.metric == AH_BQ_EUCLIDEAN) d = (float)(ham * 4u);
        else if (dv.metric == AH_BQ_MANHATTAN) d = (float)(ham * 2u);
        else d = bq_cosine_from_dot((float)bq_dot_from_hamming(ham, dv.words), s_hdr[0], dv.headers[row]);
        out[i] = d;
    }
}

template <bool GATHER>
static int launch_distances_t(const DataView &dv, const void *qvec, const float *qhdr, const uint32_t *ids, uint64_t n,
                              float *out, uint32_t *err, hipStream_t s) {
    if (n == 0) return AH_OK;
    if (metric_is_bq(dv.metric)) {
        const uint32_t C = dv.pitch >> 1;
        if (C <= kBqMaxChunks) {
            const size_t sh = (size_t)dv.pitch * 8 + (size_t)(kBlock / 64) * 64 * C * 4;
            hipLaunchKernelGGL((k_distances_bq<GATHER>), dim3(grid_for(n, kBlock)), dim3(kBlock), sh, s, dv,
                               (const uint64_t *)qvec, qhdr, ids, n, out, err);
        } else {
            hipLaunchKernelGGL((k_distances_bq_wide<GATHER>), dim3(grid_for(n, kBlock)), dim3(kBlock), dv.pitch * 8, s,
                               dv, (const uint64_t *)qvec, qhdr, ids, n, out, err);
        }
    } else if (dv.dims >= 32) {
        const unsigned g = grid_for(n, kBlock / 8);
        const size_t sh = (size_t)dv.pitch * 4;
#define AH_LAUNCH_F32(M)                                                                                          \
    hipLaunchKernelGGL((k_distances_f32<M, GATHER>), dim3(g), dim3(kBlock), sh, s, dv, (const float *)qvec, qhdr, \
                       ids, n, out, err)
        switch (dv.metric) {
        case AH_EUCLIDEAN: AH_LAUNCH_F32(AH_EUCLIDEAN); break;
        case AH_MANHATTAN:
            if (g_manhattan_rows && dv.n < 0xFFFFFFFFull)
                hipLaunchKernelGGL((k_distances_manhattan<GATHER>), dim3(grid_for(n, kManRows)), dim3(kBlock), 0, s, dv,
                                   (const float *)qvec, ids, n, out, err);
            else
                AH_LAUNCH_F32(AH_MANHATTAN);
            break;
        case AH_COSINE: AH_LAUNCH_F32(AH_COSINE); break;
        default: AH_LAUNCH_F32(AH_DOT_PRODUCT); break;
        }
#undef AH_LAUNCH_F32
    } else {
        const unsigned g = grid_for(n, kBlock);
#define AH_LAUNCH_SMALL(M)                                                                                      \
    hipLaunchKernelGGL((k_distances_f32_small<M, GATHER>), dim3(g), dim3(kBlock), 0, s, dv, (const float *)qvec, \
                       qhdr, ids, n, out, err)
        switch (dv.metric) {
        case AH_EUCLIDEAN: AH_LAUNCH_SMALL(AH_EUCLIDEAN); break;
        case AH_MANHATTAN: AH_LAUNCH_SMALL(AH_MANHATTAN); break;
        case AH_COSINE: AH_LAUNCH_SMALL(AH_COSINE); break;
        default: AH_LAUNCH_SMALL(AH_DOT_PRODUCT); break;
        }
#undef AH_LAUNCH_SMALL
    }
    AH_HIP(hipGetLastError());
    return AH_OK;
}

int launch_distances(const DataView &dv, const void *d_qvec, const float *d_qhdr, const uint32_t *d_ids, uint64_t n,
                     float *d_out, uint32_t *d_err, hipStream_t s) {
    return d_ids ? launch_distances_t<true>(dv, d_qvec, d_qhdr, d_ids, n, d_out, d_err, s)
                 : launch_distances_t<false>(dv, d_qvec, d_qhdr, nullptr, n, d_out, d_err, s);
}

int launch_prepare_query(const DataView &dv, const float *d_query_f32, void *d_qvec, float *d_qhdr, hipStream_t s) {
    // (a thread per element of the usual query: d_query_f32 may be the caller's pinned buffer, and a read over the link that
    // waits for the one before it is 1.5 us)
    hipLaunchKernelGGL(k_prepare_query, dim3(1), dim3(metric_is_bq_dev(dv.metric) ? 64 : 1024), 0, s, dv, d_query_f32, d_qvec, d_qhdr);
    AH_HIP(hipGetLastError());
    return AH_OK;
}
int launch_load_item_as_query(const DataView &dv, uint32_t row, void *d_qvec, float *d_qhdr, hipStream_t s) {
    hipLaunchKernelGGL(k_load_item_as_query, dim3(1), dim3(64), 0, s, dv, row, d_qvec, d_qhdr);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// ------------------------------------------------------------------------------------------------
// top-k: `median_based_top_k` (src/reader.rs:607-640) = the k smallest (OrderedFloat(dist), id) tuples
// in ascending order.  Candidates arrive in ascending id order, so the POSITION in the candidate list
// is used as the tie-break word: same order as the id, and it recovers the original distance and id.
// Key = orderable(dist) << 32 | position.  A tournament of LDS bitonic sorts: every block sorts a chunk
// of kChunk keys and keeps its `keep` smallest; rounds repeat until one block is left.
// Faithful quirk: the reference skips items >= (f32::MAX, u32::MAX) once its 2k prefill buffer is full
// (reader.rs:611,619-621); such items at positions >= 2k get the sentinel key and are never selected.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t kChunk = 4096;
static constexpr uint64_t kSentinel = ~0ull;

__device__ __forceinline__ uint64_t make_key(float d, uint64_t pos, uint32_t id, uint64_t two_k) {
    uint32_t ok = orderable_key(d);
    if (pos >= two_k) {
        // item >= threshold (OrderedFloat(f32::MAX), u32::MAX)
        const uint32_t max_key = 0xFF7FFFFFu;  // orderable_key(f32::MAX)
        if (ok > max_key || (ok == max_key && id == 0xFFFFFFFFu)) return kSentinel;
    }
    return ((uint64_t)ok << 32) | (uint64_t)(uint32_t)pos;
}

__device__ __forceinline__ void bitonic_sort_lds(uint64_t *s, uint32_t n_pow2) {
    for (uint32_t size = 2; size <= n_pow2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (n_pow2 >> 1); t += blockDim.x) {
                uint32_t lo = 2 * t - (t & (stride - 1));
                uint32_t hi = lo + stride;
                bool up = (lo & size) == 0;
                uint64_t a = s[lo], b = s[hi];
                if ((a > b) == up) {
                    s[lo] = b;
                    s[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// FIRST round reads distances and builds keys; later rounds read keys.
template <bool FIRST>
__global__ __launch_bounds__(kBlock) void k_topk_round(const float *__restrict__ dist, const uint32_t *__restrict__ ids,
                                                       const uint32_t *__restrict__ row_ids, int have_ids,
                                                       const uint64_t *__restrict__ keys_in, uint64_t n_in,
                                                       uint64_t two_k, uint32_t keep, uint64_t *__restrict__ keys_out) {
    __shared__ uint64_t s[kChunk];
    const uint64_t base = (uint64_t)blockIdx.x * kChunk;
    for (uint32_t t = threadIdx.x; t < kChunk; t += blockDim.x) {
        uint64_t g = base + t;
        uint64_t key = kSentinel;
        if (g < n_in) {
            if (FIRST) {
                uint32_t id = have_ids ? ids[g] : (row_ids ? row_ids[g] : (uint32_t)g);
                key = make_key(dist[g], g, id, two_k);
            } else {
                key = keys_in[g];
            }
        }
        s[t] = key;
    }
    bitonic_sort_lds(s, kChunk);
    for (uint32_t t = threadIdx.x; t < keep; t += blockDim.x) keys_out[(uint64_t)blockIdx.x * keep + t] = s[t];
}

// Global-memory bitonic step for k > kChunk/2 (rare: `count` in the thousands).
__global__ void k_bitonic_global(uint64_t *keys, uint64_t n_pow2, uint64_t size, uint64_t stride) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (n_pow2 >> 1)) return;
    uint64_t lo = 2 * t - (t & (stride - 1));
    uint64_t hi = lo + stride;
    bool up = (lo & size) == 0;
    uint64_t a = keys[lo], b = keys[hi];
    if ((a > b) == up) {
        keys[lo] = b;
        keys[hi] = a;
    }
}
__global__ void k_make_keys(const float *dist, const uint32_t *ids, const uint32_t *row_ids, int have_ids, uint64_t n,
                            uint64_t n_pow2, uint64_t two_k, uint64_t *keys) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pow2) return;
    uint64_t key = kSentinel;
    if (g < n) {
        uint32_t id = have_ids ? ids[g] : (row_ids ? row_ids[g] : (uint32_t)g);
        key = make_key(dist[g], g, id, two_k);
    }
    keys[g] = key;
}

// Final: sorted keys -> (id, normalized distance).  src/reader.rs:396-399.
__global__ void k_topk_emit(DataView dv, const uint64_t *__restrict__ keys, const float *__restrict__ dist,
                            const uint32_t *__restrict__ ids, size_t k, uint32_t *out_ids, float *out_dist) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    uint64_t key = keys[t];
    uint32_t pos = (uint32_t)key;
    uint32_t id = ids ? ids[pos] : (dv.identity_ids ? pos : dv.ids[pos]);
    out_ids[t] = id;
    out_dist[t] = normalized_distance(dv.metric, dist[pos], dv.dims);
}

// The same selection for ONE short list in ONE launch (arroy re-ranks one query per call, src/reader.rs:381-399, and its
// candidate list is search_k = 10 000-odd ids): n <= kSmallTopkMax distances, k <= kSmallTopkCap.  A block holds every
// (distance, id) in registers (thread t owns the positions t + 1024 r), finds the histogram bin of the k-th smallest key, ranks
// the <= kSmallTopkCap keys up to that bin by counting (keys are unique: the position is their low word) and writes ids,
// normalized distances and the call's status word straight into the caller's PINNED buffers — the two tournament rounds,
// the emit kernel and three copies back of the general path were 170 of a call's 220 us.
// *err bit 2: a non-finite distance (src/reader.rs:611-621 looks at positions); bit 3: more than kSmallTopkCap keys up to
// the k-th's bin.  Either way nothing is written and the caller takes the general path.
static constexpr uint32_t kSmallTopkMax = 16384, kSmallTopkCap = 1024, kSmallTopkBins = 2048;
__global__ __launch_bounds__(1024) void k_topk_small(DataView dv, const float *__restrict__ dist, const uint32_t *__restrict__ ids,
                                                     uint32_t n, uint32_t k, uint32_t *__restrict__ out_ids,
                                                     float *__restrict__ out_dist, uint32_t *err, uint32_t *__restrict__ host_err) {
    constexpr uint32_t kThreads = 1024, kOwn = kSmallTopkMax / kThreads;
    __shared__ uint32_t s_hist[kSmallTopkBins];
    __shared__ uint64_t s_key[kSmallTopkCap];
    __shared__ uint32_t s_id[kSmallTopkCap];
    __shared__ float s_val[kSmallTopkCap];
    __shared__ uint32_t s_min, s_max, s_wave[kThreads / 64], s_bin, s_count, s_n;
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < kSmallTopkBins; b += kThreads) s_hist[b] = 0;
    if (tid == 0) {
        s_min = 0xFFFFFFFFu;
        s_max = 0u;
        s_n = 0u;
    }
    float own_d[kOwn];
    uint32_t own_id[kOwn];
#pragma unroll
    for (uint32_t r = 0; r < kOwn; r++) {
        const uint32_t g = tid + r * kThreads;
        own_d[r] = g < n ? dist[g] : 0.0f;
        own_id[r] = g < n ? (ids ? ids[g] : (dv.identity_ids ? g : dv.ids[g])) : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);  // (all the loads in flight before the first use)
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (uint32_t r = 0; r < kOwn; r++)
        if (tid + r * kThreads < n) {
            const uint32_t w = orderable_key(own_d[r]);
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
    uint32_t fail = s_max > 0xFF7FFFFFu ? 4u : 0u;  // +inf / NaN somewhere
    uint32_t n_sel = 0;
    if (!fail) {
        const uint64_t span = (uint64_t)(s_max - w_min) + 1ull;
        const bool direct = span <= kSmallTopkBins;
        const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)kSmallTopkBins << 32) / span);
        auto bin_of = [&](uint32_t w) -> uint32_t { return direct ? w - w_min : (uint32_t)(((uint64_t)(w - w_min) * scale) >> 32); };
#pragma unroll
        for (uint32_t r = 0; r < kOwn; r++)
            if (tid + r * kThreads < n) atomicAdd(&s_hist[bin_of(orderable_key(own_d[r]))], 1u);
        __syncthreads();
        {  // the bin of the k-th smallest key: thread t owns kPer consecutive bins
            constexpr uint32_t kPer = kSmallTopkBins / kThreads;
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
                if (before < k && before + c[u] >= k) {  // exactly one bin qualifies (k <= n)
                    s_bin = tid * kPer + u;
                    s_count = before + c[u];
                }
                before += c[u];
            }
        }
        __syncthreads();
        n_sel = s_count;
        const uint32_t bin_k = s_bin;
        if (n_sel > kSmallTopkCap) {
            fail = 8u;
        } else {
#pragma unroll
            for (uint32_t r = 0; r < kOwn; r++) {
                const uint32_t g = tid + r * kThreads;
                if (g >= n) continue;
                const uint32_t w = orderable_key(own_d[r]);
                if (bin_of(w) <= bin_k) {
                    const uint32_t at = atomicAdd(&s_n, 1u);
                    s_key[at] = ((uint64_t)w << 32) | g;  // (distance, position): the ids are ascending, so this is (distance, id)
                    s_id[at] = own_id[r];
                    s_val[at] = own_d[r];
                }
            }
            __syncthreads();
            for (uint32_t e = tid; e < n_sel; e += kThreads) {
                const uint64_t mine = s_key[e];
                uint32_t rank = 0, i = 0;
                for (; i + 8 <= n_sel; i += 8) {
                    uint64_t other[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++) other[u] = s_key[i + u];
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++) rank += other[u] < mine ? 1u : 0u;
                }
                for (; i < n_sel; i++) rank += s_key[i] < mine ? 1u : 0u;
                if (rank < k) {
                    out_ids[rank] = s_id[e];
                    out_dist[rank] = normalized_distance(dv.metric, s_val[e], dv.dims);
                }
            }
        }
    }
    if (tid == 0) {  // the status word: what the kernels before this one raised, and this one's verdict
        const uint32_t before = fail ? atomicOr(err, fail) : __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (host_err) *host_err = before | fail;
    }
}
// n and k within the kernel's limits?
bool topk_small_fits(uint64_t n, size_t k) { return n <= kSmallTopkMax && k <= kSmallTopkCap && k <= n; }
int launch_topk_small(const DataView &dv, const float *d_dist, const uint32_t *ids, uint64_t n, size_t k, uint32_t *out_ids,
                      float *out_dist, uint32_t *d_err, uint32_t *host_err, hipStream_t s) {
    hipLaunchKernelGGL(k_topk_small, dim3(1), dim3(1024), 0, s, dv, d_dist, ids, (uint32_t)n, (uint32_t)k, out_ids, out_dist, d_err, host_err);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

static uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

size_t topk_scratch_bytes(uint64_t n, size_t k) {
    if (k > kChunk / 2) return next_pow2(n) * 8 + 64;
    uint64_t blocks = (n + kChunk - 1) / kChunk;
    return (size_t)(blocks * (uint64_t)kChunk * 8 * 2 + 64);  // two ping-pong key buffers (upper bound)
}

// k = min(count, n) already; d_ids == nullptr means "positions are rows" (ids from the dataset).
int launch_topk(const DataView &dv, const float *d_dist, const uint32_t *d_ids, uint64_t n, size_t k, void *d_scratch,
                uint32_t *d_out_ids, float *d_out_dist, hipStream_t s) {
    if (k == 0 || n == 0) return AH_OK;
    const uint64_t two_k = 2 * (uint64_t)k;
    const uint32_t *row_ids = (d_ids == nullptr && !dv.identity_ids) ? dv.ids : nullptr;
    uint64_t *bufA = reinterpret_cast<uint64_t *>(d_scratch);
    const uint64_t *sorted = nullptr;
    if (k > kChunk / 2) {
        const uint64_t np = next_pow2(n);
        hipLaunchKernelGGL(k_make_keys, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, d_dist, d_ids, row_ids,
                           d_ids != nullptr, n, np, two_k, bufA);
        for (uint64_t size = 2; size <= np; size <<= 1)
            for (uint64_t stride = size >> 1; stride > 0; stride >>= 1)
                hipLaunchKernelGGL(k_bitonic_global, dim3((unsigned)(((np >> 1) + 255) / 256)), dim3(256), 0, s, bufA,
                                   np, size, stride);
        sorted = bufA;
    } else {
        uint64_t blocks = (n + kChunk - 1) / kChunk;
        uint64_t *bufB = bufA + blocks * kChunk;
        const uint32_t keep = (uint32_t)(blocks == 1 ? k : (k < kChunk / 2 ? k : kChunk / 2));
        hipLaunchKernelGGL((k_topk_round<true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, d_dist, d_ids, row_ids,
                           d_ids != nullptr, (const uint64_t *)nullptr, n, two_k, keep, bufA);
        uint64_t n_cur = blocks * keep;
        uint64_t *cur = bufA, *nxt = bufB;
        while (blocks > 1) {
            blocks = (n_cur + kChunk - 1) / kChunk;
            hipLaunchKernelGGL((k_topk_round<false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const float *)nullptr,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0, (const uint64_t *)cur, n_cur,
                               two_k, keep, nxt);
            n_cur = blocks * keep;
            uint64_t *tmp = cur;
            cur = nxt;
            nxt = tmp;
        }
        sorted = cur;
    }
    hipLaunchKernelGGL(k_topk_emit, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, dv, sorted, d_dist, d_ids, k,
                       d_out_ids, d_out_dist);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// ------------------------------------------------------------------------------------------------
// Dataset helper kernels
// ------------------------------------------------------------------------------------------------

// D::new_header for freshly uploaded rows (src/writer.rs:380-394 -> cosine.rs:39-41, bq_cosine.rs:45-47).
__global__ __launch_bounds__(kBlock) void k_headers_from_vectors(DataView dv, uint64_t first_row, uint64_t n) {
    const uint32_t hf = dv.metric == AH_DOT_PRODUCT ? 2u : 1u;
    if (dv.metric == AH_COSINE) {
        const uint32_t j = threadIdx.x & 7u;
        const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
        for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += n_octets) {
            const float *rp = dv.rows_f32 + (first_row + i) * dv.pitch;
            float d = octet_reduce_any<OP_DOT>(rp, rp, dv.dims, j);
            if (j == 0) dv.headers[first_row + i] = f_sqrt(d);
        }
    } else {
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float h0 = 0.0f;
            if (dv.metric == AH_BQ_COSINE) h0 = f_sqrt((float)(int32_t)(64u * dv.words));
            dv.headers[(first_row + i) * hf] = h0;
            if (hf == 2) dv.headers[(first_row + i) * hf + 1] = 0.0f;
        }
    }
}
int launch_headers_from_vectors(const DataView &dv, uint64_t first_row, uint64_t n, hipStream_t s) {
    if (n == 0) return AH_OK;
    hipLaunchKernelGGL(k_headers_from_vectors, dim3(grid_for(n, kBlock / 8)), dim3(kBlock), 0, s, dv, first_row, n);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// UnalignedVector::<BinaryQuantized>::from_slice for n rows (binary_quantized.rs:80-91).  One wave per run of 64
// consecutive output words: for each word the 64 lanes read its 64 source floats as one coalesced 256-byte
// load and the ballot of "sign bit clear" IS the packed word (bit i = is_sign_positive(x[64w + i]), :84-87;
// elements past `dims` and the pitch padding words stay 0).  Lane j keeps word j; one coalesced store per run.
__global__ __launch_bounds__(256) void k_quantize_rows(const float *__restrict__ src, uint32_t src_pitch, uint32_t dims,
                                                       uint64_t *__restrict__ dst, uint32_t dst_pitch, uint32_t words,
                                                       uint64_t n) {
    const uint64_t total = n * dst_pitch;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
    for (uint64_t g0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) << 6; g0 < total; g0 += n_waves << 6) {
        uint64_t row = g0 / dst_pitch;
        uint32_t w = (uint32_t)(g0 - row * dst_pitch);
        uint64_t mine = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < 64; j++) {
            bool positive = false;
            const uint32_t e = 64u * w + lane;
            if (g0 + j < total && w < words && e < dims)
                positive = (__float_as_uint(__builtin_nontemporal_load(src + row * src_pitch + e)) >> 31) == 0u;
            const uint64_t word = __ballot(positive);
            if (lane == j) mine = word;
            if (++w == dst_pitch) {
                w = 0;
                row++;
            }
        }
        if (g0 + lane < total) dst[g0 + lane] = mine;
    }
}
int launch_quantize_rows(const float *d_src, uint32_t src_pitch, uint32_t dims, uint64_t *d_dst, uint32_t dst_pitch,
                         uint32_t words, uint64_t n, hipStream_t s) {
    if (n == 0) return AH_OK;
    hipLaunchKernelGGL(k_quantize_rows, dim3(grid_for(n * dst_pitch, 256)), dim3(256), 0, s, d_src, src_pitch, dims,
                       d_dst, dst_pitch, words, n);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// Synthetic rows (include/arroy_hip_policy.h): one thread per float4 of the padded row.  The two structured distributions
// compute the row's part (its cluster / its 32 factors) once per thread instead of once per component; the values are those
// of ah_synth_value (tests/test_gpu_synth.py compares every component with the host's).
__global__ void k_synth_fill(float *__restrict__ rows, uint32_t pitch, uint32_t dims, uint64_t first_item, uint64_t n,
                             uint64_t seed, int distribution) {
    const uint64_t per_row = pitch >> 2;
    const uint64_t total = n * per_row;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint64_t row = g / per_row;
        const uint64_t item = first_item + row;
        const uint32_t c = (uint32_t)(g % per_row) * 4;
        float e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (distribution == AH_SYNTH_CLUSTERED) {
            int is_copy;
            const uint32_t cl = ah_synth_cluster_of(seed, item, &is_copy);
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (c + j < dims)
                    e[j] = ah_synth_clustered_from(ah_synth_centre_q20(seed, cl, c + j, dims),
                                                   is_copy ? 0 : ah_synth_normal_q20(ah_synth_hash(seed, item, c + j, dims)));
        } else if (distribution == AH_SYNTH_LOW_RANK) {
            int32_t signal[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < AH_SYNTH_FACTORS; k++) {
                const int32_t f = ah_synth_factor(seed, item, k);
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    if (c + j < dims) signal[j] += f * ah_synth_loading(seed, k, c + j, dims);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (c + j < dims) e[j] = ah_synth_low_rank_from(signal[j], ah_synth_normal_q20(ah_synth_hash(seed, item, c + j, dims)));
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (c + j < dims) e[j] = ah_synth_value(seed, item, c + j, dims, distribution);
        }
        reinterpret_cast<float4 *>(rows)[g] = make_float4(e[0], e[1], e[2], e[3]);
    }
}
int launch_synth_fill(float *d_rows, uint32_t pitch, uint32_t dims, uint64_t first_item, uint64_t n, uint64_t seed,
                      int distribution, hipStream_t s) {
    if (n == 0) return AH_OK;
    hipLaunchKernelGGL(k_synth_fill, dim3(grid_for(n * (pitch >> 2), 256)), dim3(256), 0, s, d_rows, pitch, dims,
                       first_item, n, seed, distribution);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// Read ceiling probe: every lane streams 8 x 16 bytes per step like the scan kernels (non-temporal), folds them into an
// integer and only one lane per block touches memory again (so nothing but the reads is measured).
__global__ __launch_bounds__(kBlock) void k_bench_read(const float4 *__restrict__ src, uint64_t n16,
                                                       unsigned long long *sink) {
    uint32_t acc = 0;
    const uint64_t base = (uint64_t)blockIdx.x * (kBlock * 8) + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * (kBlock * 8);
    for (uint64_t i = base; i + 7 * kBlock < n16; i += stride) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = ld_stream(src + i + (uint64_t)u * kBlock);
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc += __float_as_uint(x[u].x) ^ __float_as_uint(x[u].y) ^ __float_as_uint(x[u].z) ^ __float_as_uint(x[u].w);
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ull);  // keeps the loads alive; practically never taken
}
int launch_bench_read(const void *d_src, uint64_t bytes, unsigned long long *d_sink, hipStream_t s) {
    const uint64_t n16 = bytes / 16;
    const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(n16 / (kBlock * 8), 1u << 20));
    hipLaunchKernelGGL(k_bench_read, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const float4 *)d_src, n16, d_sink);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

__global__ void k_build_lut(const uint32_t *__restrict__ ids, uint64_t n, uint32_t *__restrict__ lut) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) lut[ids[g]] = (uint32_t)g;
}
int launch_build_lut(const uint32_t *d_ids, uint64_t n, uint32_t *d_lut, uint32_t lut_len, hipStream_t s) {
    AH_HIP(hipMemsetAsync(d_lut, 0xFF, (size_t)lut_len * 4, s));
    if (n) hipLaunchKernelGGL(k_build_lut, dim3(grid_for(n, 256)), dim3(256), 0, s, d_ids, n, d_lut);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// DotProduct::preprocess (src/distance/dot_product.rs:119-165).  Pass 1: max over items of
// sqrt(dot(v,v)) (f32::max ignores NaN; norms are >= 0 so the u32 bit pattern orders like the float).
// Pass 2: norm = max*max, extra_dim = sqrt(max*max - |v|*|v|).
__global__ __launch_bounds__(kBlock) void k_dot_max_norm(DataView dv, unsigned int *max_bits) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < dv.n; i += n_octets) {
        const float *rp = dv.rows_f32 + i * dv.pitch;
        float nrm = f_sqrt(octet_reduce_any<OP_DOT>(rp, rp, dv.dims, j));
        if (j == 0 && nrm == nrm) atomicMax(max_bits, __float_as_uint(fmaxf(nrm, 0.0f)));
    }
}
__global__ __launch_bounds__(kBlock) void k_dot_write_headers(DataView dv, const unsigned int *max_bits) {
    const float max_norm = __uint_as_float(*max_bits);
    const float m2 = f_mul(max_norm, max_norm);
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < dv.n; i += n_octets) {
        const float *rp = dv.rows_f32 + i * dv.pitch;
        float nrm = f_sqrt(octet_reduce_any<OP_DOT>(rp, rp, dv.dims, j));
        if (j == 0) {
            float diff = f_sub(m2, f_mul(nrm, nrm));
            dv.headers[2 * i + 0] = f_sqrt(diff);  // extra_dim
            dv.headers[2 * i + 1] = m2;            // norm
        }
    }
}
int launch_preprocess_dot(const DataView &dv, float *d_max_norm_bits, hipStream_t s) {
    AH_HIP(hipMemsetAsync(d_max_norm_bits, 0, 4, s));
    if (dv.n) {
        const unsigned g = grid_for(dv.n, kBlock / 8);
        hipLaunchKernelGGL(k_dot_max_norm, dim3(g), dim3(kBlock), 0, s, dv, (unsigned int *)d_max_norm_bits);
        hipLaunchKernelGGL(k_dot_write_headers, dim3(g), dim3(kBlock), 0, s, dv, (const unsigned int *)d_max_norm_bits);
    }
    AH_HIP(hipGetLastError());
    return AH_OK;
}

// Reader::item_vector (src/reader.rs:266-276): f32 codec = the floats; BQ codec = +-1.0 per bit
// (binary_quantized.rs:261-290), truncated to `dims`.
__global__ void k_decode_item(DataView dv, uint32_t row, float *out) {
    for (uint32_t i = threadIdx.x; i < dv.dims; i += blockDim.x) {
        if (!metric_is_bq_dev(dv.metric)) {
            out[i] = dv.rows_f32[(uint64_t)row * dv.pitch + i];
        } else {
            uint64_t w = dv.rows_bq[(uint64_t)row * dv.pitch + (i >> 6)];
            out[i] = f_sub(f_mul((float)((w >> (i & 63)) & 1ull), 2.0f), 1.0f);
        }
    }
}
int launch_decode_item(const DataView &dv, uint32_t row, float *d_out, hipStream_t s) {
    hipLaunchKernelGGL(k_decode_item, dim3(1), dim3(256), 0, s, dv, row, d_out);
    AH_HIP(hipGetLastError());
    return AH_OK;
}

}  // namespace ah
