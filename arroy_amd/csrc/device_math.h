// device_math.h — device-side arithmetic that reproduces arroy's f32 reduction ORDER exactly.
//
// Why order matters: the north star asks for bit-exact item-id ordering under the (distance, id)
// tie-break.  "Within 1e-5" distances do not give that for near-ties; reproducing the reference's
// summation tree does.  The reference's x86-64 AVX+FMA kernel (src/spaces/simple_avx.rs:17-110) is
// 32 independent FMA chains — element i feeds chain i mod 32 — reduced by a fixed tree
// (hsum256 :8-13, then ((h0+h1)+h2)+h3 :55-58) and a sequential scalar tail.
//
// Mapping to CDNA4: an OCTET = 8 consecutive lanes of a wave64 owns one vector.  Lane j loads the
// float4 at element 32k+4j (so the octet reads one whole 128-byte line per step, 16 B per lane) and
// owns chains 4j..4j+3, i.e. AVX accumulator a = j/2, AVX lanes 4(j%2)..4(j%2)+3.  The k-loop is a
// per-lane fmaf chain in the same k order as the reference; the reduction is the same tree done with
// three cross-lane steps.  No LDS, no atomics, no dependence on scheduling.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

namespace ah {

// IEEE single ops.  The TU is built with -ffp-contract=off (Rust never fuses a*b+c) and
// -fhip-fp32-correctly-rounded-divide-sqrt ('/' and sqrtf lower to the correctly rounded sequences).
// NOTE: HIP's __fsqrt_rn / __fdiv_rn are NOT used: without OCML_BASIC_ROUNDED_OPERATIONS __fsqrt_rn is
// __ocml_native_sqrt_f32 (1 ulp), which breaks bit parity with the reference's f32::sqrt.
#pragma clang fp contract(off)
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }
__device__ __forceinline__ float f_sqrt(float a) { return sqrtf(a); }

enum { OP_DOT = 0, OP_EUCLID = 1 };

// Streaming (read-once) 16-byte load: item rows are touched once per pass, so they are loaded with the
// non-temporal policy and do not displace the query / normals / headers from L2.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
    f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__host__ __device__ __forceinline__ bool metric_is_bq_dev(int m) { return m >= AH_BQ_EUCLIDEAN; }

template <int OP>
__device__ __forceinline__ void fma_step(float4 &acc, const float4 x, const float4 q) {
    if (OP == OP_DOT) {
        acc.x = fmaf(x.x, q.x, acc.x);
        acc.y = fmaf(x.y, q.y, acc.y);
        acc.z = fmaf(x.z, q.z, acc.z);
        acc.w = fmaf(x.w, q.w, acc.w);
    } else {  // _mm256_sub_ps then _mm256_fmadd_ps(s, s, acc): simple_avx.rs:34-48
        float sx = f_sub(x.x, q.x), sy = f_sub(x.y, q.y), sz = f_sub(x.z, q.z), sw = f_sub(x.w, q.w);
        acc.x = fmaf(sx, sx, acc.x);
        acc.y = fmaf(sy, sy, acc.y);
        acc.z = fmaf(sz, sz, acc.z);
        acc.w = fmaf(sw, sw, acc.w);
    }
}

// Cross-lane moves inside an octet as DPP modifiers (VALU data path): no LDS crossbar round trip (ds_bpermute), which
// for short rows (128-d: 4 FMA steps) used to cost more than the arithmetic.
//   0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2), 0x00 / 0xAA = broadcast lane 0 / 2 of the
//   quad, 0x141 = row_half_mirror (lane 7 - i of the octet: the other quad)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// hsum256 + the 4-accumulator sum, simple_avx.rs:8-13,55-58.  `acc` holds chains 4j..4j+3 of lane j
// of the octet.  All 8 lanes return the result.
__device__ __forceinline__ float octet_finish(const float4 acc) {
    // x128[l] = c[l+4] + c[l]: partner lane j^1 holds the other half of the same AVX accumulator.
    float x0 = f_add(dpp_f32<0xB1>(acc.x), acc.x);
    float x1 = f_add(dpp_f32<0xB1>(acc.y), acc.y);
    float x2 = f_add(dpp_f32<0xB1>(acc.z), acc.z);
    float x3 = f_add(dpp_f32<0xB1>(acc.w), acc.w);
    // x64[0] = x128[0]+x128[2], x64[1] = x128[1]+x128[3]; x32 = x64[0]+x64[1]
    float h = f_add(f_add(x0, x2), f_add(x1, x3));
    // lanes 2a, 2a+1 hold hsum(acc_a); result = ((h0 + h1) + h2) + h3.  Inside each quad: lane 0's and lane 2's value;
    // the other quad's pair through the half-row mirror; the low quad holds (h0, h1), the high quad (h2, h3).
    const float a = dpp_f32<0x00>(h), b = dpp_f32<0xAA>(h);
    const float ma = dpp_f32<0x141>(a), mb = dpp_f32<0x141>(b);
    const bool low = (threadIdx.x & 4u) == 0;
    const float h0 = low ? a : ma, h1 = low ? b : mb, h2 = low ? ma : a, h3 = low ? mb : b;
    return f_add(f_add(f_add(h0, h1), h2), h3);
}

// Sequential tail `for i in m..n: r += a[i]*b[i]` (mul, then add) simple_avx.rs:59-63,104-108.
template <int OP>
__device__ __forceinline__ float scalar_tail(float r, const float *a, const float *b, uint32_t from, uint32_t to) {
    for (uint32_t i = from; i < to; i++) {
        if (OP == OP_DOT) {
            r = f_add(r, f_mul(a[i], b[i]));
        } else {
            float s = f_sub(a[i], b[i]);
            r = f_add(r, f_mul(s, s));
        }
    }
    return r;
}

// Full AVX-tier reduction by one octet; a and b are 16-byte aligned rows (global or LDS), dims >= 32.
// j = lane index inside the octet.  Every lane of the octet gets the result.
template <int OP>
__device__ __forceinline__ float octet_reduce(const float *a, const float *b, uint32_t dims, uint32_t j) {
    const uint32_t blocks = dims >> 5;
    const float4 *a4 = reinterpret_cast<const float4 *>(a) + j;
    const float4 *b4 = reinterpret_cast<const float4 *>(b) + j;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (uint32_t k = 0; k < blocks; k++) fma_step<OP>(acc, a4[k * 8], b4[k * 8]);
    float r = octet_finish(acc);
    return scalar_tail<OP>(r, a, b, blocks << 5, dims);
}

// The same reduction — same chains, same order, same bits — with the line-loads of a whole chunk of the streamed operand `a`
// requested before the first multiply-add, the chunk as large as the row allows (48 blocks of 32 dims, then 24, 12, 6, 3, 1):
// for callers whose time is the LATENCY of one row after the other (the tree descent: a pop's margin reads a 6 KB normal
// nobody has touched before; with four loads in flight a 1536-d margin is twelve round trips to L2 / HBM, here it is one).
// `b` (the query) is expected in LDS or L1-hot.  Costs up to 192 registers: for kernels that run a few waves per CU.
template <int OP, int CH, bool B_LDS>
__device__ __forceinline__ void octet_wide_chunks(float4 &acc, const float4 *a4, const float4 *b4, uint32_t &k, uint32_t blocks) {
    while (k + CH <= blocks) {
        float4 x[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) x[u] = a4[(k + u) * 8];
        // Every request of the chunk leaves before the first value is used.  Left alone, the scheduler sinks the loads to
        // their uses to save registers and the chain becomes one trip to memory per pair of loads (measured: 12 000 cycles
        // for a 1536-d margin, 5 us of the 6 us a pop of the descent took).
        if constexpr (!B_LDS) {
            float4 y[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) y[u] = b4[(k + u) * 8];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; u++) fma_step<OP>(acc, x[u], y[u]);
        } else {
            // b in LDS: its reads go in groups of G, group g + 1 requested before the multiply-adds of group g
            constexpr int G = (CH % 8 == 0) ? 8 : CH, NG = CH / G;
            float4 y[2][G];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; u++) y[0][u] = b4[(k + u) * 8];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int u = 0; u < G; u++) y[(g + 1) & 1][u] = b4[(k + (g + 1) * G + u) * 8];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < G; u++) fma_step<OP>(acc, x[g * G + u], y[g & 1][u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        k += CH;
    }
    if constexpr (CH > 1) octet_wide_chunks<OP, CH / 2, B_LDS>(acc, a4, b4, k, blocks);
}
// a: global memory; b: global memory, or LDS (B_LDS: the query leaf of the descent kernels).  Same arithmetic as octet_reduce.
template <int OP, bool B_LDS = false>
__device__ __forceinline__ float octet_reduce_wide(const float *a, const float *b, uint32_t dims, uint32_t j) {
    const uint32_t blocks = dims >> 5;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t k = 0;
    octet_wide_chunks<OP, B_LDS ? 48 : 24, B_LDS>(acc, reinterpret_cast<const float4 *>(a) + j, reinterpret_cast<const float4 *>(b) + j, k,
                                                   blocks);
    float r = octet_finish(acc);
    return scalar_tail<OP>(r, a, b, blocks << 5, dims);
}

// Same reduction with the streamed operand `row` in global memory (read once, non-temporal) and the broadcast
// operand `s_b4` (query / normal) in LDS: 8 line-loads (128 B per lane, 8 KiB per wave) are issued before the
// first use so HBM latency is covered by loads in flight rather than by occupancy alone.  dims >= 32.
template <int OP, int FLY = 8, bool FENCE = false>
__device__ __forceinline__ float octet_reduce_stream(const float4 *s_b4, const float *row, uint32_t dims, uint32_t j) {
    const uint32_t blocks = dims >> 5;
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + j;
    const float4 *b4 = s_b4 + j;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t k = 0;
    for (; k + FLY <= blocks; k += FLY) {
        float4 x[FLY];
#pragma unroll
        for (int u = 0; u < FLY; u++) x[u] = ld_stream(r4 + (k + u) * 8);
        // FENCE: all FLY requests leave before the first use (a latency-bound caller — few octets, nothing else to switch to —
        // cannot afford the scheduler sinking loads to their uses; see octet_wide_chunks)
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < FLY; u++) fma_step<OP>(acc, b4[(k + u) * 8], x[u]);
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    for (; k < blocks; k++) fma_step<OP>(acc, b4[k * 8], r4[k * 8]);
    float r = octet_finish(acc);
    return scalar_tail<OP>(r, reinterpret_cast<const float *>(s_b4), row, blocks << 5, dims);
}

// SSE tier (16 <= dims < 32, simple_sse.rs) and scalar tier (dims < 16, simple.rs:49-51,81-83),
// executed by ONE thread.  16 chains, multiply THEN add (never fused); hsum128 = (x0+x2)+(x1+x3).
template <int OP>
__device__ __forceinline__ float thread_reduce_small(const float *a, const float *b, uint32_t dims) {
    float r = 0.0f;
    uint32_t m = 0;
    if (dims >= 16) {
        m = dims - (dims % 16);  // == 16 here
        float c[16];
#pragma unroll
        for (int l = 0; l < 16; l++) c[l] = 0.0f;
        for (uint32_t i = 0; i < m; i += 16) {
#pragma unroll
            for (int l = 0; l < 16; l++) {
                float p;
                if (OP == OP_DOT) {
                    p = f_mul(a[i + l], b[i + l]);
                } else {
                    float s = f_sub(a[i + l], b[i + l]);
                    p = f_mul(s, s);
                }
                c[l] = f_add(p, c[l]);
            }
        }
        float h[4];
#pragma unroll
        for (int g = 0; g < 4; g++) h[g] = f_add(f_add(c[4 * g + 0], c[4 * g + 2]), f_add(c[4 * g + 1], c[4 * g + 3]));
        r = f_add(f_add(f_add(h[0], h[1]), h[2]), h[3]);
    }
    return scalar_tail<OP>(r, a, b, m, dims);
}

// Generic exact reduction callable by all 8 lanes of an octet for ANY dims (small dims: every lane
// redundantly runs the one-thread tier; results identical).
template <int OP>
__device__ __forceinline__ float octet_reduce_any(const float *a, const float *b, uint32_t dims, uint32_t j) {
    if (dims >= 32) return octet_reduce<OP>(a, b, dims, j);
    return thread_reduce_small<OP>(a, b, dims);
}

// Manhattan built_distance: strictly sequential sum of |p-q| (src/distance/manhattan.rs:44-46).  The
// octet loads whole lines like the other metrics; |x-q| is elementwise-exact; the running sum is
// handed from lane to lane in element order.
// One 32-element block: lane j owns elements 4j..4j+3; the running sum enters at lane 0 and is handed to the next lane
// by a DPP row shift (an ALU move, no LDS round trip); after 8 steps lane 7 holds the block's sum and one octet-wide
// broadcast returns it to every lane (lanes other than the one whose turn it is compute values nobody reads).
__device__ __forceinline__ float manhattan_block(float r, const float4 x, const float4 q) {
    const float e0 = fabsf(f_sub(x.x, q.x)), e1 = fabsf(f_sub(x.y, q.y)), e2 = fabsf(f_sub(x.z, q.z)),
                e3 = fabsf(f_sub(x.w, q.w));
    float mine = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        mine = f_add(f_add(f_add(f_add(r, e0), e1), e2), e3);
        // row_shr:1 — lane l receives lane l-1's value (lane 0 / 8 of a 16-lane row receive a value that is never used)
        r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x111, 0xF, 0xF, false));
    }
    return __shfl(mine, 7, 8);
}
// `a` = query / centroid (LDS or global), `b` = the streamed row: its line-loads are issued 8 at a time ahead of the
// serial chain, like octet_reduce_stream, so the chain's latency is not added to the memory latency.
__device__ __forceinline__ float octet_manhattan(const float *a, const float *b, uint32_t dims, uint32_t j) {
    const uint32_t blocks = dims >> 5;
    const float4 *a4 = reinterpret_cast<const float4 *>(a) + j;
    const float4 *b4 = reinterpret_cast<const float4 *>(b) + j;
    float r = 0.0f;
    uint32_t k = 0;
    for (; k + 8 <= blocks; k += 8) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = b4[(k + u) * 8];
#pragma unroll
        for (int u = 0; u < 8; u++) r = manhattan_block(r, a4[(k + u) * 8], x[u]);
    }
    for (; k < blocks; k++) r = manhattan_block(r, a4[k * 8], b4[k * 8]);
    for (uint32_t i = blocks << 5; i < dims; i++) r = f_add(r, fabsf(f_sub(a[i], b[i])));
    return r;
}

// ---- 1-bit codec -------------------------------------------------------------------------------
// hamming over `words` 64-bit words (src/distance/binary_quantized_euclidean.rs:117-124 counts bytes;
// popcount is order-free).
__device__ __forceinline__ uint32_t bq_hamming(const uint64_t *a, const uint64_t *b, uint32_t words) {
    uint32_t h = 0;
    for (uint32_t w = 0; w < words; w++) h += (uint32_t)__popcll(a[w] ^ b[w]);
    return h;
}
// dot_product_binary_quantized (src/spaces/simple.rs:119-131) = 64*words - 2*hamming, as i32.
__device__ __forceinline__ int32_t bq_dot_from_hamming(uint32_t hamming, uint32_t words) {
    return (int32_t)(64u * words) - 2 * (int32_t)hamming;
}

// ---- metric epilogues (src/distance/<metric>.rs built_distance / margin) --------------------------
// f32 metrics: `r` is dot(p,q) (or sqeuclid / manhattan); ph/qh are the two headers.
__device__ __forceinline__ float cosine_from_dot(float pq, float pn, float qn) {  // cosine.rs:43-59
    float pnqn = f_mul(pn, qn);
    if (pnqn > 1.1920929e-7f) {
        float c = f_div(pq, pnqn);
        if (c < -1.0f) c = -1.0f;  // f32::clamp: NaN passes through
        if (c > 1.0f) c = 1.0f;
        return f_div(f_sub(1.0f, c), 2.0f);
    }
    return 0.0f;
}
__device__ __forceinline__ float bq_cosine_from_dot(float pq, float pn, float qn) {  // bq_cosine.rs:49-64
    float pnqn = f_mul(pn, qn);
    if (pnqn != 0.0f) {
        float c = f_div(pq, pnqn);
        return f_div(f_sub(1.0f, c), 2.0f);
    }
    return 0.0f;
}

// D::normalized_distance (src/distance/mod.rs:59-61 and the per-metric overrides).
__device__ __forceinline__ float normalized_distance(int metric, float d, uint32_t dims) {
    switch (metric) {
    case AH_EUCLIDEAN: return f_sqrt(d);
    case AH_MANHATTAN: return fmaxf(d, 0.0f);
    case AH_COSINE: return d;
    case AH_DOT_PRODUCT: return -d;
    case AH_BQ_EUCLIDEAN: return f_div(d, (float)dims);
    case AH_BQ_MANHATTAN: return f_div(fmaxf(d, 0.0f), (float)dims);
    default: return d;  // AH_BQ_COSINE
    }
}

// OrderedFloat<f32> as an unsigned key: NaN greatest and all equal, -0 == +0 (ordered-float 4.6).
__device__ __forceinline__ uint32_t orderable_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    if (f == 0.0f) return 0x80000000u;
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// id -> row.  Returns 0xFFFFFFFFFFFFFFFF when the item does not exist.
__device__ __forceinline__ uint64_t row_of_id(const DataView &dv, uint32_t id) {
    if (dv.identity_ids) return id < dv.n ? (uint64_t)id : ~0ull;
    if (dv.lut) {
        if (id >= dv.lut_len) return ~0ull;
        uint32_t r = dv.lut[id];
        return r == 0xFFFFFFFFu ? ~0ull : (uint64_t)r;
    }
    uint64_t lo = 0, hi = dv.n;  // binary search in the ascending id array
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        uint32_t v = dv.ids[mid];
        if (v < id) lo = mid + 1;
        else hi = mid;
    }
    return (lo < dv.n && dv.ids[lo] == id) ? lo : ~0ull;
}

}  // namespace ah
