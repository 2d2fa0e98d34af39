// screen_device.h — certified binary16 screen of the build's margin loop (src/writer.rs:1201-1207).
//
// The tree build never uses the VALUE of a margin, only its sign (`D::side`, src/distance/mod.rs:103-110).  The
// reference computes r = fl_ref(<n, x>) (+ bias, or + h0 * extra_dim) in f32 with its fixed 32-chain order; the forest
// kernels reproduce that order exactly, which costs 4*dims bytes of HBM (node-major) or of L2 -> CU traffic
// (row-major) per (item, node) pair — the bound of every level of the build.
//
// The screen evaluates s = <n~, x~> on binary16 copies n~, x~ (half the bytes, `v_dot2c_f32_f16`) together with a
// RIGOROUS bound E >= |s - r|:
//
//     |s - r| <= |s - <n~,x~>|  +  |<n~,x~> - <n,x>|            +  |<n,x> - r|
//             <= gamma_s |n~||x~| + (|n - n~||x~| + |n||x - x~|)  +  gamma_r |n||x|          (Cauchy-Schwarz)
//
// where the six 2-norms are MEASURED when the copies are made (rounded up), gamma_r bounds the rounding error of the
// reference's f32 reduction (dims/32 chained FMAs, the hsum tree, the scalar tail) and gamma_s that of the screen's own
// f32 accumulation, both with a safety factor.  If |s| > E then r != 0 and sign(r) = sign(s): the side is decided
// without touching the f32 data.  Otherwise (about 1 % of the pairs for 768-d data; always, for rows whose values
// overflow or vanish in binary16: their measured error norm is inf / large) the SAME kernel falls back to the reference
// arithmetic for that pair.  Sides are therefore identical to the f32-only kernels, bit for bit, by construction; the
// parity tests run every forest in both modes, and AH_SCREEN_VERIFY=1 makes the kernels evaluate both values for
// EVERY pair and count |s - r| > E violations (must be 0).
#pragma once

#include "device_math.h"
#include "split_device.h"

namespace ah {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// f32 -> binary16 for the shadow copies: round to nearest even; values that would be binary16 subnormals become 0 (the
// measured error norm accounts for it), so the screen never depends on how v_dot2c treats subnormal inputs.
__device__ __forceinline__ _Float16 to_shadow_half(float x) {
    _Float16 h = (_Float16)x;
    if (fabsf((float)h) < 6.103515625e-05f) h = (_Float16)0.0f;  // NaN stays NaN, inf stays inf (-> fallback)
    return h;
}
// Rows whose largest |x| is below 2^-40 (and not zero) are never decided by a screen: the f32 sums of squares behind the
// measured norms underflow there (at 1e-23 they collapse to 0 and a bound built from them would let the bias alone decide
// a Euclidean margin).  Their stats are +inf, i.e. every pair with such a row takes the reference arithmetic.
constexpr uint32_t kTinyBits = 0x2B800000u;  // 2^-40

// 8 halves (16 bytes) x 8 halves -> f32 accumulate: 4 x v_dot2c_f32_f16
__device__ __forceinline__ float screen_dot8(const uint4 a, const uint4 b, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.x), __builtin_bit_cast(f16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.y), __builtin_bit_cast(f16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.z), __builtin_bit_cast(f16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.w), __builtin_bit_cast(f16x2_t, b.w), acc, false);
    return acc;
}

// int8 copies (forest.hip: k_shadow_rows8; search.hip: the queries' digits): round to nearest, clamped to +-127
__device__ __forceinline__ int quantize8(float x, float inv_scale) {
    const float q = rintf(x * inv_scale);
    return (int)fminf(fmaxf(q, -127.0f), 127.0f);
}
// 16 int8 (16 bytes) x 16 int8 -> i32 accumulate: 4 x v_dot4_i32_i8, exact
__device__ __forceinline__ int dot16_i8(const uint4 a, const uint4 b, int acc) {
    acc = __builtin_amdgcn_sdot4((int)a.x, (int)b.x, acc, false);
    acc = __builtin_amdgcn_sdot4((int)a.y, (int)b.y, acc, false);
    acc = __builtin_amdgcn_sdot4((int)a.z, (int)b.z, acc, false);
    acc = __builtin_amdgcn_sdot4((int)a.w, (int)b.w, acc, false);
    return acc;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream_u4(const uint4 *p) {
    u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// sum over the 8 lanes of an octet (any order: the screen's accumulation error is covered by gamma_s)
__device__ __forceinline__ float octet_sum(float v) {
    v += dpp_f32<0xB1>(v);   // lane ^ 1
    v += dpp_f32<0x4E>(v);   // lane ^ 2: every lane of a quad now holds the quad's sum
    v += dpp_f32<0x141>(v);  // + the other quad of the octet (half-row mirror)
    return v;
}

// Transposing reduction over an octet.  On entry lane j holds its partial sums v[0..N) for N trees; on exit every lane
// holds the sum over the 8 lanes for ONE tree, octet_owned_tree<N>(j).  Three butterfly steps (half-row mirror, lane ^ 2,
// lane ^ 1), each lane keeping the half of its values that its side of the exchange owns: 7 N / 2 VALU operations instead
// of the 6 N of N separate octet sums — and, more importantly, the per-tree epilogue (error bound, decision, store)
// then runs ONCE per tree on its owner lane instead of 8 times, and its loads / stores are issued for 8 trees at once.
template <int N>
__device__ __forceinline__ uint32_t octet_owned_tree(uint32_t j) {
    return N == 8 ? j : N == 4 ? ((j >> 2) * 2u + ((j >> 1) & 1u)) : (j >> 2);
}
template <int N>
__device__ __forceinline__ bool octet_is_owner(uint32_t j) {  // N < 8: several lanes end up with the same tree
    return N == 8 ? true : N == 4 ? (j & 1u) == 0u : (j & 3u) == 0u;
}
template <int N>
__host__ __device__ constexpr uint32_t octet_owner_lane(uint32_t t) {
    return N == 8 ? t : N == 4 ? ((t >> 1) * 4u + (t & 1u) * 2u) : t * 4u;
}
template <int N>
__device__ __forceinline__ float octet_transpose_sum(const float *v, uint32_t j) {
    const bool hi = (j & 4u) != 0u, b1 = (j & 2u) != 0u, b0 = (j & 1u) != 0u;
    if (N == 8) {
        float w[4], u[2];
#pragma unroll
        for (int t = 0; t < 4; t++) w[t] = (hi ? v[t + 4] : v[t]) + dpp_f32<0x141>(hi ? v[t] : v[t + 4]);
#pragma unroll
        for (int t = 0; t < 2; t++) u[t] = (b1 ? w[t + 2] : w[t]) + dpp_f32<0x4E>(b1 ? w[t] : w[t + 2]);
        return (b0 ? u[1] : u[0]) + dpp_f32<0xB1>(b0 ? u[0] : u[1]);
    } else if (N == 4) {
        float w[2];
#pragma unroll
        for (int t = 0; t < 2; t++) w[t] = (hi ? v[t + 2] : v[t]) + dpp_f32<0x141>(hi ? v[t] : v[t + 2]);
        float r = (b1 ? w[1] : w[0]) + dpp_f32<0x4E>(b1 ? w[0] : w[1]);
        return r + dpp_f32<0xB1>(r);
    } else {
        float r = (hi ? v[1] : v[0]) + dpp_f32<0x141>(hi ? v[0] : v[1]);
        r += dpp_f32<0x4E>(r);
        return r + dpp_f32<0xB1>(r);
    }
}

// Per-normal record of the shadow chunk: [hpitch halves][stats: |n~|, |n - n~|, |n|, B] with B the additive term of the
// margin that does not depend on the item (bias of Euclidean / Manhattan; 0 for Cosine; for DotProduct the normal's
// extra dimension h0, which multiplies the item's).
struct NormalStats {
    float an, bn, cn, extra;
};

// Bound E on |screen margin - reference margin| and the screen margin itself.
//   s_dot  = screen dot product; rs = row stats {|x~|, |x - x~|, |x|}; ns = normal stats
//   row_extra = the item's extra dimension (DotProduct) else unused
template <int METRIC>
__device__ __forceinline__ bool screen_decides(float s_dot, const float4 rs, const NormalStats ns, float row_extra,
                                               float gamma_s, float gamma_r, uint32_t &side) {
    // E_dot: all terms non-negative; evaluated in f32 and inflated by 2^-9 to cover the rounding of E itself
    float e = ns.bn * rs.x + ns.cn * rs.y + gamma_s * (ns.an * rs.x) + gamma_r * (ns.cn * rs.z);
    float m = s_dot;
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) {
        // r = fl(bias + fl_ref(dot)): one more rounding of a value bounded by |bias| + |dot| (euclidean.rs:79-81)
        m = ns.extra + s_dot;
        e += 2.4e-7f * (fabsf(ns.extra) + fabsf(s_dot) + e);
    } else if (METRIC == AH_DOT_PRODUCT) {
        // r = fl(fl_ref(dot) + fl(h0 * e_item)) (dot_product.rs:115-117): the product and the sum round once each
        const float p = ns.extra * row_extra;
        m = s_dot + p;
        e += 2.4e-7f * (2.0f * fabsf(p) + fabsf(s_dot) + e);
    }
    e = e * 1.002f + 1e-30f;
    side = (__float_as_uint(m) >> 31) ^ 1u;
    return fabsf(m) > e;  // false for NaN / inf bounds: those pairs take the reference path
}

}  // namespace ah
