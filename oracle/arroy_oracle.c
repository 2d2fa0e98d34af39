/*
 * arroy_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the arithmetic of arroy's distance hot path, written from the
 * reference's behaviour (meilisearch/arroy v0.7.0, /root/reference), each function citing the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product (arroy_amd/, libarroy_hip.so) never
 * does and has no CPU fallback.
 *
 * PARITY PINNING.  The reference is Rust and cannot be compiled in this environment (no
 * cargo/rustc/LMDB), so this oracle is pinned against the reference's own golden vectors
 * instead (tests/test_oracle_golden.py):
 *   - src/spaces/simple_avx.rs:120-144 and simple_sse.rs:120-142 (SIMD == scalar on literals),
 *   - src/tests/upgrade.rs:58-67,116-128 with the binary fixtures src/tests/assets/v0_6/{smol,large}.mdb
 *     (Euclidean distances 2.4881108 / 2.5068686 / 2.5809734, ids 92/24/78; ties by id),
 *   - src/unaligned_vector/binary_quantized_test.rs:11-167 (1-bit codec bit patterns),
 *   - src/tests/reader.rs:81-144 (cosine of zero-ish item, exact distances on a line),
 *   - src/tests/writer.rs:266-293 (normal [0.5774]*3 of a 3-d split).
 * What the reference's tests do NOT pin — numeric cosine / dot / Manhattan / BQ distance
 * values and the RNG-driven tree shape — is "parity unpinned by golden vectors" and rests on
 * the line-by-line restatement below plus a float64 cross-check.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mavx2 -mfma -fopenmp).
 * -ffp-contract=off matters: Rust never contracts a*b+c; FMAs appear only where the reference
 * writes _mm256_fmadd_ps.
 */
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/arroy_hip.h"
#include "../include/arroy_hip_policy.h"

#define AO_API __attribute__((visibility("default")))
/* Loops shorter than this run serially: a fork/join over 128+ host threads per tiny tree node costs more
 * than the node (and was observed to stall for minutes on a 256-thread GPU host). */
#define AO_PAR_MIN 4096

/* ---------------------------------------------------------------------------------------
 * Tier selection: which of the reference's runtime-dispatched code paths is "the reference"
 * (src/spaces/simple.rs:19-45,53-79).  0 = x86-64 with AVX+FMA (the benchmark host),
 * 1 = SSE only, 2 = no SIMD.
 * ------------------------------------------------------------------------------------- */
static int g_tier = 0;
static int g_use_intrinsics = -1; /* -1 unknown, 0 emulate, 1 real AVX2+FMA instructions */

AO_API void ao_set_tier(int tier) { g_tier = tier; }
AO_API int ao_get_tier(void) { return g_tier; }
AO_API void ao_set_use_intrinsics(int on) { g_use_intrinsics = on; }

static int have_avx_fma(void) {
    if (g_use_intrinsics < 0) {
        __builtin_cpu_init();
        g_use_intrinsics = (__builtin_cpu_supports("avx") && __builtin_cpu_supports("fma")) ? 1 : 0;
    }
    return g_use_intrinsics;
}
AO_API int ao_cpu_has_avx_fma(void) {
    __builtin_cpu_init();
    return (__builtin_cpu_supports("avx") && __builtin_cpu_supports("fma")) ? 1 : 0;
}

/* ---------------------------------------------------------------------------------------
 * spaces: scalar tier.  src/spaces/simple.rs:49-51 and :81-83 (`Iterator::sum::<f32>()` is a
 * left-to-right accumulation starting from 0.0... strictly, from the first element added to
 * -0.0/0.0; Rust's f32 Sum folds from 0.0 with `+`).
 * ------------------------------------------------------------------------------------- */
AO_API float ao_dot_scalar(const float *u, const float *v, size_t n) {
    float r = 0.0f;
    for (size_t i = 0; i < n; i++) r = r + u[i] * v[i];
    return r;
}
AO_API float ao_euclid_scalar(const float *u, const float *v, size_t n) {
    float r = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float s = u[i] - v[i];
        r = r + s * s;
    }
    return r;
}

/* ---------------------------------------------------------------------------------------
 * spaces: SSE tier, emulated lane by lane.  src/spaces/simple_sse.rs:10-14 (hsum128),
 * :17-61 (euclid), :64-110 (dot).  16 chains = 4 accumulators x 4 lanes; mul THEN add
 * (`_mm_add_ps(_mm_mul_ps(..), acc)`), never fused.
 * ------------------------------------------------------------------------------------- */
static float hsum128_emul(const float *x) {
    /* x64 = x + movehl(x,x): x64[0]=x[0]+x[2], x64[1]=x[1]+x[3]; x32 = x64[0] + x64[1] */
    float a = x[0] + x[2];
    float b = x[1] + x[3];
    return a + b;
}
AO_API float ao_dot_sse(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 16);
    float c[16];
    for (int j = 0; j < 16; j++) c[j] = 0.0f;
    for (size_t i = 0; i < m; i += 16)
        for (int j = 0; j < 16; j++) {
            float p = u[i + j] * v[i + j];
            c[j] = p + c[j];
        }
    float r = hsum128_emul(c) + hsum128_emul(c + 4) + hsum128_emul(c + 8) + hsum128_emul(c + 12);
    for (size_t i = m; i < n; i++) r = r + u[i] * v[i];
    return r;
}
AO_API float ao_euclid_sse(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 16);
    float c[16];
    for (int j = 0; j < 16; j++) c[j] = 0.0f;
    for (size_t i = 0; i < m; i += 16)
        for (int j = 0; j < 16; j++) {
            float s = u[i + j] - v[i + j];
            float p = s * s;
            c[j] = p + c[j];
        }
    float r = hsum128_emul(c) + hsum128_emul(c + 4) + hsum128_emul(c + 8) + hsum128_emul(c + 12);
    for (size_t i = m; i < n; i++) {
        float s = u[i] - v[i];
        r = r + s * s; /* (a - b).powi(2): x*x, then += */
    }
    return r;
}

/* ---------------------------------------------------------------------------------------
 * spaces: AVX+FMA tier.  src/spaces/simple_avx.rs:8-13 (hsum256), :17-65 (euclid), :69-110 (dot).
 * 32 chains = 4 accumulators x 8 lanes, element i feeds chain i mod 32, fused multiply-add.
 * Two implementations that must agree bit for bit: a lane-by-lane emulation with fmaf()
 * (documents the order; runs anywhere) and the real instruction sequence (fast; used for the
 * CPU baseline timing).
 * ------------------------------------------------------------------------------------- */
static float hsum256_emul(const float *x) {
    /* x128[j] = x[j+4] + x[j]; x64[j] = x128[j] + x128[j+2]; x32 = x64[0] + x64[1] */
    float x128[4];
    for (int j = 0; j < 4; j++) x128[j] = x[j + 4] + x[j];
    float a = x128[0] + x128[2];
    float b = x128[1] + x128[3];
    return a + b;
}
AO_API float ao_dot_avx_emul(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 32);
    float c[32];
    for (int j = 0; j < 32; j++) c[j] = 0.0f;
    for (size_t i = 0; i < m; i += 32)
        for (int j = 0; j < 32; j++) c[j] = fmaf(u[i + j], v[i + j], c[j]);
    float r = hsum256_emul(c) + hsum256_emul(c + 8) + hsum256_emul(c + 16) + hsum256_emul(c + 24);
    for (size_t i = m; i < n; i++) r = r + u[i] * v[i];
    return r;
}
AO_API float ao_euclid_avx_emul(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 32);
    float c[32];
    for (int j = 0; j < 32; j++) c[j] = 0.0f;
    for (size_t i = 0; i < m; i += 32)
        for (int j = 0; j < 32; j++) {
            float s = u[i + j] - v[i + j];
            c[j] = fmaf(s, s, c[j]);
        }
    float r = hsum256_emul(c) + hsum256_emul(c + 8) + hsum256_emul(c + 16) + hsum256_emul(c + 24);
    for (size_t i = m; i < n; i++) {
        float s = u[i] - v[i];
        r = r + s * s;
    }
    return r;
}

__attribute__((target("avx,fma"))) static float hsum256_real(__m256 x) {
    __m128 x128 = _mm_add_ps(_mm256_extractf128_ps(x, 1), _mm256_castps256_ps128(x));
    __m128 x64 = _mm_add_ps(x128, _mm_movehl_ps(x128, x128));
    __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
    return _mm_cvtss_f32(x32);
}
__attribute__((target("avx,fma"))) AO_API float ao_dot_avx_real(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 32);
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(u + i), _mm256_loadu_ps(v + i), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(u + i + 8), _mm256_loadu_ps(v + i + 8), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(u + i + 16), _mm256_loadu_ps(v + i + 16), s3);
        s4 = _mm256_fmadd_ps(_mm256_loadu_ps(u + i + 24), _mm256_loadu_ps(v + i + 24), s4);
    }
    float r = hsum256_real(s1) + hsum256_real(s2) + hsum256_real(s3) + hsum256_real(s4);
    for (size_t i = m; i < n; i++) r = r + u[i] * v[i];
    return r;
}
__attribute__((target("avx,fma"))) AO_API float ao_euclid_avx_real(const float *u, const float *v, size_t n) {
    size_t m = n - (n % 32);
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(u + i), _mm256_loadu_ps(v + i));
        s1 = _mm256_fmadd_ps(d1, d1, s1);
        __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(u + i + 8), _mm256_loadu_ps(v + i + 8));
        s2 = _mm256_fmadd_ps(d2, d2, s2);
        __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(u + i + 16), _mm256_loadu_ps(v + i + 16));
        s3 = _mm256_fmadd_ps(d3, d3, s3);
        __m256 d4 = _mm256_sub_ps(_mm256_loadu_ps(u + i + 24), _mm256_loadu_ps(v + i + 24));
        s4 = _mm256_fmadd_ps(d4, d4, s4);
    }
    float r = hsum256_real(s1) + hsum256_real(s2) + hsum256_real(s3) + hsum256_real(s4);
    for (size_t i = m; i < n; i++) {
        float s = u[i] - v[i];
        r = r + s * s;
    }
    return r;
}

/* Runtime dispatch, src/spaces/simple.rs:19-45 and :53-79 (thresholds 32 and 16). */
AO_API float ao_dot(const float *u, const float *v, size_t n) {
    if (g_tier == 0 && n >= 32) return have_avx_fma() ? ao_dot_avx_real(u, v, n) : ao_dot_avx_emul(u, v, n);
    if (g_tier <= 1 && n >= 16) return ao_dot_sse(u, v, n);
    return ao_dot_scalar(u, v, n);
}
AO_API float ao_euclid(const float *u, const float *v, size_t n) {
    if (g_tier == 0 && n >= 32) return have_avx_fma() ? ao_euclid_avx_real(u, v, n) : ao_euclid_avx_emul(u, v, n);
    if (g_tier <= 1 && n >= 16) return ao_euclid_sse(u, v, n);
    return ao_euclid_scalar(u, v, n);
}

/* ---------------------------------------------------------------------------------------
 * 1-bit codec.  src/unaligned_vector/binary_quantized.rs:80-91 (quantize: bit i of word w =
 * is_sign_positive(x[64w+i]); last word zero padded), :261-290 (iterator: bit -> bit*2-1, over
 * whole words, so padding decodes to -1.0), :67-69 (len rounds up to 64).
 * ------------------------------------------------------------------------------------- */
AO_API size_t ao_bq_bytes(size_t dims) { return ((dims + 63) / 64) * 8; }

AO_API void ao_bq_quantize(const float *x, size_t dims, uint8_t *out) {
    size_t words = (dims + 63) / 64;
    for (size_t w = 0; w < words; w++) {
        uint64_t word = 0;
        size_t lo = w * 64, hi = lo + 64 < dims ? lo + 64 : dims;
        for (size_t i = hi; i-- > lo;) { /* chunk.iter().rev(): word <<= 1; word += sign_positive */
            uint32_t bits;
            memcpy(&bits, &x[i], 4);
            word <<= 1;
            word += (uint64_t)((bits >> 31) == 0); /* is_sign_positive: sign bit clear (+0.0, +NaN -> 1) */
        }
        memcpy(out + 8 * w, &word, 8); /* to_ne_bytes on little-endian x86-64 */
    }
}
AO_API void ao_bq_dequantize(const uint8_t *bytes, size_t nbytes, float *out) {
    for (size_t w = 0; w < nbytes / 8; w++) {
        uint64_t word;
        memcpy(&word, bytes + 8 * w, 8);
        for (int i = 0; i < 64; i++) {
            uint64_t bit = word & 1;
            word >>= 1;
            out[64 * w + i] = (float)bit * 2.0f - 1.0f;
        }
    }
}
/* src/spaces/simple.rs:119-131: per byte, popcnt(!(u^v)) - popcnt0(!(u^v)), summed in i32. */
AO_API int32_t ao_bq_dot_i32(const uint8_t *u, const uint8_t *v, size_t nbytes) {
    int32_t s = 0;
    for (size_t i = 0; i < nbytes; i++) {
        uint8_t r = (uint8_t) ~(u[i] ^ v[i]);
        int ones = __builtin_popcount(r);
        s += ones - (8 - ones);
    }
    return s;
}
/* src/distance/binary_quantized_euclidean.rs:117-124 / binary_quantized_manhattan.rs:113-120. */
AO_API uint32_t ao_bq_hamming(const uint8_t *u, const uint8_t *v, size_t nbytes) {
    uint32_t s = 0;
    for (size_t i = 0; i < nbytes; i++) s += (uint32_t)__builtin_popcount((uint8_t)(u[i] ^ v[i]));
    return s;
}

/* ---------------------------------------------------------------------------------------
 * Distance trait, per metric.  Headers are float[2]: {bias} | {norm} | {extra_dim, norm}
 * (repr(C) structs in src/distance/<metric>.rs).
 * ------------------------------------------------------------------------------------- */
static int is_bq(int metric) { return metric >= AH_BQ_EUCLIDEAN; }
AO_API size_t ao_header_floats(int metric) { return metric == AH_DOT_PRODUCT ? 2 : 1; }
AO_API size_t ao_vector_bytes(int metric, size_t dims) { return is_bq(metric) ? ao_bq_bytes(dims) : 4 * dims; }

/* D::norm_no_header */
AO_API float ao_norm_no_header(int metric, const void *v, size_t dims) {
    switch (metric) {
    case AH_EUCLIDEAN:   /* euclidean.rs:49-51 */
    case AH_MANHATTAN:   /* manhattan.rs:52-54 */
    case AH_COSINE:      /* cosine.rs:65-67 */
    case AH_DOT_PRODUCT: /* dot_product.rs:77-79 */
        return sqrtf(ao_dot((const float *)v, (const float *)v, dims));
    case AH_BQ_EUCLIDEAN: /* binary_quantized_euclidean.rs:60-62 */
    case AH_BQ_COSINE:    /* binary_quantized_cosine.rs:73-75 */
        return sqrtf((float)ao_bq_dot_i32((const uint8_t *)v, (const uint8_t *)v, ao_bq_bytes(dims)));
    case AH_BQ_MANHATTAN: { /* binary_quantized_manhattan.rs:59-66: sqrt(ones - zeros) */
        const uint8_t *b = (const uint8_t *)v;
        int32_t s = 0;
        for (size_t i = 0; i < ao_bq_bytes(dims); i++) {
            int ones = __builtin_popcount(b[i]);
            s += ones - (8 - ones);
        }
        return sqrtf((float)s);
    }
    }
    return NAN;
}

/* D::new_header (cosine.rs:39-41, dot_product.rs:47-50, euclidean.rs:41-43, ...) */
AO_API void ao_new_header(int metric, const void *v, size_t dims, float *hdr) {
    hdr[0] = 0.0f;
    if (metric == AH_DOT_PRODUCT) hdr[1] = 0.0f;
    if (metric == AH_COSINE || metric == AH_BQ_COSINE) hdr[0] = ao_norm_no_header(metric, v, dims);
}

/* D::norm(leaf): default = norm_no_header (mod.rs:70-72); DotProduct adds extra_dim (dot_product.rs:72-75). */
AO_API float ao_norm(int metric, const void *v, const float *hdr, size_t dims) {
    if (metric == AH_DOT_PRODUCT) {
        float dot = ao_dot((const float *)v, (const float *)v, dims);
        float e2 = hdr[0] * hdr[0];
        return sqrtf(dot + e2);
    }
    return ao_norm_no_header(metric, v, dims);
}

/* D::built_distance */
AO_API float ao_built_distance(int metric, const void *pv, const float *ph, const void *qv, const float *qh,
                               size_t dims) {
    switch (metric) {
    case AH_EUCLIDEAN: /* euclidean.rs:45-47 */
        return ao_euclid((const float *)pv, (const float *)qv, dims);
    case AH_MANHATTAN: { /* manhattan.rs:44-46: sequential sum of |p-q| */
        const float *p = (const float *)pv, *q = (const float *)qv;
        float r = 0.0f;
        for (size_t i = 0; i < dims; i++) r = r + fabsf(p[i] - q[i]);
        return r;
    }
    case AH_COSINE: { /* cosine.rs:43-59 */
        float pn = ph[0], qn = qh[0];
        float pq = ao_dot((const float *)pv, (const float *)qv, dims);
        float pnqn = pn * qn;
        if (pnqn > 1.1920929e-7f /* f32::EPSILON */) {
            float c = pq / pnqn;
            /* f32::clamp(-1,1): NaN stays NaN */
            if (c < -1.0f) c = -1.0f;
            if (c > 1.0f) c = 1.0f;
            return (1.0f - c) / 2.0f;
        }
        return 0.0f;
    }
    case AH_DOT_PRODUCT: /* dot_product.rs:52-56 */
        return -ao_dot((const float *)pv, (const float *)qv, dims);
    case AH_BQ_EUCLIDEAN:
        return (float)(ao_bq_hamming((const uint8_t *)pv, (const uint8_t *)qv, ao_bq_bytes(dims)) * 4u);
    case AH_BQ_MANHATTAN:
        return (float)(ao_bq_hamming((const uint8_t *)pv, (const uint8_t *)qv, ao_bq_bytes(dims)) * 2u);
    case AH_BQ_COSINE: { /* binary_quantized_cosine.rs:49-64: guard is != 0, no clamp */
        float pn = ph[0], qn = qh[0];
        float pq = (float)ao_bq_dot_i32((const uint8_t *)pv, (const uint8_t *)qv, ao_bq_bytes(dims));
        float pnqn = pn * qn;
        if (pnqn != 0.0f) {
            float c = pq / pnqn;
            return (1.0f - c) / 2.0f;
        }
        return 0.0f;
    }
    }
    return NAN;
}

/* D::non_built_distance: default = built_distance (mod.rs:54-56); DotProduct dot_product.rs:58-70. */
AO_API float ao_non_built_distance(int metric, const void *pv, const float *ph, const void *qv, const float *qh,
                                   size_t dims) {
    if (metric == AH_DOT_PRODUCT) {
        float pp = ph[1], qq = qh[1];
        float ee = ph[0] * qh[0];
        float pq = ao_dot((const float *)pv, (const float *)qv, dims) + ee;
        float ppqq = pp * qq;
        if (ppqq >= 1.17549435e-38f /* f32::MIN_POSITIVE */) {
            float t = 2.0f * pq;
            return 2.0f - t / sqrtf(ppqq);
        }
        return 2.0f;
    }
    return ao_built_distance(metric, pv, ph, qv, qh, dims);
}

/* D::normalized_distance (mod.rs:59-61 default sqrt; cosine.rs:61-63; dot_product.rs:81-83;
 * manhattan.rs:48-50; bq_euclidean.rs:56-58; bq_manhattan.rs:55-57; bq_cosine.rs:67-69). */
AO_API float ao_normalized_distance(int metric, float d, size_t dims) {
    switch (metric) {
    case AH_EUCLIDEAN: return sqrtf(d);
    case AH_MANHATTAN: return fmaxf(d, 0.0f);
    case AH_COSINE: return d;
    case AH_DOT_PRODUCT: return -d;
    case AH_BQ_EUCLIDEAN: return d / (float)dims;
    case AH_BQ_MANHATTAN: return fmaxf(d, 0.0f) / (float)dims;
    case AH_BQ_COSINE: return d;
    }
    return NAN;
}

/* D::margin (euclidean.rs:79-81, manhattan.rs:82-84, cosine.rs:87-89, dot_product.rs:115-117, bq_*.rs) */
AO_API float ao_margin(int metric, const void *nv, const float *nh, const void *qv, const float *qh, size_t dims) {
    switch (metric) {
    case AH_EUCLIDEAN:
    case AH_MANHATTAN:
        return nh[0] + ao_dot((const float *)nv, (const float *)qv, dims);
    case AH_COSINE:
        return ao_dot((const float *)nv, (const float *)qv, dims);
    case AH_DOT_PRODUCT: {
        float ee = nh[0] * qh[0];
        return ao_dot((const float *)nv, (const float *)qv, dims) + ee;
    }
    case AH_BQ_EUCLIDEAN:
    case AH_BQ_MANHATTAN:
        return nh[0] + (float)ao_bq_dot_i32((const uint8_t *)nv, (const uint8_t *)qv, ao_bq_bytes(dims));
    case AH_BQ_COSINE:
        return (float)ao_bq_dot_i32((const uint8_t *)nv, (const uint8_t *)qv, ao_bq_bytes(dims));
    }
    return NAN;
}
/* D::side (mod.rs:103-110): Right (1) iff margin.is_sign_positive(), i.e. sign bit clear. */
AO_API int ao_side_of_margin(float margin) {
    uint32_t bits;
    memcpy(&bits, &margin, 4);
    return (bits >> 31) == 0 ? 1 : 0;
}
/* D::pq_distance (mod.rs:63-68). Rust f32::min returns the non-NaN operand. side: 0 left, 1 right. */
AO_API float ao_pq_distance(float distance, float margin, int side) {
    float m = side == 0 ? -margin : margin;
    return fminf(m, distance);
}

/* ---------------------------------------------------------------------------------------
 * Data set view used by the batched entry points.
 * ------------------------------------------------------------------------------------- */
typedef struct ao_data {
    int metric;
    uint32_t dims;
    uint64_t n;
    const uint8_t *vectors; /* n rows of ao_vector_bytes(metric,dims) bytes, contiguous */
    float *headers;         /* n * ao_header_floats(metric) */
    const uint32_t *ids;    /* NULL: id == row */
} ao_data;

static const void *row_vec(const ao_data *d, uint64_t row) {
    return d->vectors + row * ao_vector_bytes(d->metric, d->dims);
}
static const float *row_hdr(const ao_data *d, uint64_t row) { return d->headers + row * ao_header_floats(d->metric); }

/* D::new_header for every row (Writer::add_item, src/writer.rs:380-394), parallel over rows. */
AO_API void ao_new_headers(ao_data *d) {
    size_t hf = ao_header_floats(d->metric);
#pragma omp parallel for schedule(static) if (d->n >= AO_PAR_MIN)
    for (int64_t r = 0; r < (int64_t)d->n; r++) ao_new_header(d->metric, row_vec(d, (uint64_t)r), d->dims, d->headers + (size_t)r * hf);
}

/* The re-rank loop, src/reader.rs:381-391 (rows == NULL: rows 0..n-1). OpenMP over rows = the
 * rayon-style CPU baseline. */
AO_API void ao_distances(const ao_data *d, const void *qv, const float *qh, const uint32_t *rows, uint64_t n,
                         float *out) {
#pragma omp parallel for schedule(static) if (n >= AO_PAR_MIN)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        uint64_t r = rows ? rows[i] : (uint64_t)i;
        out[i] = ao_built_distance(d->metric, qv, qh, row_vec(d, r), row_hdr(d, r), d->dims);
    }
}

/* OrderedFloat<f32> total order (ordered-float 4.6, un-vendored dependency, Cargo.toml:22): NaN is
 * greater than everything and equal to itself; -0.0 == +0.0. Tuples compare lexicographically. */
static int of_cmp(float a, float b) {
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn; /* NaN == NaN, NaN > x */
    return (a > b) - (a < b);
}
typedef struct pair_t {
    float d;
    uint32_t id;
} pair_t;
static int pair_cmp(const void *x, const void *y) {
    const pair_t *a = (const pair_t *)x, *b = (const pair_t *)y;
    int c = of_cmp(a->d, b->d);
    if (c) return c;
    return (a->id > b->id) - (a->id < b->id);
}

/* median_based_top_k, src/reader.rs:607-640, statement by statement.  `select_nth_unstable(k-1)` +
 * `truncate(k)` keeps the k smallest of the buffer (a set, order irrelevant because of the final
 * sort); it is restated with a full sort of the 2k buffer.  k == 0 (a panic in the reference when
 * the input is non-empty) returns 0 items. */
AO_API size_t ao_top_k(const float *dists, const uint32_t *ids, size_t n, size_t k, uint32_t *out_ids,
                       float *out_dists) {
    if (k == 0 || n == 0) return 0;
    pair_t threshold = {3.40282347e+38f /* f32::MAX */, 0xFFFFFFFFu};
    pair_t *buf = (pair_t *)malloc(sizeof(pair_t) * 2 * k);
    size_t len = 0, i = 0;
    for (; i < n && len < 2 * k; i++) { /* prefill with no threshold checks */
        buf[len].d = dists[i];
        buf[len].id = ids[i];
        len++;
    }
    for (; i < n; i++) {
        pair_t item = {dists[i], ids[i]};
        if (pair_cmp(&item, &threshold) >= 0) continue;
        if (len == 2 * k) {
            qsort(buf, len, sizeof(pair_t), pair_cmp);
            threshold = buf[k - 1];
            len = k;
        }
        buf[len++] = item;
    }
    qsort(buf, len, sizeof(pair_t), pair_cmp);
    if (len > k) len = k;
    for (size_t j = 0; j < len; j++) {
        out_ids[j] = buf[j].id;
        out_dists[j] = buf[j].d;
    }
    free(buf);
    return len;
}

/* binary_heap_based_top_k of the reference's proptest (src/tests/reader.rs:283-299 compares the two):
 * here simply "sort everything, take k" — the specification median_based_top_k is tested against. */
AO_API size_t ao_top_k_spec(const float *dists, const uint32_t *ids, size_t n, size_t k, uint32_t *out_ids,
                            float *out_dists) {
    pair_t *buf = (pair_t *)malloc(sizeof(pair_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        buf[i].d = dists[i];
        buf[i].id = ids[i];
    }
    qsort(buf, n, sizeof(pair_t), pair_cmp);
    size_t len = k < n ? k : n;
    for (size_t j = 0; j < len; j++) {
        out_ids[j] = buf[j].id;
        out_dists[j] = buf[j].d;
    }
    free(buf);
    return len;
}

/* nns_by_leaf after the tree descent, src/reader.rs:376-400: rows ascending & unique. */
AO_API size_t ao_rerank(const ao_data *d, const void *qv, const float *qh, const uint32_t *rows, uint64_t n, size_t k,
                        uint32_t *out_ids, float *out_dists) {
    float *dist = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    ao_distances(d, qv, qh, rows, n, dist);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t r = rows ? rows[i] : i;
        ids[i] = d->ids ? d->ids[r] : (uint32_t)r;
    }
    size_t kk = k < n ? k : (size_t)n;
    size_t m = ao_top_k(dist, ids, n, kk, out_ids, out_dists);
    for (size_t j = 0; j < m; j++) out_dists[j] = ao_normalized_distance(d->metric, out_dists[j], d->dims);
    free(dist);
    free(ids);
    return m;
}

/* The margin loop, src/writer.rs:1201-1207: sides[i] = 1 for Right. */
AO_API void ao_split_sides(const ao_data *d, const void *nv, const float *nh, const uint32_t *rows, uint64_t n,
                           uint8_t *sides, uint64_t *n_left, float *margins) {
    uint64_t left = 0;
#pragma omp parallel for schedule(static) reduction(+ : left) if (n >= AO_PAR_MIN)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        uint64_t r = rows ? rows[i] : (uint64_t)i;
        float m = ao_margin(d->metric, nv, nh, row_vec(d, r), row_hdr(d, r), d->dims);
        int s = ao_side_of_margin(m);
        sides[i] = (uint8_t)s;
        if (margins) margins[i] = m;
        left += (uint64_t)(1 - s);
    }
    *n_left = left;
}

/* DotProduct::preprocess, src/distance/dot_product.rs:119-165. */
AO_API void ao_preprocess_dot(ao_data *d, float *out_max_norm) {
    float max_norm = 0.0f;
    for (uint64_t r = 0; r < d->n; r++) {
        float norm = ao_norm_no_header(AH_DOT_PRODUCT, row_vec(d, r), d->dims);
        max_norm = fmaxf(max_norm, norm); /* f32::max: ignores NaN */
    }
    float m2 = max_norm * max_norm;
    for (uint64_t r = 0; r < d->n; r++) {
        float node_norm = ao_norm_no_header(AH_DOT_PRODUCT, row_vec(d, r), d->dims);
        float n2 = node_norm * node_norm;
        float diff = m2 - n2;
        d->headers[2 * r + 1] = m2;
        d->headers[2 * r + 0] = sqrtf(diff);
    }
    *out_max_norm = max_norm;
}

/* ---------------------------------------------------------------------------------------
 * two_means / create_split.  src/distance/mod.rs:126-171 (f32 metrics) and :173-223 (BQ: every
 * sampled leaf is de-quantised to +-1.0 and given a header by the NON-BQ metric's new_header).
 * `sample_rows[0..1]` = choose_two, [2..11] = the ten `choose` draws.
 * ------------------------------------------------------------------------------------- */
static int f32_space_metric(int metric) {
    switch (metric) {
    case AH_BQ_EUCLIDEAN: return AH_EUCLIDEAN;
    case AH_BQ_MANHATTAN: return AH_MANHATTAN;
    case AH_BQ_COSINE: return AH_COSINE;
    }
    return metric;
}
static int uses_cosine_two_means(int metric) { /* the `cosine` flag passed to two_means */
    return metric == AH_COSINE || metric == AH_DOT_PRODUCT || metric == AH_BQ_COSINE;
}
/* D::init (cosine.rs:69-71 norm = sqrt(dot); dot_product.rs:94-96 norm = dot (squared!); else no-op) */
static void leaf_init(int m, const float *v, float *hdr, size_t dims) {
    if (m == AH_COSINE) hdr[0] = sqrtf(ao_dot(v, v, dims));
    if (m == AH_DOT_PRODUCT) hdr[1] = ao_dot(v, v, dims);
}
/* D::normalize (mod.rs:76-82; dot_product.rs:85-92 also divides extra_dim) */
static void leaf_normalize(int m, float *v, float *hdr, size_t dims) {
    float norm = ao_norm(m, v, hdr, dims);
    if (norm > 0.0f) {
        for (size_t i = 0; i < dims; i++) v[i] = v[i] / norm;
        if (m == AH_DOT_PRODUCT) hdr[0] = hdr[0] / norm;
    }
}
/* D::update_mean (mod.rs:86-94): (x * c + n / norm) / (c + 1.0), each op rounded. */
static void leaf_update_mean(float *mean, const float *nw, float norm, float c, size_t dims) {
    float c1 = c + 1.0f;
    for (size_t i = 0; i < dims; i++) {
        float a = mean[i] * c;
        float b = nw[i] / norm;
        float s = a + b;
        mean[i] = s / c1;
    }
}

/* Load row `row` as a leaf of the f32-space metric: returns f32 dims (padded to 64 for BQ). */
static size_t load_leaf_f32(const ao_data *d, uint64_t row, float *v, float *hdr) {
    if (is_bq(d->metric)) {
        size_t nb = ao_bq_bytes(d->dims);
        ao_bq_dequantize((const uint8_t *)row_vec(d, row), nb, v);
        size_t fd = nb * 8;
        ao_new_header(f32_space_metric(d->metric), v, fd, hdr); /* new_leaf, mod.rs:34-37 */
        return fd;
    }
    memcpy(v, row_vec(d, row), 4 * (size_t)d->dims);
    const float *h = row_hdr(d, row);
    hdr[0] = h[0];
    if (d->metric == AH_DOT_PRODUCT) hdr[1] = h[1];
    return d->dims;
}

/* out_p/out_q: f32-space vectors (fd floats each), headers float[2]. Returns fd. */
AO_API size_t ao_two_means(const ao_data *d, const uint32_t *sample_rows, float *p, float *ph, float *q, float *qh) {
    int m = f32_space_metric(d->metric);
    int cosine = uses_cosine_two_means(d->metric);
    size_t fd = is_bq(d->metric) ? ao_bq_bytes(d->dims) * 8 : d->dims;
    float *k = (float *)malloc(sizeof(float) * fd);
    float kh[2] = {0, 0};
    ph[0] = ph[1] = qh[0] = qh[1] = 0.0f;
    load_leaf_f32(d, sample_rows[0], p, ph);
    load_leaf_f32(d, sample_rows[1], q, qh);
    if (cosine) {
        leaf_normalize(m, p, ph, fd);
        leaf_normalize(m, q, qh, fd);
    }
    leaf_init(m, p, ph, fd);
    leaf_init(m, q, qh, fd);
    float ic = 1.0f, jc = 1.0f;
    for (int it = 0; it < 10; it++) {
        load_leaf_f32(d, sample_rows[2 + it], k, kh);
        float di = ic * ao_non_built_distance(m, p, ph, k, kh, fd);
        float dj = jc * ao_non_built_distance(m, q, qh, k, kh, fd);
        float norm = cosine ? ao_norm(m, k, kh, fd) : 1.0f;
        if (isnan(norm) || norm <= 0.0f) continue;
        if (di < dj) {
            leaf_update_mean(p, k, norm, ic, fd);
            leaf_init(m, p, ph, fd);
            ic += 1.0f;
        } else if (dj < di) {
            leaf_update_mean(q, k, norm, jc, fd);
            leaf_init(m, q, qh, fd);
            jc += 1.0f;
        }
    }
    free(k);
    return fd;
}

/* create_split: euclidean.rs:55-77, manhattan.rs:58-80, cosine.rs:73-85, dot_product.rs:98-113,
 * binary_quantized_cosine.rs:77-93, binary_quantized_euclidean.rs:66-88, binary_quantized_manhattan.rs:70-92.
 * out_normal_vec is in the metric's codec. */
AO_API void ao_create_split(const ao_data *d, const uint32_t *sample_rows, void *out_normal_vec, float *out_normal_hdr) {
    size_t fd = is_bq(d->metric) ? ao_bq_bytes(d->dims) * 8 : d->dims;
    float *p = (float *)malloc(sizeof(float) * fd), *q = (float *)malloc(sizeof(float) * fd);
    float *nv = (float *)malloc(sizeof(float) * fd);
    float ph[2], qh[2];
    ao_two_means(d, sample_rows, p, ph, q, qh);
    for (size_t i = 0; i < fd; i++) nv[i] = p[i] - q[i];
    out_normal_hdr[0] = 0.0f;
    if (d->metric == AH_DOT_PRODUCT) out_normal_hdr[1] = 0.0f;

    if (!is_bq(d->metric)) {
        float nh[2] = {0.0f, 0.0f};
        if (d->metric == AH_DOT_PRODUCT) nh[0] = ph[0] - qh[0]; /* extra_dim = p.e - q.e, then normalize */
        leaf_normalize(d->metric, nv, nh, fd);
        if (d->metric == AH_EUCLIDEAN || d->metric == AH_MANHATTAN) {
            float bias = 0.0f; /* .map(|((n,p),q)| -n * (p + q) / 2.0).sum() */
            for (size_t i = 0; i < fd; i++) {
                float s = p[i] + q[i];
                float t = (-nv[i]) * s;
                bias = bias + t / 2.0f;
            }
            nh[0] = bias;
        }
        memcpy(out_normal_vec, nv, 4 * fd);
        out_normal_hdr[0] = nh[0];
        if (d->metric == AH_DOT_PRODUCT) out_normal_hdr[1] = nh[1];
    } else {
        /* UnalignedVector::<BinaryQuantized>::from_vec(p - q): quantise; Self::normalize divides the
         * +-1 values by a positive norm (or leaves them) and re-quantises: signs unchanged, except that
         * a NaN/<=0 norm (BQ-Manhattan, bq_manhattan.rs:59-66) skips the division — identical bits. */
        size_t nb = ao_bq_bytes(d->dims);
        uint8_t *nq = (uint8_t *)out_normal_vec;
        ao_bq_quantize(nv, fd, nq);
        float norm = ao_norm_no_header(d->metric, nq, d->dims);
        if (norm > 0.0f) {
            float *tmp = (float *)malloc(sizeof(float) * fd);
            ao_bq_dequantize(nq, nb, tmp);
            for (size_t i = 0; i < fd; i++) tmp[i] = tmp[i] / norm;
            ao_bq_quantize(tmp, fd, nq);
            free(tmp);
        }
        if (d->metric != AH_BQ_COSINE) {
            uint8_t *pq8 = (uint8_t *)malloc(nb), *qq8 = (uint8_t *)malloc(nb);
            float *nn = (float *)malloc(sizeof(float) * fd), *pp = (float *)malloc(sizeof(float) * fd),
                  *qq = (float *)malloc(sizeof(float) * fd);
            ao_bq_quantize(p, fd, pq8);
            ao_bq_quantize(q, fd, qq8);
            ao_bq_dequantize(nq, nb, nn);
            ao_bq_dequantize(pq8, nb, pp);
            ao_bq_dequantize(qq8, nb, qq);
            float bias = 0.0f;
            for (size_t i = 0; i < fd; i++) {
                float s = pp[i] + qq[i];
                float t = (-nn[i]) * s;
                bias = bias + t / 2.0f;
            }
            out_normal_hdr[0] = bias;
            free(pq8); free(qq8); free(nn); free(pp); free(qq);
        }
    }
    free(p); free(q); free(nv);
}

/* ---------------------------------------------------------------------------------------
 * make_tree_in_file, src/writer.rs:1167-1261, depth-first exactly like the reference, with the
 * randomness policy of include/arroy_hip_policy.h in place of `R: Rng`.
 * ------------------------------------------------------------------------------------- */
typedef struct ao_tree {
    ah_node *nodes;
    size_t n_nodes, cap_nodes;
    uint8_t *normals;
    size_t normals_len, cap_normals;
    uint32_t *desc;
    size_t desc_len, cap_desc;
    uint32_t root;
    uint64_t margin_evals, retries, dummy_normals;
} ao_tree;

static uint32_t push_node(ao_tree *t, ah_node nd) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 64;
        t->nodes = (ah_node *)realloc(t->nodes, t->cap_nodes * sizeof(ah_node));
    }
    t->nodes[t->n_nodes] = nd;
    return (uint32_t)t->n_nodes++;
}

/* split_imbalance, src/writer.rs:1348-1353 (f64). */
AO_API double ao_split_imbalance(uint64_t l, uint64_t r) {
    double ls = (double)l, rs = (double)r;
    double f = ls / (ls + rs + 2.220446049250313e-16);
    double g = 1.0 - f;
    return f > g ? f : g;
}

static uint32_t build_rec(const ao_data *d, ao_tree *t, uint32_t split_after, uint32_t *rows, uint64_t n,
                          uint64_t node_key, uint32_t depth, uint32_t *scratch) {
    ah_node nd;
    memset(&nd, 0, sizeof nd);
    nd.depth = depth;
    if (n <= split_after) { /* fit_in_descendant, src/writer.rs:474-477,1183-1188 */
        nd.kind = AH_NODE_DESCENDANTS;
        nd.offset = t->desc_len;
        nd.count = (uint32_t)n;
        if (t->desc_len + n > t->cap_desc) {
            t->cap_desc = (t->desc_len + n) * 2 + 64;
            t->desc = (uint32_t *)realloc(t->desc, t->cap_desc * sizeof(uint32_t));
        }
        for (uint64_t i = 0; i < n; i++) t->desc[t->desc_len + i] = d->ids ? d->ids[rows[i]] : rows[i];
        t->desc_len += n;
        return push_node(t, nd);
    }
    size_t hs = ao_header_floats(d->metric) * 4, vs = ao_vector_bytes(d->metric, d->dims);
    uint8_t *normal = (uint8_t *)malloc(hs + vs);
    uint8_t *sides = (uint8_t *)malloc(n);
    uint64_t n_left = 0;
    int remaining = 3;
    uint32_t attempt = 0;
    for (;;) { /* src/writer.rs:1193-1216 */
        uint32_t sample[AH_SPLIT_SAMPLES];
        uint64_t a, b;
        ah_choose_two(node_key, attempt, n, &a, &b);
        sample[0] = rows[a];
        sample[1] = rows[b];
        for (uint32_t it = 0; it < 10; it++) sample[2 + it] = rows[ah_choose(node_key, attempt, it, n)];
        float nh[2] = {0, 0};
        ao_create_split(d, sample, normal + hs, nh);
        memcpy(normal, nh, hs);
        ao_split_sides(d, normal + hs, nh, rows, n, sides, &n_left, NULL);
        t->margin_evals += n;
        if (ao_split_imbalance(n_left, n - n_left) < 0.95 || remaining == 0) break;
        remaining--;
        attempt++;
        t->retries++;
    }
    int has_normal = 1;
    if (ao_split_imbalance(n_left, n - n_left) > 0.99) { /* src/writer.rs:1220-1227 */
        has_normal = 0;
        t->dummy_normals++;
        n_left = 0;
        for (uint64_t i = 0; i < n; i++) {
            int left = (int)ah_random_side_is_left(node_key, i);
            sides[i] = (uint8_t)(1 - left);
            n_left += (uint64_t)left;
        }
    }
    /* stable partition: children are ascending-id lists (RoaringBitmap::from_sorted_iter, :1230-1231) */
    uint64_t li = 0, ri = n_left;
    for (uint64_t i = 0; i < n; i++) {
        if (sides[i]) scratch[ri++] = rows[i];
        else scratch[li++] = rows[i];
    }
    memcpy(rows, scratch, n * sizeof(uint32_t));
    free(sides);
    uint32_t left = build_rec(d, t, split_after, rows, n_left, ah_node_key_child(node_key, 0), depth + 1, scratch);
    uint32_t right = build_rec(d, t, split_after, rows + n_left, n - n_left, ah_node_key_child(node_key, 1), depth + 1,
                               scratch + n_left);
    nd.kind = AH_NODE_SPLIT;
    nd.has_normal = (uint8_t)has_normal;
    nd.left = left;
    nd.right = right;
    nd.count = (uint32_t)n;
    nd.offset = t->normals_len;
    if (has_normal) {
        if (t->normals_len + hs + vs > t->cap_normals) {
            t->cap_normals = (t->normals_len + hs + vs) * 2;
            t->normals = (uint8_t *)realloc(t->normals, t->cap_normals);
        }
        memcpy(t->normals + t->normals_len, normal, hs + vs);
        t->normals_len += hs + vs;
    }
    free(normal);
    return push_node(t, nd);
}

AO_API ao_tree *ao_build_tree(const ao_data *d, uint32_t split_after, uint64_t tree_seed) {
    ao_tree *t = (ao_tree *)calloc(1, sizeof(ao_tree));
    if (split_after == 0) split_after = d->dims;
    uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * (d->n ? d->n : 1));
    uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * (d->n ? d->n : 1));
    for (uint64_t i = 0; i < d->n; i++) rows[i] = (uint32_t)i;
    t->root = build_rec(d, t, split_after, rows, d->n, ah_node_key_root(tree_seed), 0, scratch);
    free(rows);
    free(scratch);
    return t;
}
/* make_tree_in_file over a caller-given ascending row subset (incremental_index_large_descendant, src/writer.rs:660-739) */
AO_API ao_tree *ao_build_tree_on(const ao_data *d, uint32_t split_after, uint64_t tree_seed, const uint32_t *subset, uint64_t n) {
    ao_tree *t = (ao_tree *)calloc(1, sizeof(ao_tree));
    if (split_after == 0) split_after = d->dims;
    uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(rows, subset, n * sizeof(uint32_t));
    t->root = build_rec(d, t, split_after, rows, n, ah_node_key_root(tree_seed), 0, scratch);
    free(rows);
    free(scratch);
    return t;
}
AO_API void ao_tree_view(const ao_tree *t, ah_forest_view *v, uint32_t *root_out) {
    memset(v, 0, sizeof *v);
    v->n_trees = 1;
    v->n_nodes = t->n_nodes;
    v->nodes = t->nodes;
    v->normals = t->normals;
    v->normals_len = t->normals_len; /* oracle records are [header][vector] */
    v->normal_header_offset = 0;
    v->normal_vector_offset = 0; /* set by ao_tree_view_for(metric) below when the metric is known */
    v->descendants = t->desc;
    v->descendants_len = t->desc_len;
    *root_out = t->root;
}
AO_API void ao_tree_counters(const ao_tree *t, uint64_t *margin_evals, uint64_t *retries, uint64_t *dummy) {
    *margin_evals = t->margin_evals;
    *retries = t->retries;
    *dummy = t->dummy_normals;
}
AO_API void ao_tree_free(ao_tree *t) {
    if (!t) return;
    free(t->nodes);
    free(t->normals);
    free(t->desc);
    free(t);
}

/* Build `n_trees` trees in parallel over trees (the analogue of rayon::scope over root tasks,
 * src/writer.rs:568-591); returns total margin evaluations. Used by the CPU-baseline timing. */
AO_API uint64_t ao_build_forest_count(const ao_data *d, uint32_t split_after, const uint64_t *seeds, uint32_t n_trees) {
    uint64_t total = 0;
    if ((int)n_trees >= omp_get_max_threads()) {
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
        for (int t = 0; t < (int)n_trees; t++) {
            ao_tree *tr = ao_build_tree(d, split_after, seeds[t]);
            total += tr->margin_evals;
            ao_tree_free(tr);
        }
    } else { /* fewer trees than cores: parallelise the margin loop inside each tree instead */
        for (uint32_t t = 0; t < n_trees; t++) {
            ao_tree *tr = ao_build_tree(d, split_after, seeds[t]);
            total += tr->margin_evals;
            ao_tree_free(tr);
        }
    }
    return total;
}

/* Synthetic data of the benchmark harness (include/arroy_hip_policy.h). */
AO_API void ao_synth_fill(uint64_t seed, int distribution, uint64_t first_item, uint64_t n, uint32_t dims, float *out) {
    /* the per-dataset part of the structured distributions (cluster centres / factor loadings), computed once */
    const uint64_t tlen = ah_synth_table_len(dims, distribution);
    int32_t *table = tlen ? (int32_t *)malloc(tlen * sizeof(int32_t)) : NULL;
    if (table) ah_synth_table_fill(seed, dims, distribution, table);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
        ah_synth_row(seed, first_item + (uint64_t)i, dims, distribution, table, out + (uint64_t)i * dims);
    free(table);
}

/* One component straight from the definition (tests: the row-wise fill above must agree with it). */
AO_API float ao_synth_value(uint64_t seed, uint64_t item, uint32_t dim, uint32_t dims, int distribution) {
    return ah_synth_value(seed, item, dim, dims, distribution);
}

AO_API int ao_num_threads(void) { return omp_get_max_threads(); }
AO_API void ao_set_num_threads(int n) { omp_set_num_threads(n); }

/* =======================================================================================
 * "Reference-order" build: the SAME make_tree_in_file, but with a restatement of the reference's
 * own randomness — `StdRng` of rand 0.8.5 (= ChaCha12 of rand_chacha 0.3.1) consumed in the
 * reference's depth-first order — so that the insta snapshots of src/tests/writer.rs can be
 * replayed.  rand / rand_chacha are UN-VENDORED dependencies (Cargo.toml:16-32, no Cargo.lock):
 * the algorithms below are restated from their published sources:
 *   - ChaCha (D. J. Bernstein), 12 rounds, 64-bit block counter in words 12-13, stream id 0;
 *     rand_chacha refills 4 consecutive blocks (64 words) at a time, words consumed in order;
 *   - `Standard` u8  = low byte of next_u32;  [u8; 32] = 32 such draws (rand/src/distributions/other.rs);
 *   - `Standard` bool = top bit of next_u32;
 *   - `gen_range(lo..=hi)` for u32 = UniformInt::sample_single_inclusive: widening multiply with the
 *     rejection zone `(range << range.leading_zeros()) - 1` (rand/src/distributions/uniform.rs);
 *   - `seq::index::sample(rng, len, 2)` = Floyd's algorithm, fully shuffled variant for amount < 50
 *     (rand/src/seq/index.rs).
 * Pinned by tests/test_oracle_reference_order.py against snapshots in src/tests/writer.rs.
 * Scope: one tree per build (n_trees(1)): with several trees the reference iterates a hashbrown
 * `IntMap` and a rayon scope whose orders are not restated here.
 * ===================================================================================== */
typedef struct ao_chacha {
    uint32_t key[8];
    uint64_t counter;
    uint32_t buf[64];
    int index;
} ao_chacha;

static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define AO_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

static void chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4],      key[5],      key[6],      key[7],      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 6; r++) { /* 12 rounds = 6 double rounds */
        AO_QR(x[0], x[4], x[8], x[12]) AO_QR(x[1], x[5], x[9], x[13]) AO_QR(x[2], x[6], x[10], x[14]) AO_QR(x[3], x[7], x[11], x[15])
        AO_QR(x[0], x[5], x[10], x[15]) AO_QR(x[1], x[6], x[11], x[12]) AO_QR(x[2], x[7], x[8], x[13]) AO_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
AO_API void ao_rng_from_seed(ao_chacha *r, const uint8_t seed[32]) {
    for (int i = 0; i < 8; i++) memcpy(&r->key[i], seed + 4 * i, 4); /* little-endian words */
    r->counter = 0;
    r->index = 64;
}
AO_API uint32_t ao_rng_next_u32(ao_chacha *r) {
    if (r->index >= 64) {
        for (int b = 0; b < 4; b++) chacha12_block(r->key, r->counter + (uint64_t)b, r->buf + 16 * b);
        r->counter += 4;
        r->index = 0;
    }
    return r->buf[r->index++];
}
AO_API void ao_rng_gen_seed(ao_chacha *r, uint8_t out[32]) {
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)ao_rng_next_u32(r);
}
AO_API int ao_rng_gen_bool(ao_chacha *r) { return (int32_t)ao_rng_next_u32(r) < 0; }
AO_API uint32_t ao_rng_gen_range_inclusive_u32(ao_chacha *r, uint32_t low, uint32_t high) {
    uint32_t range = high - low + 1u;
    if (range == 0) return ao_rng_next_u32(r);
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint32_t v = ao_rng_next_u32(r);
        uint64_t m = (uint64_t)v * (uint64_t)range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return low + hi;
    }
}
/* seq::index::sample(rng, length, 2) -> (index(0), index(1)) */
AO_API void ao_rng_index_sample2(ao_chacha *r, uint32_t length, uint32_t out[2]) {
    uint32_t idx[2];
    int n = 0;
    for (uint32_t j = length - 2; j < length; j++) {
        uint32_t t = ao_rng_gen_range_inclusive_u32(r, 0, j);
        int pos = -1;
        for (int i = 0; i < n; i++)
            if (idx[i] == t) { pos = i; break; }
        if (pos >= 0) { /* indices.insert(pos, j) */
            for (int i = n; i > pos; i--) idx[i] = idx[i - 1];
            idx[pos] = j;
            n++;
            continue;
        }
        idx[n++] = t;
    }
    out[0] = idx[0];
    out[1] = idx[1];
}

typedef struct ao_ref_node {
    uint32_t id;          /* the reference's tree-node id (ConcurrentNodeIds) */
    uint8_t kind, has_normal;
    uint32_t left, right; /* node ids */
    uint64_t offset;      /* normals blob / descendants blob */
    uint32_t count;
} ao_ref_node;
typedef struct ao_ref_tree {
    ao_ref_node *nodes;
    size_t n_nodes, cap_nodes;
    uint8_t *normals;
    size_t normals_len, cap_normals;
    uint32_t *desc;
    size_t desc_len, cap_desc;
    uint32_t next_id;
    /* ConcurrentNodeIds (src/parallel.rs:206-254): ids freed by earlier builds are handed out first, in ascending
     * order, then `current` counts up */
    uint32_t *avail;
    size_t n_avail, avail_pos;
} ao_ref_tree;

static uint32_t ref_next_id(ao_ref_tree *t) {
    if (t->avail_pos < t->n_avail) return t->avail[t->avail_pos++];
    return t->next_id++;
}

static void ref_push(ao_ref_tree *t, ao_ref_node nd) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 64;
        t->nodes = (ao_ref_node *)realloc(t->nodes, t->cap_nodes * sizeof(ao_ref_node));
    }
    t->nodes[t->n_nodes++] = nd;
}

/* make_tree_in_file, src/writer.rs:1167-1261, with `next_id` exactly as the reference threads it. */
static uint32_t ref_build_rec(const ao_data *d, ao_ref_tree *t, uint32_t split_after, uint32_t *rows, uint64_t n,
                              ao_chacha *rng, int has_next_id, uint32_t next_id, uint32_t *scratch) {
    ao_ref_node nd;
    memset(&nd, 0, sizeof nd);
    if (n <= split_after) {
        nd.id = has_next_id ? next_id : ref_next_id(t);
        nd.kind = AH_NODE_DESCENDANTS;
        nd.offset = t->desc_len;
        nd.count = (uint32_t)n;
        if (t->desc_len + n > t->cap_desc) {
            t->cap_desc = (t->desc_len + n) * 2 + 64;
            t->desc = (uint32_t *)realloc(t->desc, t->cap_desc * sizeof(uint32_t));
        }
        for (uint64_t i = 0; i < n; i++) t->desc[t->desc_len + i] = d->ids ? d->ids[rows[i]] : rows[i];
        t->desc_len += n;
        ref_push(t, nd);
        return nd.id;
    }
    size_t hs = ao_header_floats(d->metric) * 4, vs = ao_vector_bytes(d->metric, d->dims);
    uint8_t *normal = (uint8_t *)malloc(hs + vs);
    uint8_t *sides = (uint8_t *)malloc(n);
    uint64_t n_left = 0;
    int remaining = 3;
    for (;;) {
        uint32_t sample[AH_SPLIT_SAMPLES], two[2];
        ao_rng_index_sample2(rng, (uint32_t)n, two); /* choose_two, src/parallel.rs:342-355 */
        sample[0] = rows[two[0]];
        sample[1] = rows[two[1]];
        for (int it = 0; it < 10; it++) /* choose, :358-367: gen_range(0..=len-1), one per two-means iteration */
            sample[2 + it] = rows[ao_rng_gen_range_inclusive_u32(rng, 0, (uint32_t)n - 1u)];
        float nh[2] = {0, 0};
        ao_create_split(d, sample, normal + hs, nh);
        memcpy(normal, nh, hs);
        ao_split_sides(d, normal + hs, nh, rows, n, sides, &n_left, NULL);
        if (ao_split_imbalance(n_left, n - n_left) < 0.95 || remaining == 0) break;
        remaining--;
    }
    int has_normal = 1;
    if (ao_split_imbalance(n_left, n - n_left) > 0.99) {
        has_normal = 0;
        n_left = 0;
        for (uint64_t i = 0; i < n; i++) { /* Side::random: true -> Left (src/lib.rs:135-141) */
            int left = ao_rng_gen_bool(rng);
            sides[i] = (uint8_t)(1 - left);
            n_left += (uint64_t)left;
        }
    }
    uint64_t li = 0, ri = n_left;
    for (uint64_t i = 0; i < n; i++) {
        if (sides[i]) scratch[ri++] = rows[i];
        else scratch[li++] = rows[i];
    }
    memcpy(rows, scratch, n * sizeof(uint32_t));
    free(sides);
    uint32_t left = ref_build_rec(d, t, split_after, rows, n_left, rng, 0, 0, scratch);
    uint32_t right = ref_build_rec(d, t, split_after, rows + n_left, n - n_left, rng, 0, 0, scratch + n_left);
    nd.id = has_next_id ? next_id : ref_next_id(t); /* allocated AFTER the children (src/writer.rs:1257) */
    nd.kind = AH_NODE_SPLIT;
    nd.has_normal = (uint8_t)has_normal;
    nd.left = left;
    nd.right = right;
    nd.count = (uint32_t)n;
    nd.offset = t->normals_len;
    if (t->normals_len + hs + vs > t->cap_normals) {
        t->cap_normals = (t->normals_len + hs + vs) * 2;
        t->normals = (uint8_t *)realloc(t->normals, t->cap_normals);
    }
    memcpy(t->normals + t->normals_len, normal, hs + vs);
    t->normals_len += hs + vs;
    free(normal);
    ref_push(t, nd);
    return nd.id;
}

/* Writer::build for a FRESH index on ONE rayon thread (src/tests/mod.rs:94), src/writer.rs:487-629:
 *   - the caller's rng may already have been used (`skip_u32` draws, e.g. to generate the test vectors);
 *   - roots take node ids 0..n_trees-1 (:556-561), remaining ids are handed out in allocation order;
 *   - the task that walks `descendants` gets `StdRng::from_seed(rng.gen())` (:575); it visits the roots in
 *     ascending id order (IntMap = hashbrown with the identity hash: bucket = key) and gives every root its
 *     own `StdRng::from_seed(rng.gen())` (:795) in THAT order;
 *   - the spawned tasks then run last-in-first-out on the single worker (rayon's local deque), i.e. the
 *     highest root first: that is what fixes which tree gets which node ids.
 * Each task = make_tree_in_file on all items (fit_in_memory with unlimited memory consumes no randomness).
 * The order facts in the last two bullets are not in arroy's sources (hashbrown / rayon internals); they are
 * confirmed by the 10-tree snapshot arroy__tests__writer__write_and_update_lot_of_random_points.snap. */
AO_API ao_ref_tree *ao_build_forest_reference_order(const ao_data *d, uint32_t split_after, uint32_t n_trees,
                                                    const uint8_t seed[32], uint64_t skip_u32) {
    ao_ref_tree *t = (ao_ref_tree *)calloc(1, sizeof(ao_ref_tree));
    if (split_after == 0) split_after = d->dims;
    ao_chacha rng0, rng1;
    uint8_t s[32];
    ao_rng_from_seed(&rng0, seed);
    for (uint64_t i = 0; i < skip_u32; i++) (void)ao_rng_next_u32(&rng0);
    t->next_id = n_trees;
    ao_rng_gen_seed(&rng0, s);
    ao_rng_from_seed(&rng1, s);
    uint8_t *task_seeds = (uint8_t *)malloc(32 * (size_t)(n_trees ? n_trees : 1));
    for (uint32_t r = 0; r < n_trees; r++) ao_rng_gen_seed(&rng1, task_seeds + 32 * (size_t)r);
    uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * (d->n ? d->n : 1));
    uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * (d->n ? d->n : 1));
    for (uint32_t k = 0; k < n_trees; k++) {
        uint32_t r = n_trees - 1 - k; /* LIFO */
        ao_chacha rng2;
        ao_rng_from_seed(&rng2, task_seeds + 32 * (size_t)r);
        for (uint64_t i = 0; i < d->n; i++) rows[i] = (uint32_t)i;
        ref_build_rec(d, t, split_after, rows, d->n, &rng2, 1, r, scratch);
    }
    free(task_seeds);
    free(rows);
    free(scratch);
    return t;
}
AO_API ao_ref_tree *ao_build_tree_reference_order(const ao_data *d, uint32_t split_after, const uint8_t seed[32]) {
    return ao_build_forest_reference_order(d, split_after, 1, seed, 0);
}
AO_API size_t ao_ref_tree_nodes(const ao_ref_tree *t, const ao_ref_node **nodes, const uint8_t **normals,
                                const uint32_t **desc) {
    *nodes = t->nodes;
    *normals = t->normals;
    *desc = t->desc;
    return t->n_nodes;
}
/* The general form of the build step of `Writer::build` (src/writer.rs:556-591) in the reference's order, used by the
 * incremental snapshot replays: `descendants` is the map the reference walks in
 * insert_descendants_in_file_and_spawn_tasks (:744-844) — the Descendants nodes modified by the routing of the new
 * items plus one all-items entry per missing tree — given here in the map's iteration order.
 *   - the walking task gets `StdRng::from_seed(main_rng.gen())` (:575);
 *   - an entry that fits in a descendant is written as is (no randomness);
 *   - every other entry gets its own `StdRng::from_seed(rng.gen())` (:795), in iteration order, and becomes a task
 *     `make_tree_in_file(.., Some(descendant_id))`: the sub-tree's root keeps the entry's node id;
 *   - tasks run last-in-first-out on the single worker; new node ids come from `avail` first, then `current`. */
AO_API ao_ref_tree *ao_ref_build_descendants(const ao_data *d, uint32_t split_after, const uint32_t *desc_ids,
                                             const uint64_t *offsets, const uint32_t *rows, uint32_t n_desc,
                                             ao_chacha *main_rng, const uint32_t *avail, uint32_t n_avail,
                                             uint32_t current) {
    ao_ref_tree *t = (ao_ref_tree *)calloc(1, sizeof(ao_ref_tree));
    if (split_after == 0) split_after = d->dims;
    t->next_id = current;
    t->avail = (uint32_t *)malloc(sizeof(uint32_t) * (n_avail ? n_avail : 1));
    memcpy(t->avail, avail, sizeof(uint32_t) * n_avail);
    t->n_avail = n_avail;
    ao_chacha rng1;
    uint8_t s[32];
    ao_rng_gen_seed(main_rng, s);
    ao_rng_from_seed(&rng1, s);
    uint8_t *task_seeds = (uint8_t *)malloc(32 * (size_t)(n_desc ? n_desc : 1));
    uint64_t max_n = 1;
    for (uint32_t k = 0; k < n_desc; k++) {
        const uint64_t n = offsets[k + 1] - offsets[k];
        if (n > max_n) max_n = n;
        if (n > split_after) ao_rng_gen_seed(&rng1, task_seeds + 32 * (size_t)k);
    }
    uint32_t *work = (uint32_t *)malloc(sizeof(uint32_t) * max_n);
    uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * max_n);
    for (uint32_t k = 0; k < n_desc; k++) { /* entries that fit: written by the walking task itself */
        const uint64_t n = offsets[k + 1] - offsets[k];
        if (n > split_after) continue;
        memcpy(work, rows + offsets[k], n * sizeof(uint32_t));
        ref_build_rec(d, t, split_after, work, n, NULL, 1, desc_ids[k], scratch);
    }
    for (uint32_t i = 0; i < n_desc; i++) { /* LIFO */
        const uint32_t k = n_desc - 1 - i;
        const uint64_t n = offsets[k + 1] - offsets[k];
        if (n <= split_after) continue;
        ao_chacha rng2;
        ao_rng_from_seed(&rng2, task_seeds + 32 * (size_t)k);
        memcpy(work, rows + offsets[k], n * sizeof(uint32_t));
        ref_build_rec(d, t, split_after, work, n, &rng2, 1, desc_ids[k], scratch);
    }
    free(task_seeds);
    free(work);
    free(scratch);
    return t;
}
/* One `make_tree_in_file(.., Some(root_id))` (src/writer.rs:1167-1261) over `rows` with the caller's task rng and the
 * caller's id allocator (`avail` / `*avail_pos` / `*current` are read and advanced): the building block of the
 * low-memory replay, where a task first builds a small tree and keeps feeding items into its leaves. */
AO_API ao_ref_tree *ao_ref_subtree(const ao_data *d, uint32_t split_after, const uint32_t *rows, uint64_t n, ao_chacha *rng,
                                   uint32_t root_id, const uint32_t *avail, uint32_t n_avail, uint32_t *avail_pos,
                                   uint32_t *current) {
    ao_ref_tree *t = (ao_ref_tree *)calloc(1, sizeof(ao_ref_tree));
    if (split_after == 0) split_after = d->dims;
    t->next_id = *current;
    t->avail = (uint32_t *)malloc(sizeof(uint32_t) * (n_avail ? n_avail : 1));
    memcpy(t->avail, avail, sizeof(uint32_t) * n_avail);
    t->n_avail = n_avail;
    t->avail_pos = *avail_pos;
    uint32_t *work = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(work, rows, n * sizeof(uint32_t));
    ref_build_rec(d, t, split_after, work, n, rng, 1, root_id, scratch);
    free(work);
    free(scratch);
    *avail_pos = (uint32_t)t->avail_pos;
    *current = t->next_id;
    return t;
}
AO_API uint32_t ao_ref_tree_next_id(const ao_ref_tree *t, uint32_t *avail_used) {
    if (avail_used) *avail_used = (uint32_t)t->avail_pos;
    return t->next_id;
}
AO_API void ao_ref_tree_free(ao_ref_tree *t) {
    if (!t) return;
    free(t->avail);
    free(t->nodes);
    free(t->normals);
    free(t->desc);
    free(t);
}

/* =======================================================================================
 * Reader::nns_by_leaf, src/reader.rs:317-401, over a forest given as an ah_forest_view: the best-first
 * descent (BinaryHeap of (OrderedFloat(distance), NodeId), all roots at +inf), candidate collection,
 * sort + dedup, re-rank, top-k, normalized_distance.  Node identity = index in `view->nodes` (the
 * reference's NodeId::tree(id); ties in the queue are broken by it exactly as the derived Ord does).
 * ===================================================================================== */
typedef struct heap_ent {
    float d;
    uint32_t node;
} heap_ent;
static int heap_less(heap_ent a, heap_ent b) { /* a < b in (OrderedFloat, NodeId) order */
    int c = of_cmp(a.d, b.d);
    if (c) return c < 0;
    return a.node < b.node;
}
static void heap_push(heap_ent **h, size_t *n, size_t *cap, heap_ent e) {
    if (*n == *cap) {
        *cap = *cap ? 2 * *cap : 64;
        *h = (heap_ent *)realloc(*h, *cap * sizeof(heap_ent));
    }
    size_t i = (*n)++;
    (*h)[i] = e;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!heap_less((*h)[p], (*h)[i])) break;
        heap_ent t = (*h)[p]; (*h)[p] = (*h)[i]; (*h)[i] = t;
        i = p;
    }
}
static heap_ent heap_pop(heap_ent *h, size_t *n) {
    heap_ent top = h[0];
    h[0] = h[--(*n)];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && heap_less(h[m], h[l])) m = l;
        if (r < *n && heap_less(h[m], h[r])) m = r;
        if (m == i) break;
        heap_ent t = h[m]; h[m] = h[i]; h[i] = t;
        i = m;
    }
    return top;
}
static int u32_cmp(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}
static int sorted_contains(const uint32_t *a, size_t n, uint32_t x) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == x;
}

/* Returns the number of results; if out_cand != NULL also returns the sorted, de-duplicated candidate ids
 * (at most cand_cap of them) and their count in *out_n_cand. */
AO_API size_t ao_search(const ao_data *d, const ah_forest_view *f, const void *qv, const float *qh, size_t count,
                        size_t search_k, size_t oversampling, const uint32_t *filter_sorted, size_t n_filter,
                        int have_filter, uint32_t *out_ids, float *out_dists, uint32_t *out_cand, size_t cand_cap,
                        size_t *out_n_cand) {
    if (out_n_cand) *out_n_cand = 0;
    if (d->n == 0) return 0; /* reader.rs:323-325 */
    size_t sk = search_k ? search_k : count * (size_t)f->n_trees; /* :330 */
    size_t os = oversampling ? oversampling : (is_bq(d->metric) ? 3 : 1); /* DEFAULT_OVERSAMPLING, :331-335 */
    sk = sk * os;
    heap_ent *heap = NULL;
    size_t hn = 0, hcap = 0;
    for (uint32_t t = 0; t < f->n_trees; t++) heap_push(&heap, &hn, &hcap, (heap_ent){INFINITY, f->roots[t]});
    uint32_t *nns = NULL;
    size_t nn = 0, ncap = 0;
    size_t hs = ao_header_floats(d->metric) * 4;
    while (nn < sk && hn > 0) { /* :341-374 */
        heap_ent top = heap_pop(heap, &hn);
        const ah_node *nd = &f->nodes[top.node];
        if (nd->kind == AH_NODE_DESCENDANTS) {
            const uint32_t *ids = f->descendants + nd->offset;
            if (nn + nd->count > ncap) {
                ncap = (nn + nd->count) * 2 + 64;
                nns = (uint32_t *)realloc(nns, ncap * sizeof(uint32_t));
            }
            for (uint32_t i = 0; i < nd->count; i++)
                if (!have_filter || sorted_contains(filter_sorted, n_filter, ids[i])) nns[nn++] = ids[i];
        } else {
            float margin = 0.0f; /* `None => 0.0`, :366-369 */
            if (nd->has_normal) {
                const uint8_t *rec = f->normals + nd->offset;
                float nh[2] = {0, 0};
                memcpy(nh, rec + f->normal_header_offset, hs);
                margin = ao_margin(d->metric, rec + f->normal_vector_offset, nh, qv, qh, d->dims);
            }
            heap_push(&heap, &hn, &hcap, (heap_ent){ao_pq_distance(top.d, margin, 0), nd->left});
            heap_push(&heap, &hn, &hcap, (heap_ent){ao_pq_distance(top.d, margin, 1), nd->right});
        }
    }
    free(heap);
    qsort(nns, nn, sizeof(uint32_t), u32_cmp); /* sort_unstable + dedup, :378-379 */
    size_t u = 0;
    for (size_t i = 0; i < nn; i++)
        if (u == 0 || nns[i] != nns[u - 1]) nns[u++] = nns[i];
    if (out_cand) {
        for (size_t i = 0; i < u && i < cand_cap; i++) out_cand[i] = nns[i];
        if (out_n_cand) *out_n_cand = u;
    }
    /* ids -> rows */
    uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * (u ? u : 1));
    for (size_t i = 0; i < u; i++) {
        if (!d->ids) rows[i] = nns[i];
        else {
            size_t lo = 0, hi = d->n;
            while (lo < hi) { size_t mid = (lo + hi) / 2; if (d->ids[mid] < nns[i]) lo = mid + 1; else hi = mid; }
            rows[i] = (uint32_t)lo;
        }
    }
    size_t m = u ? ao_rerank(d, qv, qh, rows, u, count, out_ids, out_dists) : 0;
    free(rows);
    free(nns);
    return m;
}

/* insert_items_in_descendants_from_frozen_reader, src/writer.rs:1398-1459: every item walks independently from
 * each root to a Descendants node; `normal: None` nodes use the policy coin.  out_leaf[t * n + i]. */
AO_API void ao_route_items(const ao_data *d, const ah_forest_view *f, const uint32_t *rows, size_t n,
                           const uint64_t *tree_seeds, uint32_t *out_leaf) {
    size_t hs = ao_header_floats(d->metric) * 4;
    for (uint32_t t = 0; t < f->n_trees; t++)
        for (size_t i = 0; i < n; i++) {
            uint32_t node = f->roots[t];
            uint32_t id = d->ids ? d->ids[rows[i]] : rows[i];
            for (;;) {
                const ah_node *nd = &f->nodes[node];
                if (nd->kind == AH_NODE_DESCENDANTS) break;
                int right;
                if (nd->has_normal) {
                    const uint8_t *rec = f->normals + nd->offset;
                    float nh[2] = {0, 0};
                    memcpy(nh, rec + f->normal_header_offset, hs);
                    float m = ao_margin(d->metric, rec + f->normal_vector_offset, nh, row_vec(d, rows[i]), row_hdr(d, rows[i]), d->dims);
                    right = ao_side_of_margin(m);
                } else {
                    right = (int)(ah_route_side_is_left(tree_seeds[t], node, id) ^ 1u);
                }
                node = right ? nd->right : nd->left;
            }
            out_leaf[(size_t)t * n + i] = node;
        }
}
