// split_device.h — device code of the build side: margins and `create_split` (two_means + plane).
//
// Reference: src/distance/mod.rs:126-223 (two_means, two_means_binary_quantized), the per-metric
// create_split / margin / normalize / init / norm (src/distance/<metric>.rs) and update_mean
// (mod.rs:86-94).  One WAVE (64 lanes) runs one create_split; the d-length vector operations are spread
// over the lanes, the exact-order reductions are done by octets (all 8 octets redundantly, so every lane
// holds every scalar and no broadcast is needed).  Centroids live in LDS.
#pragma once

#include "device_math.h"

namespace ah {

// f32-space metric in which a metric runs its two-means (mod.rs:173-223: BQ variants de-quantise).
__host__ __device__ __forceinline__ int f32_space_metric(int metric) {
    switch (metric) {
    case AH_BQ_EUCLIDEAN: return AH_EUCLIDEAN;
    case AH_BQ_MANHATTAN: return AH_MANHATTAN;
    case AH_BQ_COSINE: return AH_COSINE;
    default: return metric;
    }
}
// the `cosine` flag each create_split passes to two_means
__host__ __device__ __forceinline__ bool two_means_is_cosine(int metric) {
    return metric == AH_COSINE || metric == AH_DOT_PRODUCT || metric == AH_BQ_COSINE;
}
// number of f32 elements of a centroid: dims, or 64*words for the 1-bit codec (padding included,
// binary_quantized.rs:67-69)
__host__ __device__ __forceinline__ uint32_t f32_space_dims(int metric, uint32_t dims) {
    return metric_is_bq_dev(metric) ? ((dims + 63u) / 64u) * 64u : dims;
}
__host__ __device__ __forceinline__ uint32_t f32_space_pitch(int metric, uint32_t dims) {
    return (f32_space_dims(metric, dims) + 31u) & ~31u;
}

struct LeafHdr {
    float h0, h1;  // {bias} | {norm} | {extra_dim, norm}
};

// ---- margins ---------------------------------------------------------------------------------
// D::margin(normal, item) for an f32 metric; `s_n` = normal in LDS (pitch floats), octet-cooperative.
template <int METRIC>
__device__ __forceinline__ float margin_f32(const DataView &dv, const float *s_n, LeafHdr nh, uint64_t row, uint32_t j) {
    const float *rp = dv.rows_f32 + row * dv.pitch;
    float d;
    if (dv.dims >= 32) {
        d = octet_reduce_stream<OP_DOT>(reinterpret_cast<const float4 *>(s_n), rp, dv.dims, j);
    } else {
        d = thread_reduce_small<OP_DOT>(s_n, rp, dv.dims);
    }
    if (METRIC == AH_EUCLIDEAN || METRIC == AH_MANHATTAN) return f_add(nh.h0, d);  // euclidean.rs:79-81
    if (METRIC == AH_DOT_PRODUCT) return f_add(d, f_mul(nh.h0, dv.headers[2 * row]));  // dot_product.rs:115-117
    return d;                                                                           // cosine.rs:87-89
}
// 1-bit metrics: bias + bqdot (bq_euclidean.rs:90-92, bq_manhattan.rs:94-96) or bqdot (bq_cosine.rs:95-97).
__device__ __forceinline__ float margin_bq(const DataView &dv, const uint64_t *s_n, LeafHdr nh, uint64_t row) {
    const uint64_t *rp = dv.rows_bq + row * dv.pitch;
    uint32_t ham = 0;
    for (uint32_t w = 0; w < dv.pitch; w++) ham += (uint32_t)__popcll(rp[w] ^ s_n[w]);
    float d = (float)bq_dot_from_hamming(ham, dv.words);
    return dv.metric == AH_BQ_COSINE ? d : f_add(nh.h0, d);
}
// D::side: Right (1) iff the sign bit of the margin is clear (mod.rs:103-110).
__device__ __forceinline__ uint32_t side_of_margin(float m) { return (__float_as_uint(m) >> 31) ^ 1u; }

// ---- two_means / create_split, one wave --------------------------------------------------------

// Load dataset row `row` as a leaf of the f32-space metric into LDS (`dst`, fpitch floats).
// BQ: new_leaf(vector.to_vec()) (mod.rs:34-37,189-190,203): +-1.0 per bit, header by the non-BQ metric.
__device__ __forceinline__ LeafHdr wave_load_leaf(const DataView &dv, uint64_t row, float *dst, uint32_t fd,
                                                  uint32_t fpitch, uint32_t lane) {
    LeafHdr h = {0.0f, 0.0f};
    if (!metric_is_bq_dev(dv.metric)) {
        const float *rp = dv.rows_f32 + row * dv.pitch;
        for (uint32_t i = lane; i < fpitch; i += 64) dst[i] = i < dv.dims ? rp[i] : 0.0f;
        h.h0 = dv.headers[row * (dv.metric == AH_DOT_PRODUCT ? 2u : 1u)];
        if (dv.metric == AH_DOT_PRODUCT) h.h1 = dv.headers[2 * row + 1];
        __syncthreads();
    } else {
        const uint64_t *rp = dv.rows_bq + row * dv.pitch;
        for (uint32_t i = lane; i < fpitch; i += 64) {
            float v = 0.0f;
            if (i < fd) v = f_sub(f_mul((float)((rp[i >> 6] >> (i & 63)) & 1ull), 2.0f), 1.0f);
            dst[i] = v;
        }
        __syncthreads();
        if (dv.metric == AH_BQ_COSINE) h.h0 = f_sqrt(octet_reduce_any<OP_DOT>(dst, dst, fd, lane & 7u));
    }
    return h;
}

// D::norm(leaf) in the f32-space metric M (mod.rs:70-72; dot_product.rs:72-75).
template <int M>
__device__ __forceinline__ float wave_norm(const float *v, LeafHdr h, uint32_t fd, uint32_t lane) {
    float d = octet_reduce_any<OP_DOT>(v, v, fd, lane & 7u);
    if (M == AH_DOT_PRODUCT) d = f_add(d, f_mul(h.h0, h.h0));
    return f_sqrt(d);
}
// D::init (cosine.rs:69-71, dot_product.rs:94-96)
template <int M>
__device__ __forceinline__ void wave_init(const float *v, LeafHdr &h, uint32_t fd, uint32_t lane) {
    if (M == AH_COSINE) h.h0 = f_sqrt(octet_reduce_any<OP_DOT>(v, v, fd, lane & 7u));
    if (M == AH_DOT_PRODUCT) h.h1 = octet_reduce_any<OP_DOT>(v, v, fd, lane & 7u);
}
// D::normalize (mod.rs:76-82; dot_product.rs:85-92)
template <int M>
__device__ __forceinline__ void wave_normalize(float *v, LeafHdr &h, uint32_t fd, uint32_t lane) {
    float norm = wave_norm<M>(v, h, fd, lane);
    __syncthreads();
    if (norm > 0.0f) {
        for (uint32_t i = lane; i < fd; i += 64) v[i] = f_div(v[i], norm);
        if (M == AH_DOT_PRODUCT) h.h0 = f_div(h.h0, norm);
    }
    __syncthreads();
}
// D::non_built_distance(p, k) in the f32-space metric (mod.rs:54-56; dot_product.rs:58-70)
template <int M>
__device__ __forceinline__ float wave_non_built_distance(const float *p, LeafHdr ph, const float *k, LeafHdr kh,
                                                         uint32_t fd, uint32_t lane) {
    const uint32_t j = lane & 7u;
    if (M == AH_EUCLIDEAN) return octet_reduce_any<OP_EUCLID>(p, k, fd, j);
    if (M == AH_MANHATTAN) {
        if (fd >= 32) return octet_manhattan(p, k, fd, j);
        float r = 0.0f;
        for (uint32_t i = 0; i < fd; i++) r = f_add(r, fabsf(f_sub(p[i], k[i])));
        return r;
    }
    float pq = octet_reduce_any<OP_DOT>(p, k, fd, j);
    if (M == AH_COSINE) return cosine_from_dot(pq, ph.h0, kh.h0);
    // DotProduct
    pq = f_add(pq, f_mul(ph.h0, kh.h0));
    float ppqq = f_mul(ph.h1, kh.h1);
    if (ppqq >= 1.17549435e-38f) return f_sub(2.0f, f_div(f_mul(2.0f, pq), f_sqrt(ppqq)));
    return 2.0f;
}

// Up to four reductions in ONE pass over LDS (fd >= 32): octet i of the wave evaluates (a[i], b[i]) for i < n, the other
// octets repeat the last one; out[i] = its result on every lane.  Every octet runs exactly the arithmetic of
// octet_reduce / octet_manhattan — same chains, same tree — so the values are those of n separate calls; what changes is
// the LDS traffic: a ds_read_b128 returns 1 KiB to the wave whether its octets ask for the same 128 bytes or not, and
// two_means used to spend 3-4 such passes per iteration with all eight octets computing the same number (create_split was
// LDS-bound: 53 ms of the 10M x 100-tree build).
template <int OP, bool MANHATTAN>
__device__ __forceinline__ void wave_reduce_multi(const float *a0, const float *b0, const float *a1, const float *b1,
                                                  const float *a2, const float *b2, const float *a3, const float *b3, uint32_t n,
                                                  uint32_t fd, uint32_t lane, float (&out)[4]) {
    const uint32_t oct = min(lane >> 3, n - 1u);
    const float *pa = a0, *pb = b0;
    if (oct == 1u) {
        pa = a1;
        pb = b1;
    } else if (oct == 2u) {
        pa = a2;
        pb = b2;
    } else if (oct == 3u) {
        pa = a3;
        pb = b3;
    }
    const float r = MANHATTAN ? octet_manhattan(pa, pb, fd, lane & 7u) : octet_reduce<OP>(pa, pb, fd, lane & 7u);
    out[0] = __shfl(r, 0);
    out[1] = __shfl(r, 8);
    out[2] = __shfl(r, 16);
    out[3] = __shfl(r, 24);
}

// non_built_distance from the reduction it needs (`red` = dot / squared distance / Manhattan sum of (p, k))
template <int M>
__device__ __forceinline__ float non_built_from_reduction(float red, LeafHdr ph, LeafHdr kh) {
    if (M == AH_EUCLIDEAN || M == AH_MANHATTAN) return red;
    if (M == AH_COSINE) return cosine_from_dot(red, ph.h0, kh.h0);
    const float pq = f_add(red, f_mul(ph.h0, kh.h0));  // DotProduct (dot_product.rs:58-70)
    const float ppqq = f_mul(ph.h1, kh.h1);
    if (ppqq >= 1.17549435e-38f) return f_sub(2.0f, f_div(f_mul(2.0f, pq), f_sqrt(ppqq)));
    return 2.0f;
}
// D::init from <v, v>
template <int M>
__device__ __forceinline__ void init_from_reduction(float vv, LeafHdr &h) {
    if (M == AH_COSINE) h.h0 = f_sqrt(vv);
    if (M == AH_DOT_PRODUCT) h.h1 = vv;
}

// two_means (mod.rs:126-171 / 173-223).  s_p, s_q, s_k: LDS, fpitch floats each.  rows[12] = the sampled
// dataset rows (choose_two, then the ten `choose`).  Returns the two centroid headers.
// fd >= 32: the reductions of an iteration — <p,k>, <q,k>, <k,k> and the D::init of the centroid the previous iteration
// moved — go through ONE fused pass (wave_reduce_multi); the arithmetic of every value is unchanged.
template <int M>
__device__ __forceinline__ void wave_two_means(const DataView &dv, const uint32_t *rows, float *s_p, float *s_q,
                                               float *s_k, LeafHdr &ph, LeafHdr &qh, uint32_t fd, uint32_t fpitch,
                                               uint32_t lane) {
    const bool cosine = two_means_is_cosine(dv.metric);
    constexpr bool kInit = M == AH_COSINE || M == AH_DOT_PRODUCT;  // D::init is a no-op for the other metrics
    constexpr int OP = (M == AH_EUCLIDEAN || M == AH_MANHATTAN) ? OP_EUCLID : OP_DOT;
    constexpr bool MH = M == AH_MANHATTAN;
    ph = wave_load_leaf(dv, rows[0], s_p, fd, fpitch, lane);
    qh = wave_load_leaf(dv, rows[1], s_q, fd, fpitch, lane);
    const bool fused = fd >= 32;
    float red[4];
    if (cosine) {
        if (fused) {  // both norms in one pass, then both D::normalize (mod.rs:76-82; dot_product.rs:85-92)
            wave_reduce_multi<OP_DOT, false>(s_p, s_p, s_q, s_q, s_q, s_q, s_q, s_q, 2, fd, lane, red);
            float np = red[0], nq = red[1];
            if (M == AH_DOT_PRODUCT) {
                np = f_add(np, f_mul(ph.h0, ph.h0));
                nq = f_add(nq, f_mul(qh.h0, qh.h0));
            }
            np = f_sqrt(np);
            nq = f_sqrt(nq);
            __syncthreads();
            if (np > 0.0f) {
                for (uint32_t i = lane; i < fd; i += 64) s_p[i] = f_div(s_p[i], np);
                if (M == AH_DOT_PRODUCT) ph.h0 = f_div(ph.h0, np);
            }
            if (nq > 0.0f) {
                for (uint32_t i = lane; i < fd; i += 64) s_q[i] = f_div(s_q[i], nq);
                if (M == AH_DOT_PRODUCT) qh.h0 = f_div(qh.h0, nq);
            }
            __syncthreads();
        } else {
            wave_normalize<M>(s_p, ph, fd, lane);
            wave_normalize<M>(s_q, qh, fd, lane);
        }
    }
    // D::init of p and q: pending until the next fused pass (a centroid's header is only read by the distances)
    bool p_dirty = kInit, q_dirty = kInit;
    if (!fused) {
        wave_init<M>(s_p, ph, fd, lane);
        wave_init<M>(s_q, qh, fd, lane);
        p_dirty = q_dirty = false;
    } else if (kInit) {
        wave_reduce_multi<OP_DOT, false>(s_p, s_p, s_q, s_q, s_q, s_q, s_q, s_q, 2, fd, lane, red);
        init_from_reduction<M>(red[0], ph);
        init_from_reduction<M>(red[1], qh);
        p_dirty = q_dirty = false;
    }
    float ic = 1.0f, jc = 1.0f;
    // The ten sampled rows are known up front and every iteration used to start with a trip to HBM for its row (a random 3 KB
    // row: ~3 us with the TLB miss) that nothing in the wave could hide: the next row is requested into registers — 12 per lane
    // at 768 dimensions — before this iteration's arithmetic and lands in LDS when the next one starts.
    // k_forest_create_split: 48.7 -> 34.3 ms per 10M x 100-tree build (102 registers: four waves per SIMD, where LDS allowed
    // 17 per compute unit before).  Measured and not kept: a second row in flight (two buffers used in turn, 120 registers):
    // 35.2 ms; the first request issued before the centroids' own rows: 36.3 ms (118 registers through the prologue).
    // f32 rows of up to 1024 floats; longer rows and the 1-bit metrics load as before.
    constexpr uint32_t kPre = 16;
    const bool prefetch = !metric_is_bq_dev(dv.metric) && fpitch <= 64u * kPre;
    float pre[kPre];
    LeafHdr pre_h = {0.0f, 0.0f};
    auto request = [&](uint64_t row) {
        const float *rp = dv.rows_f32 + row * dv.pitch;
#pragma unroll
        for (uint32_t c = 0; c < kPre; c++) {
            const uint32_t i = lane + 64u * c;
            pre[c] = i < dv.dims ? rp[i] : 0.0f;
        }
        pre_h.h0 = dv.headers[row * (dv.metric == AH_DOT_PRODUCT ? 2u : 1u)];
        pre_h.h1 = dv.metric == AH_DOT_PRODUCT ? dv.headers[2 * row + 1] : 0.0f;
    };
    if (prefetch) request(rows[2]);
    for (int it = 0; it < 10; it++) {
        LeafHdr kh;
        if (prefetch) {
#pragma unroll
            for (uint32_t c = 0; c < kPre; c++) {
                const uint32_t i = lane + 64u * c;
                if (i < fpitch) s_k[i] = pre[c];
            }
            kh = pre_h;
            __syncthreads();
            if (it < 9) request(rows[3 + it]);
        } else {
            kh = wave_load_leaf(dv, rows[2 + it], s_k, fd, fpitch, lane);
        }
        float di, dj, norm;
        if (fused) {
            // slot 0: (p, k), slot 1: (q, k), slot 2: <k, k> (cosine family), slot 3: the pending D::init of p or q
            const float *dirty = p_dirty ? s_p : s_q;
            const uint32_t n = (p_dirty || q_dirty) ? 4u : (cosine ? 3u : 2u);
            wave_reduce_multi<OP, MH>(s_p, s_k, s_q, s_k, s_k, s_k, dirty, dirty, n, fd, lane, red);
            if (p_dirty) init_from_reduction<M>(red[3], ph);
            else if (q_dirty) init_from_reduction<M>(red[3], qh);
            p_dirty = q_dirty = false;
            di = f_mul(ic, non_built_from_reduction<M>(red[0], ph, kh));
            dj = f_mul(jc, non_built_from_reduction<M>(red[1], qh, kh));
            float kk = red[2];
            if (M == AH_DOT_PRODUCT) kk = f_add(kk, f_mul(kh.h0, kh.h0));
            norm = cosine ? f_sqrt(kk) : 1.0f;
        } else {
            di = f_mul(ic, wave_non_built_distance<M>(s_p, ph, s_k, kh, fd, lane));
            dj = f_mul(jc, wave_non_built_distance<M>(s_q, qh, s_k, kh, fd, lane));
            norm = cosine ? wave_norm<M>(s_k, kh, fd, lane) : 1.0f;
        }
        __syncthreads();
        if (norm != norm || norm <= 0.0f) continue;  // mod.rs:156-158
        if (di < dj) {
            const float c1 = f_add(ic, 1.0f);  // update_mean (mod.rs:86-94): (x*c + n/norm) / (c+1)
            for (uint32_t i = lane; i < fd; i += 64)
                s_p[i] = f_div(f_add(f_mul(s_p[i], ic), f_div(s_k[i], norm)), c1);
            __syncthreads();
            if (fused) p_dirty = kInit;
            else wave_init<M>(s_p, ph, fd, lane);
            ic = f_add(ic, 1.0f);
        } else if (dj < di) {
            const float c1 = f_add(jc, 1.0f);
            for (uint32_t i = lane; i < fd; i += 64)
                s_q[i] = f_div(f_add(f_mul(s_q[i], jc), f_div(s_k[i], norm)), c1);
            __syncthreads();
            if (fused) q_dirty = kInit;
            else wave_init<M>(s_q, qh, fd, lane);
            jc = f_add(jc, 1.0f);
        }
        __syncthreads();
    }
    if (p_dirty) wave_init<M>(s_p, ph, fd, lane);  // the last move's D::init
    if (q_dirty) wave_init<M>(s_q, qh, fd, lane);
}

// create_split.  Writes the normal in the metric's codec to `out_vec` (global: pitch floats or pitch words)
// and its header to out_hdr[0..1].  M = f32_space_metric(dv.metric).
//   euclidean.rs:55-77, manhattan.rs:58-80, cosine.rs:73-85, dot_product.rs:98-113,
//   binary_quantized_cosine.rs:77-93, binary_quantized_euclidean.rs:66-88, binary_quantized_manhattan.rs:70-92
template <int M>
__device__ __forceinline__ void wave_create_split(const DataView &dv, const uint32_t *rows, float *s_p, float *s_q,
                                                  float *s_k, void *out_vec, float *out_hdr, uint32_t lane) {
    const uint32_t fd = f32_space_dims(dv.metric, dv.dims);
    const uint32_t fpitch = f32_space_pitch(dv.metric, dv.dims);
    LeafHdr ph, qh;
    wave_two_means<M>(dv, rows, s_p, s_q, s_k, ph, qh, fd, fpitch, lane);
    // normal = p - q (into s_k)
    for (uint32_t i = lane; i < fpitch; i += 64) s_k[i] = i < fd ? f_sub(s_p[i], s_q[i]) : 0.0f;
    __syncthreads();
    if (!metric_is_bq_dev(dv.metric)) {
        LeafHdr nh = {0.0f, 0.0f};
        if (M == AH_DOT_PRODUCT) nh.h0 = f_sub(ph.h0, qh.h0);
        wave_normalize<M>(s_k, nh, fd, lane);
        if (M == AH_EUCLIDEAN || M == AH_MANHATTAN) {
            // bias = sum_i (-n_i * (p_i + q_i)) / 2.0, sequential f32 sum (euclidean.rs:69-74)
            // (the terms are elementwise: the lanes compute them in parallel into s_p — p is dead after this — and only the
            // sum itself is sequential; the serial loop over three LDS operands was most of an Euclidean create_split)
            for (uint32_t i = lane; i < fd; i += 64) s_p[i] = f_div(f_mul(-s_k[i], f_add(s_p[i], s_q[i])), 2.0f);
            __syncthreads();
            float bias = 0.0f;
#pragma unroll 8
            for (uint32_t i = 0; i < fd; i++) bias = f_add(bias, s_p[i]);
            nh.h0 = bias;
        }
        float *o = reinterpret_cast<float *>(out_vec);
        for (uint32_t i = lane; i < dv.pitch; i += 64) o[i] = i < dv.dims ? s_k[i] : 0.0f;
        if (lane == 0) {
            out_hdr[0] = nh.h0;
            out_hdr[1] = nh.h1;
        }
    } else {
        // UnalignedVector::<BinaryQuantized>::from_vec(p - q): sign bits.  Self::normalize divides the +-1
        // values by a positive norm (or skips when the norm is NaN / <= 0) and re-quantises: same bits.
        uint64_t *o = reinterpret_cast<uint64_t *>(out_vec);
        float bias = 0.0f;
        if (dv.metric != AH_BQ_COSINE) {
            // bias over the RE-QUANTISED n, p, q (bq_euclidean.rs:79-85); +-1 values, exact in any order,
            // summed sequentially like the reference
            for (uint32_t i = 0; i < fd; i++) {
                float nn = (__float_as_uint(s_k[i]) >> 31) ? -1.0f : 1.0f;
                float pp = (__float_as_uint(s_p[i]) >> 31) ? -1.0f : 1.0f;
                float qq = (__float_as_uint(s_q[i]) >> 31) ? -1.0f : 1.0f;
                bias = f_add(bias, f_div(f_mul(-nn, f_add(pp, qq)), 2.0f));
            }
        }
        for (uint32_t w = lane; w < dv.pitch; w += 64) {
            uint64_t word = 0;
            if (w < dv.words)
                for (uint32_t b = 0; b < 64; b++) word |= (uint64_t)((__float_as_uint(s_k[64 * w + b]) >> 31) == 0u) << b;
            o[w] = word;
        }
        if (lane == 0) {
            out_hdr[0] = dv.metric == AH_BQ_COSINE ? 0.0f : bias;
            out_hdr[1] = 0.0f;
        }
    }
    __syncthreads();
}

// Dispatch on the f32-space metric.
__device__ __forceinline__ void wave_create_split_any(const DataView &dv, const uint32_t *rows, float *s_p, float *s_q,
                                                      float *s_k, void *out_vec, float *out_hdr, uint32_t lane) {
    switch (f32_space_metric(dv.metric)) {
    case AH_EUCLIDEAN: wave_create_split<AH_EUCLIDEAN>(dv, rows, s_p, s_q, s_k, out_vec, out_hdr, lane); break;
    case AH_MANHATTAN: wave_create_split<AH_MANHATTAN>(dv, rows, s_p, s_q, s_k, out_vec, out_hdr, lane); break;
    case AH_COSINE: wave_create_split<AH_COSINE>(dv, rows, s_p, s_q, s_k, out_vec, out_hdr, lane); break;
    default: wave_create_split<AH_DOT_PRODUCT>(dv, rows, s_p, s_q, s_k, out_vec, out_hdr, lane); break;
    }
}

}  // namespace ah
