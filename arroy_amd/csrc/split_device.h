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

// two_means (mod.rs:126-171 / 173-223).  s_p, s_q, s_k: LDS, fpitch floats each.  rows[12] = the sampled
// dataset rows (choose_two, then the ten `choose`).  Returns the two centroid headers.
template <int M>
__device__ __forceinline__ void wave_two_means(const DataView &dv, const uint32_t *rows, float *s_p, float *s_q,
                                               float *s_k, LeafHdr &ph, LeafHdr &qh, uint32_t fd, uint32_t fpitch,
                                               uint32_t lane) {
    const bool cosine = two_means_is_cosine(dv.metric);
    ph = wave_load_leaf(dv, rows[0], s_p, fd, fpitch, lane);
    qh = wave_load_leaf(dv, rows[1], s_q, fd, fpitch, lane);
    if (cosine) {
        wave_normalize<M>(s_p, ph, fd, lane);
        wave_normalize<M>(s_q, qh, fd, lane);
    }
    wave_init<M>(s_p, ph, fd, lane);
    wave_init<M>(s_q, qh, fd, lane);
    float ic = 1.0f, jc = 1.0f;
    for (int it = 0; it < 10; it++) {
        LeafHdr kh = wave_load_leaf(dv, rows[2 + it], s_k, fd, fpitch, lane);
        float di = f_mul(ic, wave_non_built_distance<M>(s_p, ph, s_k, kh, fd, lane));
        float dj = f_mul(jc, wave_non_built_distance<M>(s_q, qh, s_k, kh, fd, lane));
        float norm = cosine ? wave_norm<M>(s_k, kh, fd, lane) : 1.0f;
        __syncthreads();
        if (norm != norm || norm <= 0.0f) continue;  // mod.rs:156-158
        if (di < dj) {
            const float c1 = f_add(ic, 1.0f);  // update_mean (mod.rs:86-94): (x*c + n/norm) / (c+1)
            for (uint32_t i = lane; i < fd; i += 64)
                s_p[i] = f_div(f_add(f_mul(s_p[i], ic), f_div(s_k[i], norm)), c1);
            __syncthreads();
            wave_init<M>(s_p, ph, fd, lane);
            ic = f_add(ic, 1.0f);
        } else if (dj < di) {
            const float c1 = f_add(jc, 1.0f);
            for (uint32_t i = lane; i < fd; i += 64)
                s_q[i] = f_div(f_add(f_mul(s_q[i], jc), f_div(s_k[i], norm)), c1);
            __syncthreads();
            wave_init<M>(s_q, qh, fd, lane);
            jc = f_add(jc, 1.0f);
        }
        __syncthreads();
    }
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
            float bias = 0.0f;
            for (uint32_t i = 0; i < fd; i++)
                bias = f_add(bias, f_div(f_mul(-s_k[i], f_add(s_p[i], s_q[i])), 2.0f));
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
