/*
 * arroy_hip_policy.h — the two *policies* that are part of the public contract of
 * libarroy_hip.so and that both the device code and any host (the Rust shim, the
 * CPU oracle under oracle/, the benchmark harness) must be able to evaluate
 * bit-identically:
 *
 *   1. the counter-based randomness policy that replaces arroy's `R: Rng` inside the
 *      tree build (reference call sites: src/parallel.rs:342-367 `choose_two`/`choose`,
 *      src/writer.rs:1310-1326 `randomly_split_children`, src/lib.rs:135 `Side::random`);
 *   2. the counter-based synthetic vector generator of the benchmark harness
 *      (keyed (seed, item, dim) so host and device materialise the
 *      same data without a 30 GB transfer).
 *
 * Why a *policy* and not rand 0.8: the reference consumes one sequential ChaCha12
 * stream per task in DFS order (src/writer.rs:1235-1254).  A level-synchronous GPU
 * build visits nodes in BFS order, so any sequential generator would hand different
 * numbers to the same node.  Here every draw is a pure function of
 * (tree_seed, node path, attempt, draw index): the DFS CPU oracle and the BFS GPU
 * build therefore sample the *same* items for the *same* node and must produce the
 * same forest bit for bit.  The host stays in charge of `tree_seeds` (the reference
 * seeds each task with `StdRng::from_seed(rng.gen())`, src/writer.rs:575,795).
 *
 * Plain C99, no dependencies; usable from C, C++ and HIP device code.
 */
#ifndef ARROY_HIP_POLICY_H
#define ARROY_HIP_POLICY_H

#include <stdint.h>

#if defined(__HIPCC__)
#define AH_HD __host__ __device__ static inline
#else
#define AH_HD static inline
#endif

/* splitmix64 finaliser (Steele/Lea/Flood 2014; public domain constants). */
AH_HD uint64_t ah_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* Key of the root node of the tree seeded with `tree_seed`. */
AH_HD uint64_t ah_node_key_root(uint64_t tree_seed) { return ah_mix64(tree_seed ^ 0xA5A5A5A55A5A5A5Aull); }

/* Key of a child: side 0 = left, 1 = right.  A hash chain, not a heap index, because a
 * 95/5 split policy (src/writer.rs:1209) allows depths far beyond 64. */
AH_HD uint64_t ah_node_key_child(uint64_t parent_key, uint32_t side) {
    return ah_mix64(parent_key * 0xD6E8FEB86659FD93ull + 2u * (uint64_t)side + 1u);
}

/* One 64-bit draw.  `attempt` is the split attempt (0..3, src/writer.rs:1193-1216);
 * `draw` is the index of the draw inside one `create_split`:
 *   0,1   -> choose_two            (src/parallel.rs:342-355)
 *   2..11 -> the ten `choose` calls of two_means (src/distance/mod.rs:151-152) */
AH_HD uint64_t ah_draw(uint64_t node_key, uint32_t attempt, uint32_t draw) {
    return ah_mix64(ah_mix64(node_key + 0x632BE59BD9B4E019ull * (uint64_t)(attempt + 1u)) ^ (uint64_t)draw);
}

/* Unbiased-enough map of a 64-bit draw to [0, n): high half of the 128-bit product
 * (bias <= n / 2^64). */
AH_HD uint64_t ah_bounded(uint64_t r, uint64_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(r, n);
#else
    /* portable C99: high 64 bits of the 128-bit product from four 32x32 partial products */
    const uint64_t a_lo = r & 0xFFFFFFFFu, a_hi = r >> 32, b_lo = n & 0xFFFFFFFFu, b_hi = n >> 32;
    const uint64_t p0 = a_lo * b_lo, p1 = a_lo * b_hi, p2 = a_hi * b_lo, p3 = a_hi * b_hi;
    const uint64_t mid = (p0 >> 32) + (p1 & 0xFFFFFFFFu) + (p2 & 0xFFFFFFFFu);
    return p3 + (p1 >> 32) + (p2 >> 32) + (mid >> 32);
#endif
}

/* choose_two: two distinct ranks in [0, len), len >= 2. */
AH_HD void ah_choose_two(uint64_t node_key, uint32_t attempt, uint64_t len, uint64_t *first, uint64_t *second) {
    uint64_t a = ah_bounded(ah_draw(node_key, attempt, 0u), len);
    uint64_t b = ah_bounded(ah_draw(node_key, attempt, 1u), len - 1u);
    if (b >= a) b += 1u;
    *first = a;
    *second = b;
}

/* choose: rank in [0, len) for two-means iteration `iter` (0..9). */
AH_HD uint64_t ah_choose(uint64_t node_key, uint32_t attempt, uint32_t iter, uint64_t len) {
    return ah_bounded(ah_draw(node_key, attempt, 2u + iter), len);
}

/* randomly_split_children: the coin of the item of rank `rank` inside the node
 * (ascending item id order, src/writer.rs:1320-1325).  1 = Left (the reference maps
 * `true` to `Side::Left`, src/lib.rs:135-141). */
AH_HD uint32_t ah_random_side_is_left(uint64_t node_key, uint64_t rank) {
    return (uint32_t)(ah_mix64(ah_mix64(node_key ^ 0x1F83D9ABFB41BD6Bull) + rank) >> 63);
}

/* Incremental routing (src/writer.rs:1398-1459): the coin of a new item at a split node whose normal is
 * `None` (`randomly_split_children`, :1421-1423).  The reference consumes a per-tree sequential RNG
 * (`R::seed_from_u64(seed + root)`, :1133) in bitmap order; here the coin is a pure function of
 * (tree seed, node, item id) so every item can be routed independently.  1 = Left. */
AH_HD uint32_t ah_route_side_is_left(uint64_t tree_seed, uint32_t node, uint32_t item_id) {
    return (uint32_t)(ah_mix64(ah_mix64(tree_seed ^ 0x3C6EF372FE94F82Bull) + ((uint64_t)node << 32 | item_id)) >> 63);
}

/* ---- synthetic vectors ------------------------------------------------------------ */

enum ah_synth_distribution {
    AH_SYNTH_UNIFORM_01 = 0,    /* i.i.d. uniform [0,1): what the reference's tests use
                                   (src/tests/writer.rs:302 `rng.gen()`)                     */
    AH_SYNTH_UNIFORM_PM1 = 1,   /* i.i.d. uniform [-1,1): sign-balanced margins for cosine   */
    AH_SYNTH_NORMAL = 2,        /* i.i.d. ~N(0,1): the sum of twelve uniforms minus 6
                                   (Irwin-Hall: mean 0, variance 1, tails out to +-6), computed in integers so that host
                                   and device agree bit for bit — no libm                      */
    AH_SYNTH_NORMAL_OUTLIERS = 3, /* the same with a few "outlier dimensions" (dim % 97 == 13) scaled by 20, the shape
                                   real embedding models show: one scale per dataset or per row then wastes the 8 bits of a
                                   quantised copy on them                                      */
    /* The two below are NOT i.i.d. per component: the shapes imported embeddings have (the reference's users feed it
     * real vectors, examples/import-vectors.rs:71-101).  Margins crowd the split planes, `split_imbalance` retries and
     * the random fallback fire (src/writer.rs:1209-1233,1310-1326), and a certified screen decides least here. */
    AH_SYNTH_CLUSTERED = 4,     /* AH_SYNTH_CLUSTERS centres ~N(0,1); a row = its centre + N(0,1)/16.  Cluster sizes are
                                   skewed (cluster = floor(4096 u^2): the largest holds 1/64 of the rows, the smallest
                                   1/8192), and one row in 61 is an exact copy of its centre — duplicates no plane can
                                   separate, as real corpora hold them                          */
    AH_SYNTH_LOW_RANK = 5       /* AH_SYNTH_FACTORS latent factors per row times one fixed factors x dims loading matrix,
                                   plus ~7 % noise: every row lies near a 32-dimensional subspace  */
};
#define AH_SYNTH_LAST AH_SYNTH_LOW_RANK
#define AH_SYNTH_CLUSTERS 4096u
#define AH_SYNTH_FACTORS 32u

/* ~N(0,1) in units of 2^-20 from one 64-bit hash: twelve 20-bit uniforms out of four 64-bit draws (three each), their
 * sum < 12 * 2^20 < 2^24; the +6 centres the twelve half-open cells.  |result| < 6 * 2^20. */
AH_HD int32_t ah_synth_normal_q20(uint64_t h) {
    uint32_t sum = 0;
    uint32_t k;
    for (k = 0; k < 4; k++) {
        const uint64_t g = k == 0 ? h : ah_mix64(h + 0x9E3779B97F4A7C15ull * (uint64_t)k);
        sum += (uint32_t)(g & 0xFFFFFu) + (uint32_t)((g >> 20) & 0xFFFFFu) + (uint32_t)((g >> 40) & 0xFFFFFu);
    }
    return (int32_t)sum - 6 * 1048576 + 6;
}

/* the hash every distribution draws the (item, dim) component from */
AH_HD uint64_t ah_synth_hash(uint64_t seed, uint64_t item, uint32_t dim, uint32_t dims) {
    return ah_mix64(ah_mix64(seed) + item * (uint64_t)dims + (uint64_t)dim);
}

/* AH_SYNTH_CLUSTERED: the cluster of a row, and whether the row is an exact copy of the cluster's centre */
AH_HD uint32_t ah_synth_cluster_of(uint64_t seed, uint64_t item, int *is_copy) {
    const uint64_t g = ah_mix64(ah_mix64(seed ^ 0x7C1B5A3D9E8F6024ull) + item);
    const uint64_t u = g >> 40;                              /* 24 random bits */
    *is_copy = (uint32_t)(g & 0xFFFFu) % 61u == 0u;
    return (uint32_t)((u * u) >> 36);                        /* floor(4096 (u / 2^24)^2) */
}
/* component `dim` of centre `cluster`, ~N(0,1) in units of 2^-20 */
AH_HD int32_t ah_synth_centre_q20(uint64_t seed, uint32_t cluster, uint32_t dim, uint32_t dims) {
    return ah_synth_normal_q20(ah_mix64(ah_mix64(seed ^ 0x51ED270B3A7C9D15ull) + (uint64_t)cluster * dims + dim));
}
/* a row component from its centre's and its own noise draw: both products are exact (24-bit integers times powers of
 * two), so the value is ONE rounded f32 addition — the same with or without a fused multiply-add */
AH_HD float ah_synth_clustered_from(int32_t centre_q20, int32_t noise_q20) {
    return (float)centre_q20 * (1.0f / 1048576.0f) + (float)noise_q20 * (1.0f / 16777216.0f);
}

/* AH_SYNTH_LOW_RANK, all in integers: factor k of a row in [-126, 126] (four 6-bit uniforms), loading (k, dim) in
 * [-510, 510] (four 8-bit uniforms); |sum over 32 factors| <= 32 * 126 * 510 < 2^21. */
AH_HD int32_t ah_synth_factor(uint64_t seed, uint64_t item, uint32_t k) {
    const uint64_t g = ah_mix64(ah_mix64(seed ^ 0x2545F4914F6CDD1Dull) + item * AH_SYNTH_FACTORS + k);
    return (int32_t)((g & 63u) + ((g >> 6) & 63u) + ((g >> 12) & 63u) + ((g >> 18) & 63u)) - 126;
}
AH_HD int32_t ah_synth_loading(uint64_t seed, uint32_t k, uint32_t dim, uint32_t dims) {
    const uint64_t g = ah_mix64(ah_mix64(seed ^ 0x6A09E667F3BCC908ull) + (uint64_t)k * dims + dim);
    return (int32_t)((g & 255u) + ((g >> 8) & 255u) + ((g >> 16) & 255u) + ((g >> 24) & 255u)) - 510;
}
/* signal (sigma ~ 31 000) + noise (sigma 2 048: the row's N(0,1) draw / 512), scaled to sigma ~ 1: an integer below 2^24
 * times 2^-15, exact */
AH_HD float ah_synth_low_rank_from(int32_t signal, int32_t noise_q20) {
    return (float)(signal + noise_q20 / 512) * (1.0f / 32768.0f);
}

/* value of component `dim` of item `item` (item = row index, ids are 0..N-1). Exact in f32:
 * a 24-bit integer scaled by a power of two (CLUSTERED: the sum of two such, one rounding), so host and device agree
 * bit for bit. */
AH_HD float ah_synth_value(uint64_t seed, uint64_t item, uint32_t dim, uint32_t dims, int distribution) {
    uint64_t h = ah_synth_hash(seed, item, dim, dims);
    if (distribution == AH_SYNTH_NORMAL || distribution == AH_SYNTH_NORMAL_OUTLIERS) {
        /* |value| < 6, a multiple of 2^-20 */
        float z = (float)ah_synth_normal_q20(h) * (1.0f / 1048576.0f);
        if (distribution == AH_SYNTH_NORMAL_OUTLIERS && dim % 97u == 13u) z *= 20.0f; /* one IEEE multiplication: the same rounding everywhere */
        return z;
    }
    if (distribution == AH_SYNTH_CLUSTERED) {
        int is_copy;
        const uint32_t c = ah_synth_cluster_of(seed, item, &is_copy);
        return ah_synth_clustered_from(ah_synth_centre_q20(seed, c, dim, dims), is_copy ? 0 : ah_synth_normal_q20(h));
    }
    if (distribution == AH_SYNTH_LOW_RANK) {
        int32_t signal = 0;
        uint32_t k;
        for (k = 0; k < AH_SYNTH_FACTORS; k++) signal += ah_synth_factor(seed, item, k) * ah_synth_loading(seed, k, dim, dims);
        return ah_synth_low_rank_from(signal, ah_synth_normal_q20(h));
    }
    {
        uint32_t m = (uint32_t)(h >> 40);                 /* 24 random bits */
        float u = (float)m * (1.0f / 16777216.0f);       /* [0,1) exactly */
        if (distribution == AH_SYNTH_UNIFORM_PM1) {
            return u * 2.0f - 1.0f;                       /* exact: 25-bit grid in [-1,1) */
        }
        return u;
    }
}

/* Host side: rows first_item .. first_item + n - 1 into out[n x dims], the values of ah_synth_value, with the per-row and
 * per-dataset parts of the two structured distributions computed once (`table`: NULL, or what ah_synth_table_len()
 * int32 hold after ah_synth_table_fill(); without it every component is computed from scratch). */
AH_HD uint64_t ah_synth_table_len(uint32_t dims, int distribution) {
    if (distribution == AH_SYNTH_CLUSTERED) return (uint64_t)AH_SYNTH_CLUSTERS * dims;
    if (distribution == AH_SYNTH_LOW_RANK) return (uint64_t)AH_SYNTH_FACTORS * dims;
    return 0;
}
AH_HD void ah_synth_table_fill(uint64_t seed, uint32_t dims, int distribution, int32_t *table) {
    uint32_t a, d;
    if (distribution == AH_SYNTH_CLUSTERED)
        for (a = 0; a < AH_SYNTH_CLUSTERS; a++)
            for (d = 0; d < dims; d++) table[(uint64_t)a * dims + d] = ah_synth_centre_q20(seed, a, d, dims);
    if (distribution == AH_SYNTH_LOW_RANK)
        for (a = 0; a < AH_SYNTH_FACTORS; a++)
            for (d = 0; d < dims; d++) table[(uint64_t)a * dims + d] = ah_synth_loading(seed, a, d, dims);
}
AH_HD void ah_synth_row(uint64_t seed, uint64_t item, uint32_t dims, int distribution, const int32_t *table, float *out) {
    uint32_t d, k;
    if (table && distribution == AH_SYNTH_CLUSTERED) {
        int is_copy;
        const int32_t *centre = table + (uint64_t)ah_synth_cluster_of(seed, item, &is_copy) * dims;
        for (d = 0; d < dims; d++)
            out[d] = ah_synth_clustered_from(centre[d], is_copy ? 0 : ah_synth_normal_q20(ah_synth_hash(seed, item, d, dims)));
    } else if (table && distribution == AH_SYNTH_LOW_RANK) {
        int32_t f[AH_SYNTH_FACTORS];
        for (k = 0; k < AH_SYNTH_FACTORS; k++) f[k] = ah_synth_factor(seed, item, k);
        for (d = 0; d < dims; d++) {
            int32_t signal = 0;
            for (k = 0; k < AH_SYNTH_FACTORS; k++) signal += f[k] * table[(uint64_t)k * dims + d];
            out[d] = ah_synth_low_rank_from(signal, ah_synth_normal_q20(ah_synth_hash(seed, item, d, dims)));
        }
    } else {
        for (d = 0; d < dims; d++) out[d] = ah_synth_value(seed, item, d, dims, distribution);
    }
}

#endif /* ARROY_HIP_POLICY_H */
