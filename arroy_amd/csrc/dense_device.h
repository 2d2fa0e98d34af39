// dense_device.h — the top levels of the forest build as ONE matrix product on the MFMA units.
//
// Included by forest.hip (after FNode / ScreenCounters / AbortFlags).  Replaces, for the levels where it is cheaper,
// the row-major passes of the margin loop (src/writer.rs:1201-1207 for every pending node of every tree).
//
// A single margin <normal, item> is a vector contraction (0.5 flop / byte) and the reference's f32 value has a fixed
// summation order no matrix unit reproduces — which is why the distance scan, the re-rank and the exact margins never
// touch MFMA.  The certified screen (screen_device.h) is different on both counts: it needs the binary16 dot product
// in ANY accumulation order (its rounding is covered by gamma_s), and at the top of the forest the normals are few:
// level L has n_trees * 2^L of them, every one of the N rows meets one per tree.  So the screen values of a whole
// level are the product
//
//        S[N x C] = X~[N x hpitch] * N~[C x hpitch]^T          (C = nodes of the level, all trees)
//
// of the binary16 shadow of the rows with the binary16 shadow of the level's normals — both K-major, the layout
// v_mfma_f32_32x32x16_f16 wants — of which row r needs the n_trees entries S[r, node_of[t][r]].  Computing all C
// columns wastes a factor 2^L of arithmetic, but the matrix units deliver ~50x the multiply-adds of the v_dot2c
// row-major pass and the rows leave HBM once per LEVEL instead of once per group of 8-16 trees: 10M x 768, 100 trees,
// level 0-3: ~3-10 ms instead of 66-70 ms.  Beyond ~64 nodes per tree the waste wins and the row-major / node-major
// passes take over (cost model in build_batch).
//
// Sides stay bit-identical to the reference arithmetic: |S| > E (the same rigorous bound, with gamma_s for a chain of
// hpitch roundings) decides the side, every other pair (~1.3 %) is marked and recomputed by k_forest_exact_pairs in the
// reference's f32 order.
#pragma once

namespace ah {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr uint32_t kDM = 256;  // rows of X~ per block tile
// Block tile = 256 rows x (64 WN) normals, WN = 2 or 4: 2 x WN waves, each owning 128 x 64 = 4 x 2 MFMA tiles (128
// accumulator registers).  WN = 4 (512 threads, two waves per SIMD) is the workhorse; WN = 2 serves levels with at most
// 128 normals, where the wider tile would only multiply padding.
template <int WN>
struct DenseShape {
    static constexpr uint32_t kBN = 64u * WN;
    static constexpr uint32_t kThreads = 128u * WN;
    static constexpr uint32_t kWaves = 2u * WN;
    static constexpr uint32_t kStage = (kDM + kBN) * 128u;  // one k-block (64 halves = 128 B per tile row) of both operands
    static constexpr uint32_t kPiecesA = 32u / kWaves, kPiecesB = (kBN / 8u) / kWaves;  // 1 KiB DMA pieces per wave
};
constexpr uint32_t kDenseHalf = 128;             // columns per epilogue round
constexpr uint32_t kDensePitch = kDM + 4;         // floats per column of the transposed result tile (bank spread)
constexpr uint32_t kDenseLds = kDenseHalf * kDensePitch * 4;  // 133 120 B >= two stages of either shape
constexpr uint32_t kDenseGroup = 8;  // row tiles whose column tiles run back to back on one XCD (X~ tiles stay in its L2)
static_assert(2 * DenseShape<4>::kStage <= kDenseLds && 2 * DenseShape<2>::kStage <= kDenseLds, "stage buffers fit");

// side-byte codes of the dense pass (resolved to 0 / 1 by k_forest_exact_pairs before anything else reads them)
constexpr uint32_t kSideUndecided = 2u;  // the screen could not decide: reference arithmetic wanted
constexpr uint32_t kSideVerify = 4u;     // AH_SCREEN_VERIFY: decided (bit 0 = side), reference arithmetic wanted as a check

struct DenseArgs {
    const uint16_t *rows;  // binary16 shadow of the rows, n x hpitch
    const float4 *stats;   // per row {|x~|, |x - x~|, |x|, 0}
    const float *headers;  // DotProduct: {extra_dim, norm} per row
    uint64_t n;
    uint32_t hpitch;
    const uint8_t *shadow;  // the level's shadow records [hpitch halves][NormalStats], n_cols of them
    uint64_t hstride;
    uint32_t n_cols;
    const FNode *nodes;  // the level's nodes (column c <-> node c; ordered by tree)
    const uint32_t *node_of;
    uint8_t *side_bytes;
    float gamma_s, gamma_r;
    uint32_t n_row_tiles, n_col_tiles, group, verify;
};

// One k-block of the block tile, global -> LDS by the DMA path (global_load_lds_dwordx4: the 64 lanes of a wave
// instruction fill 1 KiB of LDS lane-linearly = 8 tile rows of 128 bytes; every lane fetches a 16-byte chunk of a
// whole 128-byte line, so the global side is fully coalesced).  Slot s of tile row R holds chunk s ^ ((R >> 1) & 7):
// the XOR is applied to the SOURCE address here and again to the ds_read_b128 address of the fragment loads, which makes
// those conflict-free (their 16-lane groups — rows {0-3,12-15,20-27} etc. at one chunk — then cover all 64 banks).
template <int WN>
__device__ __forceinline__ void dense_stage(const uint8_t *const (&a_src)[DenseShape<WN>::kPiecesA],
                                            const uint8_t *const (&b_src)[DenseShape<WN>::kPiecesB], uint32_t kb, uint8_t *stage,
                                            uint32_t wave) {
    typedef DenseShape<WN> SH;
    const uint64_t koff = (uint64_t)kb * 128u;
#pragma unroll
    for (uint32_t i = 0; i < SH::kPiecesA; i++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src[i] + koff),
                                         (__attribute__((address_space(3))) void *)(stage + (i * SH::kWaves + wave) * 1024u), 16, 0,
                                         0);
#pragma unroll
    for (uint32_t i = 0; i < SH::kPiecesB; i++)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(b_src[i] + koff),
            (__attribute__((address_space(3))) void *)(stage + kDM * 128u + (i * SH::kWaves + wave) * 1024u), 16, 0, 0);
}

// block -> (row tile, column tile); false for the padding blocks of the last group.  Workgroups go to the XCDs
// round-robin, so block b runs on XCD b & 7: the blocks of one XCD walk groups of `group` row tiles, all column tiles of a
// group back to back.  Shared with k_dense_coverage (ah_debug_launch_coverage).
__device__ __forceinline__ bool dense_block_map(uint32_t b, uint32_t group, uint32_t n_col_tiles, uint32_t n_row_tiles,
                                                uint32_t &rt, uint32_t &ct) {
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t per_group = group * n_col_tiles;
    const uint32_t grp = slot / per_group, within = slot % per_group;
    ct = within / group;
    rt = (grp * group + within % group) * 8u + xcd;
    return rt < n_row_tiles;
}
__global__ void k_dense_coverage(uint32_t group, uint32_t n_col_tiles, uint32_t n_row_tiles, uint32_t *__restrict__ counts) {
    uint32_t rt, ct;
    if (!dense_block_map(blockIdx.x, group, n_col_tiles, n_row_tiles, rt, ct)) return;
    if (threadIdx.x == 0) atomicAdd(&counts[(uint64_t)rt * n_col_tiles + ct], 1u);
}

template <int METRIC, int WN>
__global__ __launch_bounds__(DenseShape<WN>::kThreads, 1) void k_forest_dense_screen(DenseArgs a, const AbortFlags abort_flag) {
    typedef DenseShape<WN> SH;
    extern __shared__ uint4 s_dense4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(s_dense4);
    if (abort_requested(abort_flag)) return;
    // block -> (row tile, column tile).  Workgroups go to the XCDs round-robin, so block b runs on XCD b & 7: the blocks
    // of one XCD walk groups of `group` row tiles, all column tiles of a group back to back, and the group's X~ tiles
    // (group x 384 KB) are read from HBM once and then found in that XCD's L2.
    uint32_t rt, ct;
    if (!dense_block_map(blockIdx.x, a.group, a.n_col_tiles, a.n_row_tiles, rt, ct)) return;  // block-uniform
    const uint64_t row0 = (uint64_t)rt * kDM;
    const uint32_t c0 = ct * SH::kBN;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;

    // per-lane sources of the DMA pieces of a stage: piece p = kWaves i + wave covers tile rows 8 p .. 8 p + 7
    const uint8_t *a_src[SH::kPiecesA], *b_src[SH::kPiecesB];
    {
        const uint32_t sl = lane & 7u;
#pragma unroll
        for (uint32_t i = 0; i < SH::kPiecesA; i++) {
            const uint32_t R = (i * SH::kWaves + wave) * 8u + (lane >> 3);
            const uint64_t r = min(row0 + R, a.n - 1);  // rows past the end repeat the last row (never stored)
            a_src[i] = reinterpret_cast<const uint8_t *>(a.rows) + r * ((uint64_t)a.hpitch * 2u) + ((sl ^ ((R >> 1) & 7u)) << 4);
        }
#pragma unroll
        for (uint32_t i = 0; i < SH::kPiecesB; i++) {
            const uint32_t R = (i * SH::kWaves + wave) * 8u + (lane >> 3);  // row of the B region
            const uint32_t c = min(c0 + R, a.n_cols - 1);
            b_src[i] = a.shadow + (uint64_t)c * a.hstride + ((sl ^ ((R >> 1) & 7u)) << 4);
        }
    }
    const uint32_t wm = wave / WN, wn = wave % WN;
    const uint32_t m = lane & 31u, g = lane >> 5, swz = (m >> 1) & 7u;
    f32x16_t acc[4][2];
#pragma unroll
    for (int im = 0; im < 4; im++)
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[im][jn][e] = 0.0f;

    const uint32_t nk = a.hpitch >> 6;
    dense_stage<WN>(a_src, b_src, 0, smem, wave);
    for (uint32_t kb = 0; kb < nk; kb++) {
        // this wave's DMA of stage kb has landed; after the barrier everybody's has, and every wave has finished
        // reading the other buffer (its fragment loads were consumed by the MFMAs of iteration kb - 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < nk) dense_stage<WN>(a_src, b_src, kb + 1, smem + ((kb + 1) & 1u) * SH::kStage, wave);
        const uint8_t *st = smem + (kb & 1u) * SH::kStage;
        const uint8_t *sa = st + (wm * 128u + m) * 128u;
        const uint8_t *sb = st + (kDM + wn * 64u + m) * 128u;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            // k-step q of the block: lanes 0-31 take halves [16 q, +8), lanes 32-63 halves [16 q + 8, +8) of their row —
            // the same assignment for both operands, which is all the product needs (any k order: gamma_s covers it)
            const uint32_t off = ((2u * q + g) ^ swz) << 4;
            f16x8_t af[4], bf[2];
#pragma unroll
            for (int im = 0; im < 4; im++) af[im] = *reinterpret_cast<const f16x8_t *>(sa + im * 32 * 128 + off);
#pragma unroll
            for (int jn = 0; jn < 2; jn++) bf[jn] = *reinterpret_cast<const f16x8_t *>(sb + jn * 32 * 128 + off);
#pragma unroll
            for (int im = 0; im < 4; im++)
#pragma unroll
                for (int jn = 0; jn < 2; jn++)
                    acc[im][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[im], bf[jn], acc[im][jn], 0, 0, 0);
        }
    }
    // Epilogue in rounds of 128 columns.  The waves owning them park their accumulators in LDS, transposed — S[column][row];
    // D layout of the 32x32 MFMA: lane -> column (lane & 31) of the B operand (the normals), register e -> row
    // (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the A operand (the data rows): four consecutive registers are four
    // consecutive rows = one 16-byte store.  Then a thread per row (and per share of the trees, with 512 threads): for
    // every tree that has nodes among the round's columns, the row's node (coalesced read), its screen value, the bound,
    // the decision; one side byte out (consecutive rows -> consecutive bytes).  Nodes are ordered by tree, so the
    // columns [c_lo, c_hi] cover the trees [t_lo, t_hi].
    float *S = reinterpret_cast<float *>(smem);
    const uint32_t r_in = threadIdx.x & (kDM - 1u), part = threadIdx.x / kDM;
    constexpr uint32_t kParts = SH::kThreads / kDM;
    const uint64_t row = row0 + r_in;
    const bool live = row < a.n;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    float row_extra = 0.0f;
    if (live) {
        rs = a.stats[row];
        if (METRIC == AH_DOT_PRODUCT) row_extra = a.headers[2 * row];
    }
#pragma unroll 1
    for (uint32_t h = 0; h < SH::kBN / kDenseHalf; h++) {
        const uint32_t c_lo = c0 + h * kDenseHalf;
        if (c_lo >= a.n_cols) break;  // block-uniform
        __syncthreads();  // the stage buffers (round 0) / the previous round's tile are dead
        if ((wn >> 1) == h) {
#pragma unroll
            for (int im = 0; im < 4; im++)
#pragma unroll
                for (int jn = 0; jn < 2; jn++) {
                    const uint32_t col = (wn & 1u) * 64u + (uint32_t)jn * 32u + m;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const uint32_t rr = wm * 128u + (uint32_t)im * 32u + 8u * (uint32_t)qq + 4u * g;
                        *reinterpret_cast<float4 *>(S + col * kDensePitch + rr) = make_float4(
                            acc[im][jn][4 * qq], acc[im][jn][4 * qq + 1], acc[im][jn][4 * qq + 2], acc[im][jn][4 * qq + 3]);
                    }
                }
        }
        __syncthreads();
        const uint32_t c_hi = min(c_lo + kDenseHalf, a.n_cols) - 1u;
        const uint32_t t_lo = a.nodes[c_lo].tree, t_hi = a.nodes[c_hi].tree;
        for (uint32_t t = t_lo + 8u * part; t <= t_hi; t += 8u * kParts) {
            uint32_t nd[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++)
                nd[u] = (live && t + u <= t_hi) ? a.node_of[(uint64_t)(t + u) * a.n + row] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t c = nd[u] - c_lo;  // 0xFFFFFFFF (leaf row) and nodes of other column ranges fall outside
                if (nd[u] != 0xFFFFFFFFu && c < kDenseHalf) {
                    const float s = S[c * kDensePitch + r_in];
                    const NormalStats ns =
                        *reinterpret_cast<const NormalStats *>(a.shadow + (uint64_t)nd[u] * a.hstride + (uint64_t)a.hpitch * 2u);
                    uint32_t side;
                    const bool decided = screen_decides<METRIC>(s, rs, ns, row_extra, a.gamma_s, a.gamma_r, side);
                    a.side_bytes[(uint64_t)(t + u) * a.n + row] =
                        (uint8_t)(decided ? (a.verify ? (kSideVerify | side) : side) : kSideUndecided);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The narrow product (round 5): levels with a few hundred columns at most, where the product is bound by the 2 * hpitch
// bytes per row that leave HBM, not by the matrix units.  k_forest_dense_screen stages BOTH operands through LDS with a
// full `vmcnt(0)` + barrier per k-block and keeps one block of 133 KB per CU: a tile's twelve k-blocks each expose an HBM
// round trip and nothing overlaps a tile's epilogue — 5.5-9 ms per level where the rows' HBM time is 1.9 ms (10M x 768).
// Here
//   * the ROWS never touch LDS: a compute wave owns 32 rows and fetches its A fragments straight into registers, lane
//     (m, g) the half line [64 g, 64 g + 64) of row m per k-block (4 x 16 B; the two lanes of a row cover the 128-byte
//     line), two k-blocks ahead of the one being multiplied — 8 KB in flight per wave, across the barriers (raw
//     `s_barrier`: no memory wait is attached to it);
//   * only the NORMALS (L2-resident: C x 1.6 KB) go through LDS, a ring of three 64-half k-blocks filled by LDS-DMA two
//     ahead by a LOADER wave that does nothing else.  The roles are split because hipcc orders every LDS read behind
//     the latest LDS-DMA of the same wave and, once both kinds of load are pending, waits for `vmcnt(0)`: in a wave that
//     issues both, the register prefetch is drained every third k-block (seen in the ISA).  A wave that only issues DMA and a
//     wave that only loads registers get exact counted waits from the compiler;
//   * k-step i of a k-block multiplies halves [32 g + 8 i, + 8) of both operands (any k order: gamma_s covers it);
//   * a block is 5 compute waves x 32 rows + the loader against 32 NT columns (NT = 2 / 4), 24 / 48 KB of LDS: two blocks
//     share a CU (12 waves of <= 168 registers), so one block's epilogue runs under the other's loads;
//   * the epilogue is wave-private: column tile by column tile the wave parks its 32 x 32 accumulators in LDS
//     (transposed), then lane = row (the two half-waves split the trees) reads the row's node per tree (128-byte
//     coalesced), picks its column, decides, and writes the side byte.
// Same screen values up to the order of the f32 additions (covered by gamma_s), same bound, same marks for
// k_forest_exact_pairs: forests stay bit-identical.
#ifndef AH_NARROW_SETS
#define AH_NARROW_SETS 3u   // A register sets (k-blocks in flight per wave + 1); 4 measured slower: 3.2 -> 4.0 ms per level
#endif
#ifndef AH_NARROW_PRELOAD
#define AH_NARROW_PRELOAD 1  // the epilogue's node indices requested before the k-loop (1) or at the epilogue's start (0)
#endif
constexpr uint32_t kNarrowWaves = 5;                      // compute waves of a block (+ 1 loader wave)
constexpr uint32_t kNarrowRows = 32u * kNarrowWaves;      // rows of X~ per block
constexpr uint32_t kNarrowThreads = 64u * (kNarrowWaves + 1u);
template <int NT>
struct NarrowShape {
    static constexpr uint32_t kCT = 32u * NT;         // columns (normals) per block
    static constexpr uint32_t kStage = kCT * 128u;    // one k-block (64 halves) of the block's normals
    static constexpr uint32_t kStages = 3;
    static constexpr uint32_t kPieces = kCT / 8u;     // 1 KiB DMA pieces per stage
    static constexpr uint32_t kEpiWave = 32u * 36u * 4u;  // per wave: S[32 columns][36] floats (overlays the dead ring)
    static constexpr uint32_t kRing = kStages * kStage > kNarrowWaves * kEpiWave ? kStages * kStage : kNarrowWaves * kEpiWave;
    static constexpr uint32_t kLds = kRing + kCT * 16u;  // + the NormalStats of the tile's columns (written once by the loader)
};

template <int METRIC, int NT, bool STREAM>
__global__ __launch_bounds__(kNarrowThreads, 2) void k_forest_dense_narrow(DenseArgs a, const AbortFlags abort_flag) {
    typedef NarrowShape<NT> SH;
    extern __shared__ uint4 s_dense4[];
    uint8_t *smem = reinterpret_cast<uint8_t *>(s_dense4);
    if (abort_requested(abort_flag)) return;
    uint32_t rt, ct;
    if (!dense_block_map(blockIdx.x, a.group, a.n_col_tiles, a.n_row_tiles, rt, ct)) return;  // block-uniform
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c0 = ct * SH::kCT;
    const uint32_t nk = a.hpitch >> 6;
    if (wave == kNarrowWaves) {
        // ---- loader: piece p of a stage covers the tile's columns 8 p .. 8 p + 7 (slot s of column R holds chunk
        // s ^ ((R >> 1) & 7), as in k_forest_dense_screen: the fragment reads of the compute waves are conflict-free)
        const uint8_t *b_src[SH::kPieces];
#pragma unroll
        for (uint32_t p = 0; p < SH::kPieces; p++) {
            const uint32_t R = p * 8u + (lane >> 3);
            const uint32_t c = min(c0 + R, a.n_cols - 1);
            b_src[p] = a.shadow + (uint64_t)c * a.hstride + (((lane & 7u) ^ ((R >> 1) & 7u)) << 4);
        }
        // the statistics of the tile's normals, for the compute waves' epilogues (visible to them through the k-loop's barriers)
        {
            NormalStats *nst_all = reinterpret_cast<NormalStats *>(smem + SH::kRing);
#pragma unroll
            for (uint32_t h = 0; h < SH::kCT / 64u; h++)
                nst_all[64u * h + lane] = *reinterpret_cast<const NormalStats *>(
                    a.shadow + (uint64_t)min(c0 + 64u * h + lane, a.n_cols - 1) * a.hstride + (uint64_t)a.hpitch * 2u);
        }
#define AH_NARROW_DMA(KB, SET)                                                                                            \
    _Pragma("unroll") for (uint32_t p_ = 0; p_ < SH::kPieces; p_++) __builtin_amdgcn_global_load_lds(                     \
        (const __attribute__((address_space(1))) void *)(b_src[p_] + (uint64_t)(KB) * 128u),                              \
        (__attribute__((address_space(3))) void *)(smem + (SET) * SH::kStage + p_ * 1024u), 16, 0, 0)
        AH_NARROW_DMA(0, 0);
        if (nk > 1) AH_NARROW_DMA(1, 1);
        for (uint32_t kb = 0; kb < nk; kb++) {
            // stage kb has landed (stage kb + 1 may still fly); after the barrier everybody has finished with the buffer
            // of stage kb - 1, which stage kb + 2 overwrites
            if (kb + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SH::kPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kb + 2 < nk) {
                const uint32_t set = (kb + 2) % 3u;
                AH_NARROW_DMA(kb + 2, set);
            }
        }
#undef AH_NARROW_DMA
        __builtin_amdgcn_s_barrier();  // the compute waves' "ring is dead" barrier
        return;
    }
    // v_mfma_f32_16x16x32_f16: lane (m, kg) = (lane & 15, lane >> 4) holds halves [8 kg, 8 kg + 8) of k-step s (32 halves) of
    // row / column m.  The four lanes of a row therefore fetch 64 CONTIGUOUS bytes per load instruction (16 rows x one
    // 64-byte sector: 16 sector look-ups per KiB; the 32x32x16 shape has two lanes per row and costs 64 — measured 2x slower)
    const uint32_t m = lane & 15u, kg = lane >> 4;
    const uint64_t row_base = (uint64_t)rt * kNarrowRows + wave * 32u;
    // A: row tile i (16 rows) of the wave, k-block kb at + 128 kb, k-step s at + 64 s (rows past the end repeat the last
    // row: never stored)
    const uint4 *a_src[2];
#pragma unroll
    for (uint32_t i = 0; i < 2; i++)
        a_src[i] = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.rows) +
                                                  min(row_base + 16u * i + m, a.n - 1) * ((uint64_t)a.hpitch * 2u) + 16u * kg);
    f32x4_t acc[2][2 * NT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2 * NT; j++) acc[i][j] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // the epilogue's row owner: lane & 31 (both half-waves: they split the trees, half h takes trees T0 + h + 2 u)
    const uint32_t er = lane & 31u, eh = lane >> 5;
    const uint64_t row = row_base + er;
    const bool live = row < a.n;
    // Everything the epilogue reads from memory is requested NOW, before the k-loop, and waits in registers: the row's
    // statistics and the row's node in the first 16 trees of the tile (all of them for a 13-tree share); the statistics of
    // the tile's normals wait in LDS (the loader put them there).  An epilogue that starts these loads after the last MFMA exposes one HBM round trip per round of 32
    // columns (measured: 1.1 of the 3.5 ms of a 10M x 768 level).
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    float row_extra = 0.0f;
    const uint32_t T0 = a.nodes[c0].tree, T1 = a.nodes[min(c0 + SH::kCT, a.n_cols) - 1u].tree;
    uint32_t nd0[8];
#define AH_NARROW_LOAD_ND0()                                                                                              \
    _Pragma("unroll") for (uint32_t u = 0; u < 8; u++) {                                                                  \
        const uint32_t t = T0 + eh + 2u * u;                                                                              \
        const uint32_t v = a.node_of[(uint64_t)min(t, T1) * a.n + min(row, a.n - 1)];                                     \
        nd0[u] = (live && t <= T1) ? v : 0xFFFFFFFFu; /* (clamped loads, masked: no conditional register writes) */       \
    }
    {
        const uint64_t r = min(row, a.n - 1);
        rs = a.stats[r];
        if (METRIC == AH_DOT_PRODUCT) row_extra = a.headers[2 * r];
    }
#if AH_NARROW_PRELOAD
    AH_NARROW_LOAD_ND0()
#endif
    // A ring: kSets register sets of one k-block each (4 x 16 B per lane: [2 i + s]), kSets - 1 k-blocks in flight per wave
    constexpr uint32_t kSets = NT == 2 ? AH_NARROW_SETS : 3u, kAhead = kSets - 1u;
    uint4 ar[kSets][4];
#define AH_NARROW_ISSUE(KB, SET)                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) ar[SET][i_] =                                                        \
        STREAM ? ld_stream_u4(a_src[i_ >> 1] + (uint64_t)(KB) * 8u + 4u * (i_ & 1))                                       \
               : a_src[i_ >> 1][(uint64_t)(KB) * 8u + 4u * (i_ & 1)]
    // one k-block.  STEADY: k-block kb + SET + kAhead exists (no conditional issue: the compiler's wait counts stay exact).
    // Fragment of column n = 16 j + m at k-step s: chunk 4 s + kg of the column's 128-byte line, at slot chunk ^ ((n >> 1) & 7)
    // (the loader's source swizzle): the 16-lane groups of a ds_read_b128 then cover all 64 banks
#define AH_NARROW_STEP(SET, STEADY)                                                                                       \
    if (STEADY || kb + (SET) < nk) {                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my reads of stage kb + SET - 1 have returned */             \
        __builtin_amdgcn_s_barrier();                                                                                     \
        asm volatile("" ::: "memory");                                                                                    \
        if (STEADY || kb + (SET) + kAhead < nk) AH_NARROW_ISSUE(kb + (SET) + kAhead, ((SET) + kAhead) % kSets);           \
        __builtin_amdgcn_sched_barrier(0); /* the prefetch is issued HERE, not wherever the scheduler finds a gap */       \
        const uint8_t *sb_ = smem + ((kb + (SET)) % SH::kStages) * SH::kStage + m * 128u;                                 \
        _Pragma("unroll") for (uint32_t s_ = 0; s_ < 2; s_++) {                                                           \
            const f16x8_t a0_ = __builtin_bit_cast(f16x8_t, ar[SET][s_]);                                                 \
            const f16x8_t a1_ = __builtin_bit_cast(f16x8_t, ar[SET][2 + s_]);                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < 2 * NT; j_++) {                                                       \
                /* column 16 j + m: (n >> 1) & 7 = ((m >> 1) + 8 j) & 7 = (m >> 1) & 7 */                                  \
                const f16x8_t b_ = *reinterpret_cast<const f16x8_t *>(sb_ + j_ * 2048 + (((4u * s_ + kg) ^ ((m >> 1) & 7u)) << 4)); \
                acc[0][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0_, b_, acc[0][j_], 0, 0, 0);                        \
                acc[1][j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1_, b_, acc[1][j_], 0, 0, 0);                        \
            }                                                                                                             \
        }                                                                                                                 \
    }
    // (unconditional prologue: a short row fetches its last k-block more than once, nobody reads the copies)
#pragma unroll
    for (uint32_t p = 0; p < kAhead; p++) AH_NARROW_ISSUE(min(p, nk - 1u), p);
    uint32_t kb = 0;
    for (; kb + 2u * kSets - 1u <= nk; kb += kSets) {
        AH_NARROW_STEP(0, true)
        AH_NARROW_STEP(1, true)
        AH_NARROW_STEP(2, true)
        if (kSets == 4) AH_NARROW_STEP(kSets - 1, true)
    }
    for (; kb < nk; kb += kSets) {
        AH_NARROW_STEP(0, false)
        AH_NARROW_STEP(1, false)
        AH_NARROW_STEP(2, false)
        if (kSets == 4) AH_NARROW_STEP(kSets - 1, false)
    }
#undef AH_NARROW_STEP
#undef AH_NARROW_ISSUE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the ring is dead (every wave's fragment reads have returned): it becomes the waves'
    asm volatile("" ::: "memory");  // private epilogue tiles
    // Epilogue, 32 columns at a time, private to the wave.  D layout of the 16x16 MFMA: lane -> column (lane & 15) of the B
    // operand (the normals), register e -> row 4 (lane >> 4) + e of the A operand (the wave's rows): the four registers
    // are four consecutive rows = one 16-byte store of S[column][row].
#if !AH_NARROW_PRELOAD
    AH_NARROW_LOAD_ND0()
#endif
#undef AH_NARROW_LOAD_ND0
    float *S = reinterpret_cast<float *>(smem + wave * SH::kEpiWave);
    const NormalStats *nst_all = reinterpret_cast<const NormalStats *>(smem + SH::kRing);
    // 0xFFFFFFFF (leaf row / no such tree) and nodes of other rounds fall outside
#define AH_NARROW_DECIDE(T, NODE)                                                                                         \
    do {                                                                                                                  \
        const uint32_t node_ = (NODE), c_ = node_ - c_lo;                                                                 \
        if (node_ != 0xFFFFFFFFu && c_ < 32u) {                                                                           \
            uint32_t side_;                                                                                               \
            const bool decided_ = screen_decides<METRIC>(S[c_ * 36u + er], rs, nst[c_], row_extra, a.gamma_s, a.gamma_r, side_); \
            a.side_bytes[(uint64_t)(T) * a.n + row] =                                                                     \
                (uint8_t)(decided_ ? (a.verify ? (kSideVerify | side_) : side_) : kSideUndecided);                 \
        }                                                                                                                 \
    } while (0)
#pragma unroll
    for (int jn = 0; jn < NT; jn++) {
        const uint32_t c_lo = c0 + (uint32_t)jn * 32u;
        if (c_lo >= a.n_cols) break;  // block-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the previous round's readers are done (one wave: program order)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const f32x4_t v = acc[i][2 * jn + jj];
                *reinterpret_cast<float4 *>(S + (16u * (uint32_t)jj + m) * 36u + 16u * (uint32_t)i + 4u * kg) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        const NormalStats *nst = nst_all + 32 * jn;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) AH_NARROW_DECIDE(T0 + eh + 2u * u, nd0[u]);
        if (T1 >= T0 + 16u) {  // more than 16 trees in the tile (block-uniform): the rest on demand, eight loads in flight
            // nodes are ordered by tree: the round's columns cover the trees [t_lo, t_hi]
            const uint32_t t_lo = max(a.nodes[c_lo].tree, T0 + 16u), t_hi = a.nodes[min(c_lo + 32u, a.n_cols) - 1u].tree;
            for (uint32_t t = t_lo + ((t_lo ^ T0 ^ eh) & 1u); t <= t_hi; t += 16u) {  // half h keeps the trees of its parity
                uint32_t nd[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++)
                    nd[u] = (live && t + 2u * u <= t_hi) ? a.node_of[(uint64_t)(t + 2u * u) * a.n + row] : 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) AH_NARROW_DECIDE(t + 2u * u, nd[u]);
            }
        }
    }
#undef AH_NARROW_DECIDE
}

// The pairs the dense screen left open (side byte 2; or every pair under AH_SCREEN_VERIFY), recomputed in the reference
// arithmetic: a wave scans a window of 1024 rows of one tree's side bytes (16 per lane), lists the marked ones in LDS and
// hands them to its octets, eight pairs at a time — the arithmetic of k_forest_margin_rows (rows_exact_margin).  A block
// takes one row block (1024 rows) of four trees, and all the tree groups of a row block are handed to ONE XCD (block b runs
// on XCD b & 7; the grid is a multiple of 8) back to back: an f32 row that is marked in several trees — on average every
// row is, once, at 100 trees — is then found in that XCD's L2 after its first read (1024 rows = 3 MB).
typedef uint32_t u32x4_a4_t __attribute__((ext_vector_type(4), aligned(4)));
// unit (a block's turn) + wave -> (row block of 1024 rows, tree); shared with k_exact_coverage (ah_debug_launch_coverage)
__device__ __forceinline__ bool exact_unit_map(uint64_t unit, uint32_t wave, uint64_t blocks_per_tree, uint32_t tree_groups,
                                               uint32_t n_trees, uint64_t &rb, uint64_t &t) {
    const uint64_t slot = unit >> 3;
    rb = (slot / tree_groups) * 8 + (unit & 7u);
    t = (slot % tree_groups) * 4 + wave;
    return rb < blocks_per_tree && t < n_trees;
}
__device__ __forceinline__ uint64_t exact_units(uint64_t n_rows, uint32_t n_trees, uint64_t &blocks_per_tree, uint32_t &tree_groups) {
    blocks_per_tree = (n_rows + 1023) >> 10;
    tree_groups = (n_trees + 3u) >> 2;
    return ((blocks_per_tree + 7) >> 3) * 8 * tree_groups;  // (row block, tree group) pairs, padded per XCD
}
__global__ __launch_bounds__(256) void k_exact_coverage(uint64_t n_rows, uint32_t n_trees, uint32_t *__restrict__ counts) {
    uint64_t blocks_per_tree;
    uint32_t tree_groups;
    const uint64_t n_units = exact_units(n_rows, n_trees, blocks_per_tree, tree_groups);
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        uint64_t rb, t;
        if (!exact_unit_map(unit, threadIdx.x >> 6, blocks_per_tree, tree_groups, n_trees, rb, t)) continue;
        if ((threadIdx.x & 63u) == 0) atomicAdd(&counts[t * blocks_per_tree + rb], 1u);
    }
}
template <int METRIC, bool WIDE = false>
__global__ __launch_bounds__(256) void k_forest_exact_pairs(DataView dv, const uint32_t *__restrict__ node_of,
                                                            uint8_t *__restrict__ side_bytes, uint32_t n_trees,
                                                            const uint8_t *__restrict__ normals, uint64_t nstride,
                                                            uint64_t hdr_off, ScreenCounters *__restrict__ counters,
                                                            const AbortFlags abort_flag) {
    __shared__ uint32_t s_list[4][1024];
    __shared__ uint32_t s_fb, s_bad;
    if (abort_requested(abort_flag)) return;
    if (threadIdx.x == 0) s_fb = s_bad = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, j = lane & 7u, o = lane >> 3;
    uint64_t blocks_per_tree;
    uint32_t tree_groups;
    const uint64_t n_units = exact_units(dv.n, n_trees, blocks_per_tree, tree_groups);
    const bool aligned4 = (dv.n & 3ull) == 0;  // every tree's bytes then start on a 4-byte boundary
    uint32_t fallbacks = 0, bad = 0;
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        uint64_t rb, t;
        if (!exact_unit_map(unit, wave, blocks_per_tree, tree_groups, n_trees, rb, t)) continue;  // wave-uniform
        const uint64_t r0 = (rb << 10) + lane * 16u;  // first row of this lane's 16 bytes
        const uint64_t base = t * dv.n;                // the tree's side bytes / node indices
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (r0 < dv.n) {  // the buffer is padded beyond the last tree, so a whole 16-byte load is always in bounds
            if (aligned4) {
                const u32x4_a4_t v = *reinterpret_cast<const u32x4_a4_t *>(side_bytes + base + r0);
                w[0] = v.x;
                w[1] = v.y;
                w[2] = v.z;
                w[3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) w[e >> 2] |= (uint32_t)side_bytes[base + r0 + e] << (8 * (e & 3));
            }
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; e++)
            cnt += (((w[e >> 2] >> (8 * (e & 3))) & 0xFEu) && r0 + e < dv.n) ? 1u : 0u;  // bytes past N belong to the next tree
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int)lane >= off) incl += up;
        }
        const uint32_t n_marked = __shfl(incl, 63);
        if (n_marked == 0) continue;  // wave-uniform
        uint32_t pos = incl - cnt;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint32_t b = (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            if ((b & 0xFEu) && r0 + e < dv.n) s_list[wave][pos++] = (lane * 16u + (uint32_t)e) | (b << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t e0 = 0; e0 < n_marked; e0 += 8) {
            const uint32_t idx = e0 + o;
            const bool active = idx < n_marked;
            const uint32_t ent = active ? s_list[wave][idx] : 0u;
            const uint64_t r = (rb << 10) + (ent & 0xFFFFu);
            const uint32_t code = ent >> 16;
            const uint32_t node = active ? node_of[base + r] : 0xFFFFFFFFu;
            if (node != 0xFFFFFFFFu) {  // octet-uniform
                const uint32_t exact = side_of_margin(WIDE ? rows_exact_margin_wide<METRIC>(dv, r, normals + (uint64_t)node * nstride, hdr_off, j)
                                                           : rows_exact_margin<METRIC>(dv, r, normals + (uint64_t)node * nstride, hdr_off, j));
                if (j == 0) {
                    side_bytes[base + r] = (uint8_t)exact;
                    if (code == kSideUndecided) fallbacks++;
                    else if ((code & 1u) != exact) bad++;
                }
            } else if (active && j == 0) {
                side_bytes[base + r] = 0;  // a stale mark on a row that is a leaf in this tree: nobody reads it
            }
        }
        __builtin_amdgcn_wave_barrier();  // the list is re-used by the next window
    }
    if (fallbacks) atomicAdd(&s_fb, fallbacks);
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_fb) atomicAdd(&counters->fallbacks, (unsigned long long)s_fb);
        if (s_bad) atomicAdd(&counters->violations, (unsigned long long)s_bad);
    }
}

}  // namespace ah
