// Micro-benchmark: the screened row-major margin pass rebuilt piece by piece, to see which ingredient costs the
// vector-memory throughput that a pure gather reaches (scripts/micro/gather_width.hip: 33 TB/s from L2).
//   V0  gather only: per row-octet, TC x 12 loads of 16 B from random 1536-byte records of an L2-resident table
//   V1  + the dot2c arithmetic (4 per load, one accumulator per tree)
//   V2  + the row stream: 12 non-temporal 16-byte loads per row from a 6 GB array (HBM)
//   V3  V2 with ordinary (cached) row loads
//   V4  V2 + epilogue per (row, tree): octet sum, compare, one byte stored
//   V6  V2 with the "rows" taken from an L2-resident array (same instructions, short latency): is it the HBM latency?
//   k_spec: warp-specialised V2 — wave 0 of every block streams the rows of the other three waves into LDS (two
//           buffers), the consumer waves only ever wait for L2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int REC16 = 96;
__device__ __forceinline__ float dot8(uint4 a, uint4 b, float c) {
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.x), __builtin_bit_cast(h2, b.x), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.y), __builtin_bit_cast(h2, b.y), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.z), __builtin_bit_cast(h2, b.z), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.w), __builtin_bit_cast(h2, b.w), c, false);
    return c;
}
template <int V, int TC>
__global__ __launch_bounds__(256) void k(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows, uint8_t *out) {
    const uint32_t j = threadIdx.x & 7u;
    const uint64_t n_octets = ((uint64_t)gridDim.x * blockDim.x) >> 3;
    for (uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; row < n_rows; row += n_octets) {
        uint32_t noff[TC];
        float acc[TC];
#pragma unroll
        for (int t = 0; t < TC; t++) {
            uint32_t h = ((uint32_t)row * 2654435761u) ^ (t * 40503u + 12345u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            noff[t] = (h % n_rec) * REC16 + j;
            acc[t] = 0.f;
        }
        const uint4 *r4 = rows + row * REC16 + j;
        for (int k0 = 0; k0 < 12; k0 += (k0 == 0 ? 8 : 4)) {
            const int ns = k0 == 0 ? 8 : 4;
            uint4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u < ns) {
                    if (V == 2 || V == 4) { u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(r4 + (k0 + u) * 8)); x[u] = make_uint4(v.x, v.y, v.z, v.w); }
                    else if (V == 3) x[u] = r4[(k0 + u) * 8];
                    else if (V == 6) x[u] = (rows + (row & 2047) * REC16 + j)[(k0 + u) * 8];
                    else x[u] = make_uint4(0x3c003c00u + u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (uint32_t)row);
                }
            }
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const uint4 *np = table + noff[t] + k0 * 8;
                uint4 nv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) if (u < ns) nv[u] = np[u * 8];
                float a0 = acc[t], a1 = 0.f;
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    if (u < ns) {
                        if (V == 0) { a0 += __uint_as_float(nv[u].x ^ nv[u].y ^ nv[u].z ^ nv[u].w); a1 += __uint_as_float(nv[u + 1].x ^ nv[u + 1].y ^ nv[u + 1].z ^ nv[u + 1].w); }
                        else { a0 = dot8(nv[u], x[u], a0); a1 = dot8(nv[u + 1], x[u + 1], a1); }
                    }
                }
                acc[t] = a0 + a1;
            }
        }
        if (V == 4) {
#pragma unroll
            for (int t = 0; t < TC; t++) {
                float s = acc[t];
                s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
                const uint4 st = table[noff[t] - j + 95];
                const float e = __uint_as_float(st.x) * 1e-3f + __uint_as_float(st.y) * 2e-3f + 1e-30f;
                if (j == 0) out[(uint64_t)t * n_rows + row] = fabsf(s) > e ? (s < 0.f) : 2;
            }
        } else {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TC; t++) s += acc[t];
            if (s == 123.456f) out[row] = 1;
        }
    }
}
// producer / consumer version: block = 4 waves; wave 0 loads the rows of the 24 octets of waves 1..3 (24 rows per
// iteration) into LDS, double buffered; consumers read their row from LDS and gather + dot as V2.
template <int TC>
__global__ __launch_bounds__(256) void k_spec(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows, uint8_t *out) {
    __shared__ uint4 s_rows[2][24][REC16];  // 2 x 36 KB
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, j = lane & 7u;
    const uint64_t rows_per_block = 24, n_iter = (n_rows / rows_per_block + gridDim.x - 1) / gridDim.x;
    auto produce = [&](uint64_t it, int buf) {
        const uint64_t base = (it * gridDim.x + blockIdx.x) * rows_per_block;
        if (base + rows_per_block > n_rows) return;
        const uint4 *src = rows + base * REC16;  // 24 rows are contiguous: 24 * 96 uint4 = 2304 uint4 = 36 loads of 64 lanes
#pragma unroll 4
        for (int i = 0; i < 36; i++) {
            u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src + i * 64 + lane));
            (&s_rows[buf][0][0])[i * 64 + lane] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    if (wave == 0) produce(0, 0);
    __syncthreads();
    for (uint64_t it = 0; it < n_iter; it++) {
        const int buf = (int)(it & 1);
        if (wave == 0) {
            produce(it + 1, buf ^ 1);
        } else {
            const uint64_t base = (it * gridDim.x + blockIdx.x) * rows_per_block;
            if (base + rows_per_block <= n_rows) {
                const uint32_t o = (wave - 1) * 8 + (lane >> 3);
                const uint64_t row = base + o;
                uint32_t noff[TC];
                float acc[TC];
#pragma unroll
                for (int t = 0; t < TC; t++) {
                    uint32_t h = ((uint32_t)row * 2654435761u) ^ (t * 40503u + 12345u);
                    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                    noff[t] = (h % n_rec) * REC16 + j;
                    acc[t] = 0.f;
                }
                for (int k0 = 0; k0 < 12; k0 += (k0 == 0 ? 8 : 4)) {
                    const int ns = k0 == 0 ? 8 : 4;
                    uint4 x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) if (u < ns) x[u] = s_rows[buf][o][(k0 + u) * 8 + j];
#pragma unroll
                    for (int t = 0; t < TC; t++) {
                        const uint4 *np = table + noff[t] + k0 * 8;
                        uint4 nv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) if (u < ns) nv[u] = np[u * 8];
                        float a0 = acc[t], a1 = 0.f;
#pragma unroll
                        for (int u = 0; u < 8; u += 2) if (u < ns) { a0 = dot8(nv[u], x[u], a0); a1 = dot8(nv[u + 1], x[u + 1], a1); }
                        acc[t] = a0 + a1;
                    }
                }
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TC; t++) s += acc[t];
                if (s == 123.456f) out[row] = 1;
            }
        }
        __syncthreads();
    }
}
template <int TC>
float run_spec(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows, uint8_t *out, unsigned blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_spec<TC>), dim3(blocks), dim3(256), 0, 0, table, n_rec, rows, n_rows, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
}
// chunk-major order: ONE launch covers `groups` passes; blocks are numbered chunk-major, then group, then tile, so the
// passes of a chunk run back to back and find the chunk's rows in the Infinity Cache (256 MB) instead of HBM.
template <int TC, bool NT>
__global__ __launch_bounds__(256) void k_chunked(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows,
                                                 uint32_t chunk_rows, uint32_t groups, uint8_t *out) {
    const uint32_t j = threadIdx.x & 7u;
    const uint32_t tiles = chunk_rows / 32;  // 32 rows per block
    const uint32_t chunk = blockIdx.x / (groups * tiles), rem = blockIdx.x % (groups * tiles);
    const uint32_t group = rem / tiles, tile = rem % tiles;
    const uint64_t row = (uint64_t)chunk * chunk_rows + tile * 32 + (threadIdx.x >> 3);
    if (row >= n_rows) return;
    uint32_t noff[TC];
    float acc[TC];
#pragma unroll
    for (int t = 0; t < TC; t++) {
        uint32_t h = ((uint32_t)row * 2654435761u) ^ ((group * TC + t) * 40503u + 12345u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        noff[t] = (h % n_rec) * REC16 + j;
        acc[t] = 0.f;
    }
    const uint4 *r4 = rows + row * REC16 + j;
    for (int k0 = 0; k0 < 12; k0 += (k0 == 0 ? 8 : 4)) {
        const int ns = k0 == 0 ? 8 : 4;
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u < ns) {
                if (NT) { u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(r4 + (k0 + u) * 8)); x[u] = make_uint4(v.x, v.y, v.z, v.w); }
                else x[u] = r4[(k0 + u) * 8];
            }
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            const uint4 *np = table + noff[t] + k0 * 8;
            uint4 nv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) if (u < ns) nv[u] = np[u * 8];
            float a0 = acc[t], a1 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; u += 2) if (u < ns) { a0 = dot8(nv[u], x[u], a0); a1 = dot8(nv[u + 1], x[u + 1], a1); }
            acc[t] = a0 + a1;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TC; t++) s += acc[t];
    if (s == 123.456f) out[row] = 1;
}
template <int TC, bool NT>
float run_chunked(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows, uint32_t chunk_rows, uint32_t groups, uint8_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    const uint64_t chunks = (n_rows + chunk_rows - 1) / chunk_rows;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chunked<TC, NT>), dim3((unsigned)(chunks * groups * (chunk_rows / 32))), dim3(256), 0, 0, table, n_rec, rows, n_rows, chunk_rows, groups, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
}
template <int V, int TC>
float run(const uint4 *table, uint32_t n_rec, const uint4 *rows, uint64_t n_rows, uint8_t *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, TC>), dim3((unsigned)((n_rows + 31) / 32)), dim3(256), 0, 0, table, n_rec, rows, n_rows, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    const uint64_t n_rows = 4000000;  // 6.1 GB of binary16 rows
    uint4 *table, *rows; uint8_t *out;
    hipMalloc(&rows, n_rows * 1536); hipMemset(rows, 0x3c, n_rows * 1536);
    hipMalloc(&out, n_rows * 16);
    for (uint32_t n_rec : {2048u}) {
        hipMalloc(&table, (size_t)n_rec * 1536); hipMemset(table, 0x3c, (size_t)n_rec * 1536);
        const double gb16 = (double)n_rows * 16 * 1536 / 1e9, gb8 = gb16 / 2;
        float t;
        t = run<0, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V0 gather only        : %.3f ms per 4M rows (%.1f TB/s of normals) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        t = run<1, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V1 + dot2c            : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        t = run<2, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V2 + nt row stream    : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        t = run<3, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V3 cached row stream  : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        t = run<4, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V4 + epilogue         : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        t = run<6, 16>(table, n_rec, rows, n_rows, out); printf("table %.1f MB TC16 V6 rows from L2        : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb16 / t, t * 2.5);
        for (unsigned blocks : {512u, 768u, 1024u, 2048u}) {
            t = run_spec<16>(table, n_rec, rows, n_rows, out, blocks); printf("table %.1f MB TC16 producer/consumer, %u blocks: %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, blocks, t, gb16 / t, t * 2.5);
        }
        for (uint32_t chunk_rows : {16384u, 32768u, 65536u, 131072u, 262144u}) {
            t = run_chunked<16, false>(table, n_rec, rows, n_rows, chunk_rows, 6, out);
            printf("table %.1f MB TC16 chunk-major, 6 groups, chunk %6u rows (%5.0f MB), cached rows: %.3f ms per 6 passes = %.3f per pass -> %.2f ms per pass of 10M\n", n_rec * 1536 / 1e6, chunk_rows, chunk_rows * 1536 / 1e6, t, t / 6, t / 6 * 2.5);
        }
        t = run_chunked<16, true>(table, n_rec, rows, n_rows, 65536, 6, out);
        printf("table %.1f MB TC16 chunk-major, 6 groups, chunk  65536 rows, nt rows: %.3f per pass -> %.2f ms per pass of 10M\n", n_rec * 1536 / 1e6, t / 6, t / 6 * 2.5);
        t = run_chunked<8, false>(table, n_rec, rows, n_rows, 65536, 12, out);
        printf("table %.1f MB TC8  chunk-major, 12 groups, chunk 65536 rows, cached rows: %.3f per pass -> %.2f ms per pass of 10M\n", n_rec * 1536 / 1e6, t / 12, t / 12 * 2.5);
        t = run_spec<8>(table, n_rec, rows, n_rows, out, 768); printf("table %.1f MB TC8  producer/consumer, 768 blocks: %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb8 / t, t * 2.5);
        t = run<0, 8>(table, n_rec, rows, n_rows, out);  printf("table %.1f MB TC8  V0 gather only        : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb8 / t, t * 2.5);
        t = run<2, 8>(table, n_rec, rows, n_rows, out);  printf("table %.1f MB TC8  V2 + dot2c + nt rows  : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb8 / t, t * 2.5);
        t = run<4, 8>(table, n_rec, rows, n_rows, out);  printf("table %.1f MB TC8  V4 + epilogue         : %.3f ms (%.1f TB/s) -> %.2f ms per 10M\n", n_rec * 1536 / 1e6, t, gb8 / t, t * 2.5);
        hipFree(table);
    }
    return 0;
}
