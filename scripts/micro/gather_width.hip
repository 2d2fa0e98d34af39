// Micro-benchmark: L2-resident gather of 1536-byte records by a wave, as the row-major margin pass does for its normals.
//   A: 8 lanes per record, 16 B per lane  -> one 128-byte line per record and instruction, 8 records per instruction
//   B: 16 lanes per record                -> two adjacent lines (256 B) per record and instruction, 4 records per instruction
//   C: 32 lanes per record                -> 512 B per record and instruction, 2 records per instruction
// Same bytes in every variant; which one moves more bytes per second tells whether the vector-memory pipeline is bound by
// requests in flight (then wider contiguous accesses per record help) or by bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REC16 = 96;  // uint4 per record (1536 B)
template <int LANES>
__global__ __launch_bounds__(256) void k(const uint4 *table, uint32_t n_rec, uint32_t iters, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, g = lane / LANES, j = lane % LANES;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t acc = 0;
    constexpr int GROUPS = 64 / LANES, STEPS = REC16 / LANES;
    for (uint32_t it = 0; it < iters * (8 / GROUPS); it++) {
        uint32_t h = (wave * 2654435761u) ^ ((it * GROUPS + g) * 40503u + 12345u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint4 *rec = table + (size_t)(h % n_rec) * REC16 + j;
        uint4 v[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; s++) v[s] = rec[s * LANES];
#pragma unroll
        for (int s = 0; s < STEPS; s++) acc ^= v[s].x + v[s].y + v[s].z + v[s].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    for (uint32_t n_rec : {2048u, 8192u, 65536u}) {  // 3 MB (fits the L2s), 12.6 MB, 100 MB
        uint4 *table; uint32_t *out;
        hipMalloc(&table, (size_t)n_rec * 1536); hipMemset(table, 1, (size_t)n_rec * 1536); hipMalloc(&out, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256 * 20; const uint32_t iters = 400;
        for (int variant = 0; variant < 3; variant++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
                if (variant == 1) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
                if (variant == 2) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double bytes = (double)blocks * 4 * iters * 8 * 1536;
            printf("table %6.1f MB, %2d lanes per record: %.3f ms  %.2f TB/s\n", n_rec * 1536 / 1e6, 8 << variant, best, bytes / best / 1e9);
        }
        hipFree(table); hipFree(out);
    }
    return 0;
}
