// Micro-benchmark: what bounds the node-major margin pass of the deep levels?  Octets gather whole 1536-byte rows (binary16
// shadow of a 768-d row) from a 10M-row table (15.4 GB), as k_forest_screen_node does for the items of a node.
//   window = 10M rows : every block draws rows from the whole table (what the deep levels do today: a node's items are
//                       spread over the whole dataset, and ~2000 blocks are in flight)
//   window = W rows   : "band order" — the blocks in flight at any time all draw from the same W consecutive rows, each
//                       row ~100 times (100 trees), then the band moves on.  Small bands fit the Infinity Cache (256 MB),
//                       medium ones only help the TLBs / DRAM pages.
// Prints TB/s per window size: if they differ a lot, ordering the (node, tile) work list by row band pays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const uint4 *table, uint64_t n_rows, uint64_t window, uint32_t blocks_per_band,
                                         uint32_t iters, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, o = lane >> 3, j = lane & 7u;
    const uint64_t band = blockIdx.x / blocks_per_band;
    const uint64_t n_bands = (n_rows + window - 1) / window;
    const uint64_t base = (band % n_bands) * window;
    const uint64_t span = base + window <= n_rows ? window : n_rows - base;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t h = ((uint64_t)wave * 0x9E3779B97F4A7C15ull) ^ (((uint64_t)it * 8 + o) * 0xC2B2AE3D27D4EB4Full + 12345u);
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const uint4 *rec = table + (base + h % span) * 96 + j;
        u32x4_t v[12];
#pragma unroll
        for (int s = 0; s < 12; s++) v[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(rec + s * 8));
#pragma unroll
        for (int s = 0; s < 12; s++) acc ^= v[s].x + v[s].y + v[s].z + v[s].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void fill(uint4 *t, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main() {
    const uint64_t n_rows = 10000000;
    uint4 *table; uint32_t *out;
    if (hipMalloc(&table, n_rows * 1536) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, table, n_rows * 96);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 64;  // rows per octet
    // total rows gathered = blocks * 32 octets * iters = 4 passes over the table's worth of bytes
    const uint32_t blocks = (uint32_t)(4 * n_rows / (32 * iters));
    for (uint64_t window : {10000000ull, 2500000ull, 1000000ull, 400000ull, 128000ull, 32000ull}) {
        const uint64_t n_bands = (n_rows + window - 1) / window;
        const uint32_t bpb = (uint32_t)((blocks + n_bands - 1) / n_bands);
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, n_rows, window, bpb, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double bytes = (double)blocks * 32 * iters * 1536;
        printf("window %9llu rows (%8.1f MB): %.3f ms  %.2f TB/s\n", (unsigned long long)window, window * 1536 / 1e6, best,
               bytes / best / 1e9);
    }
    return 0;
}
