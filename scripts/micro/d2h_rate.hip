// Device -> pinned host copy rates on this box: one big copy, 32 MiB pieces back to back on one stream, and two streams.
// (What bounds the build's tail: 4 GB of item ids after the last level.)  hipcc --offload-arch=gfx950 -O2 d2h_rate.hip -o d2h_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t total = 2ull << 30, piece = 32ull << 20;
    void *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, total));
    CK(hipMemset(d, 1, total));
    CK(hipHostMalloc(&h, total, hipHostMallocDefault));
    memset(h, 0, total);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        double t1 = now();
        for (size_t off = 0; off < total; off += piece) CK(hipMemcpyAsync((char *)h + off, (char *)d + off, piece, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        double t2 = now();
        int k = 0;
        for (size_t off = 0; off < total; off += piece, k++)
            CK(hipMemcpyAsync((char *)h + off, (char *)d + off, piece, hipMemcpyDeviceToHost, (k & 1) ? s1 : s0));
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        double t3 = now();
        // host -> device for comparison
        CK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        double t4 = now();
        printf("D2H one copy %.1f GB/s | 32 MiB pieces, one stream %.1f GB/s | two streams %.1f GB/s | H2D one copy %.1f GB/s\n",
               total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9, total / (t3 - t2) / 1e9, total / (t4 - t3) / 1e9);
    }
    return 0;
}
