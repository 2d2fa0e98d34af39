// Micro-benchmark: issue rate of the candidate inner-loop instructions of the binary16 screen on gfx950.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned *in, float *out) {
    unsigned a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    float c[8];
    for (int i = 0; i < 8; i++) c[i] = (float)i;
    f2 p[4];
    for (int i = 0; i < 4; i++) p[i] = f2{(float)i, (float)i};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) c[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a0 + i), __builtin_bit_cast(h2, b0), c[i], false);
            if (MODE == 1) c[i] = __builtin_fmaf(__uint_as_float(a0 + i), __uint_as_float(b0), c[i]);
            if (MODE == 2) {  // fma_mix: f16 x f16 + f32
                c[i] = __builtin_fmaf((float)__builtin_bit_cast(h2, a0 + i).x, (float)__builtin_bit_cast(h2, b0).x, c[i]);
                c[i] = __builtin_fmaf((float)__builtin_bit_cast(h2, a0 + i).y, (float)__builtin_bit_cast(h2, b0).y, c[i]);
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] = __builtin_elementwise_fma(f2{__uint_as_float(a0 + i), __uint_as_float(a1)}, f2{__uint_as_float(b0), __uint_as_float(b1)}, p[i]);
        }
        a0 += 3;
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    unsigned *in; float *out;
    hipMalloc(&in, 4096); hipMemset(in, 0x3c, 4096);
    const int blocks = 256 * 8;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"v_dot2c_f32_f16 (8 per iter)", "v_fma_f32 (8 per iter)", "fma_mix f16->f32 (16 per iter)", "v_pk_fma_f32 (4 per iter)"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, in, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per = mode == 2 ? 16 : mode == 3 ? 4 : 8;
            const double winstr = (double)blocks * 4 * ITERS * per;  // wave-instructions
            if (rep) printf("%-34s %.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", names[mode], ms,
                            ms * 1e-3 * 2.4e9 * 1024 / winstr);
        }
    }
    return 0;
}
