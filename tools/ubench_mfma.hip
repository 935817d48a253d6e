// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip: nothing but MFMAs (two independent accumulator chains per wave), W waves per
// SIMD.  The ceiling every fp32-MFMA kernel of Track X is priced against (guide: 157.3 TFLOP/s at 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_mfma tools/ubench_mfma.hip && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = 256 * wps, iters = 4000;
        k<<<blocks, 256>>>(d, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<<<blocks, 256>>>(d, iters, 1.f, 2.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
            printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (implied clock at 64 FLOP/clk/SIMD: %.2f GHz)\n", wps, ms, flops / ms / 1e9, flops / ms / 1e6 / (1024 * 64.0) / 1e3 * 1e3 / 1e3);
        }
    }
    return 0;
}
