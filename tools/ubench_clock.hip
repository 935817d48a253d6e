// Diagnostic: shader clock (s_memtime ticks per s_memrealtime 100 MHz tick) inside short and long kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_clk(unsigned long long* out, int iters) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[4 * blockIdx.x] = c1 - c0; out[4 * blockIdx.x + 1] = r1 - r0; out[4 * blockIdx.x + 2] = (unsigned long long)v; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 4 * 256);
    unsigned long long h[4 * 256];
    for (int iters : {1000, 10000, 100000, 1000000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_clk, 64, 256, 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("iters %8d: %llu shader cycles in %.2f us -> %.0f MHz (%.1f cycles/iter)\n", iters, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
    // many short kernels back to back, then measure inside the last
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_clk, 64, 256, 0, 0, d, 2000);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("after 2000 short kernels: %llu cycles in %.2f us -> %.0f MHz\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    return 0;
}
