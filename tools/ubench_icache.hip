// Diagnostic: does straight-line code cost more than the same instruction count in a loop (cold instruction cache per
// launch)?  Kernel = N independent v_fma per lane, executed once (unrolled) or as a 16-instruction body looped.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, bool UNROLL>
__global__ void k(float* out, float a) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (UNROLL) {
#pragma unroll
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], a, 1.0f + i);
    } else {
#pragma unroll 2
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], a, 1.0f + i);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ((unsigned long long*)out)[4096 + blockIdx.x] = t1 - t0;
}
template <int N, bool U> void run(const char* name, float* d) {
    unsigned long long h[64];
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k<N, U>), 32, 64, 0, 0, d, 1.0001f); hipLaunchKernelGGL((k<64, false>), 32, 64, 0, 0, d, 1.1f); }
    hipLaunchKernelGGL((k<N, U>), 32, 64, 0, 0, d, 1.0001f);
    hipDeviceSynchronize();
    hipMemcpy(h, (unsigned long long*)d + 4096, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 32; ++i) m += h[i];
    printf("%-28s N=%5d  in-kernel %.2f us  (%.1f cycles per instruction at 2.4 GHz)\n", name, N, m / 32 / 100.0, m / 32 / 100.0 * 2400 / N);
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    run<512, true>("unrolled", d); run<512, false>("looped", d);
    run<2048, true>("unrolled", d); run<2048, false>("looped", d);
    run<8192, true>("unrolled", d); run<8192, false>("looped", d);
    return 0;
}
