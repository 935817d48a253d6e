// Diagnostic micro-benchmark (not part of the product): is k_features_cpcp's output pattern the limit?  One wave per
// image writes its 784 f32 features (a) exactly as the kernel does -- 16 dword stores per lane, runs of 7 contiguous floats
// per 8-lane group -- and (b) as 13 contiguous dwordx4 stores per image from a linear buffer; no reads, no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_as_kernel(float* out, int n_img) {
    const int lane = threadIdx.x;
    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
        float* dst = out + (size_t)img * 784;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = lane + 64 * k, cg = t >> 3, py = t & 7;
            if (cg < 28 && py < 7) {
                const int i = cg / 7, px = cg - i * 7, q = px * 7 + py;
                const float v = (float)(t + img);
                dst[(4 + 3 * i + 0) * 49 + q] = v;
                dst[(4 + 3 * i + 1) * 49 + q] = v + 1.f;
                dst[(4 + 3 * i + 2) * 49 + q] = v + 2.f;
                dst[i * 49 + q] = v + 3.f;
            }
        }
    }
}
__global__ __launch_bounds__(64) void k_linear(float* out, int n_img) {
    const int lane = threadIdx.x;
    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
        float4* dst = reinterpret_cast<float4*>(out + (size_t)img * 784);
        const float v = (float)(lane + img);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = lane + 64 * k;
            if (w < 196) dst[w] = float4{v, v + 1.f, v + 2.f, v + 3.f};
        }
    }
}
int main() {
    const int n = 131072;
    float* out; CK(hipMalloc(&out, (size_t)n * 784 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        for (int grid : {3584, 5120, 8192}) {
            for (int r = 0; r < 3; ++r) { if (which) k_linear<<<grid, 64>>>(out, n); else k_as_kernel<<<grid, 64>>>(out, n); }
            CK(hipEventRecord(e0));
            for (int r = 0; r < 10; ++r) { if (which) k_linear<<<grid, 64>>>(out, n); else k_as_kernel<<<grid, 64>>>(out, n); }
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 10;
            printf("%s grid %5d: %7.1f us for %d images = %.0f GB/s written\n", which ? "contiguous dwordx4 stores" : "the kernel's 16 dword stores", grid, us, n,
                   (double)n * 3136 / us / 1e3);
        }
    }
    return 0;
}
