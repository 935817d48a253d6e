#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
__device__ inline float wave_sum_lane0(float v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    unsigned b = __float_as_uint(v);
    u2 r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    b = __float_as_uint(v);
    r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));
    return v;
}
__global__ void k(const float* in, float* out, float* out2) {
    float v = in[threadIdx.x];
    float a = wave_sum_lane0(v);
    float b = v;
    for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off, 64);
    if (threadIdx.x == 0) { out[0] = a; out2[0] = b; }
}
int main() {
    float h[64], *d, *o, *o2; unsigned seed = 1;
    for (int i = 0; i < 64; ++i) { seed = seed * 1664525u + 1013904223u; h[i] = (float)(seed >> 8) / 1e5f - 50.f; }
    hipMalloc(&d, 256); hipMalloc(&o, 4); hipMalloc(&o2, 4);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, o2);
    float a, b; hipMemcpy(&a, o, 4, hipMemcpyDeviceToHost); hipMemcpy(&b, o2, 4, hipMemcpyDeviceToHost);
    printf("dpp %.9g shfl %.9g same_bits %d\n", a, b, memcmp(&a, &b, 4) == 0);
    return memcmp(&a, &b, 4) != 0;
}
