// What limits the inner loop of k_conv3x3_halo_f32?  The same phase (3 taps x 32 channels: one ds_read_b128 of A per four k-steps,
// one ds_read_b32 of B per MFMA and column block) in a loop with nothing else -- then with the two barriers, then with the LDS
// stores of the staging -- at 1..4 workgroups per CU.  Prints achieved TFLOP/s and the shader clock during the run.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_mfma_lds tools/ubench_mfma_lds.hip && /tmp/ubench_mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NT, int MODE>   // MODE 0: loop only; 1: + two barriers per phase; 2: + barriers + the staging's LDS stores
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int phases, int pad_lds) {
    constexpr int BN = 32 * NT, LDC = 36, HWD = 18;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;
    float* Bs = smem + 180 * LDC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int e = tid; e < 180 * LDC + 96 * BN; e += 256) smem[e] = 1.0f / (1 + (e & 7));
    __syncthreads();
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int py = 2 * wave + (r >> 4), pxb = r & 15;
    const float* arow = Hs + (py * HWD + pxb) * LDC + 4 * h;
    const float* bt[2];
    if (NT == 2) { bt[0] = Bs + 256 * h + r + 32 * h; bt[1] = Bs + 256 * h + r + 32 * (1 - h); }
    else { bt[0] = Bs + 32 * h + r; bt[1] = bt[0]; }
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    f32x4 sv[6];
    for (int q = 0; q < 6; ++q) sv[q] = f32x4{1.f * tid, 2.f, 3.f, 4.f};
    int kh = 0;
#pragma unroll 1
    for (int ph = 0; ph < phases; ++ph) {
        if (MODE >= 1) __syncthreads();
        if (MODE >= 2) {
#pragma unroll
            for (int q = 0; q < 3 * NT; ++q) *reinterpret_cast<f32x4*>(&Bs[((tid + 256 * q) * 4) % (96 * BN)]) = sv[q];
            if (kh == 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) { const int e = tid + 256 * q; if (e < 1440) *reinterpret_cast<f32x4*>(&Hs[(e >> 3) * LDC + (e & 7) * 4]) = sv[q]; }
            }
        }
        if (MODE >= 1) __syncthreads();
        const float* ak = arow + kh * HWD * LDC;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ak + kw * LDC + 8 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float b = NT == 2 ? bt[t][(kw * 32 + 8 * j + i) * 64] : bt[0][(kw * 32 + 8 * j + 4 * (i & 1) + (i & 2)) * 32];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b, acc[t], 0, 0, 0);
                    }
            }
        kh = kh == 2 ? 0 : kh + 1;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.f) out[0] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int NT, int MODE> void run(float* d, unsigned long long* dc, int per_cu) {
    const int BN = 32 * NT, phases = 600;
    const size_t lds = (180 * 36 + 96 * BN) * 4;
    const int blocks = 256 * per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<NT, MODE><<<blocks, 256, lds>>>(d, dc, 10, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NT, MODE><<<blocks, 256, lds>>>(d, dc, phases, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * phases * 48.0 * NT * 4096.0;
    printf("NT %d mode %d  %d WG/CU: %7.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)  shader clock %.0f MHz\n", NT, MODE, per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100,
           hc[0] / (hc[1] / 100.0));
}
int main() {
    float* d; hipMalloc(&d, 4);
    unsigned long long* dc; hipMalloc(&dc, 16);
    for (int per_cu = 1; per_cu <= 3; ++per_cu) { run<2, 0>(d, dc, per_cu); run<2, 1>(d, dc, per_cu); run<2, 2>(d, dc, per_cu); }
    for (int per_cu = 1; per_cu <= 4; ++per_cu) { run<1, 0>(d, dc, per_cu); run<1, 1>(d, dc, per_cu); run<1, 2>(d, dc, per_cu); }
    return 0;
}
