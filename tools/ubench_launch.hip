// Diagnostic micro-benchmark (not part of the product): the cost of one DEPENDENT kernel launch, three ways --
// a captured hipGraph chain, plain in-order stream launches from a tight host loop, and hipExtLaunchKernelGGL --
// for an empty kernel and for one with k_p2_a's shape (52 workgroups of 512 threads, 1.6 MB written).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 99999) p[0] = 1; }
__global__ __launch_bounds__(512) void k_write(float* slab, int per_wg, float v) {
    float4* dst = reinterpret_cast<float4*>(slab + (size_t)blockIdx.x * per_wg);
    for (int e = threadIdx.x; e < per_wg / 4; e += blockDim.x) dst[e] = float4{v, v, v, v};
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* slab; CK(hipMalloc(&slab, 52 * 8192 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int real = 0; real < 2; ++real) {
        auto launch = [&](int i) {
            if (real) hipLaunchKernelGGL(k_write, dim3(52), dim3(512), 0, s, slab, 8192, (float)i);
            else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, slab);
        };
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s kernel, graph chain        : %.3f us per dependent launch\n", real ? "1.6 MB-writing" : "empty", ms * 1e3 / N);
        // plain stream launches
        for (int i = 0; i < 200; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipEventRecord(e1, s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s kernel, stream launches    : %.3f us per dependent launch on the device, %.3f us of host time per launch call\n",
               real ? "1.6 MB-writing" : "empty", ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
}
