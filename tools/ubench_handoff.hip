// Diagnostic micro-benchmark (not part of the product): what does it cost to hand ~1.5 MB of fp32 partials from one
// launch (W writer workgroups) to the next (R reader workgroups)?  Reader variants differ in loads in flight and width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 99999) p[0] = 1; }
__global__ void k_writer(float* slab, int per_wg) {           // each WG writes per_wg floats, coalesced
    float* dst = slab + (size_t)blockIdx.x * per_wg;
    for (int e = threadIdx.x; e < per_wg; e += blockDim.x) dst[e] = (float)e;
}
// reader: WG b sums element (b*chunk + tid) over G slabs (stride gs) -- the k_pipe_b access pattern; NL loads in flight
template <int NL>
__global__ void k_reader(const float* __restrict__ slab, int G, size_t gs, int chunk, float* out, unsigned long long* st) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const float* p = slab + (size_t)blockIdx.x * chunk + threadIdx.x;
    float v = 0;
    for (int g0 = 0; g0 < G; g0 += NL) {
        float t[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) t[q] = p[(size_t)(g0 + q < G ? g0 + q : G - 1) * gs];
#pragma unroll
        for (int q = 0; q < NL; ++q) v += (g0 + q < G) ? t[q] : 0.f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = t1; }
}
// contiguous reader: WG b reads its own contiguous region of n float4 per thread
__global__ void k_reader_contig(const float4* __restrict__ slab, int n4_per_thread, float* out, unsigned long long* st) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const float4* p = slab + (size_t)blockIdx.x * blockDim.x * n4_per_thread + threadIdx.x;
    float v = 0;
    float4 t[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) t[q] = p[(size_t)(q < n4_per_thread ? q : 0) * blockDim.x];
#pragma unroll
    for (int q = 0; q < 12; ++q) v += (q < n4_per_thread) ? t[q].x + t[q].y + t[q].z + t[q].w : 0.f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = t1; }
}

int main() {
    const int G = 49, B = 256, Mp = 32;
    const size_t gs = (size_t)B * Mp, total = gs * G;
    float *slab, *out; unsigned long long* st;
    CK(hipMalloc(&slab, total * 4)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&st, 8 * 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, int nwg) -> int {
        const int reps = 200;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * 4096);
        CK(hipMemcpy(h.data(), st, 8 * 4096, hipMemcpyDeviceToHost));
        unsigned long long mn = ~0ull, mx = 0; double avg = 0;
        for (int i = 0; i < nwg; ++i) { mn = h[2*i] < mn ? h[2*i] : mn; mx = h[2*i+1] > mx ? h[2*i+1] : mx; avg += (double)(h[2*i+1] - h[2*i]); }
        printf("%-58s %7.2f us per iteration | in-kernel: span %5.2f us, mean WG %5.2f us\n", name, ms * 1000 / reps, nwg ? (mx - mn) / 100.0 : 0.0, nwg ? avg / nwg / 100.0 : 0.0);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        return 0;
    };
    timeit("empty kernel (1 WG)", [&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, slab); }, 0);
    timeit("empty kernel (64 WG x 512)", [&] { hipLaunchKernelGGL(k_empty, 64, 512, 0, s, slab); }, 0);
    timeit("writer only (49 WG x 30 KB)", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); }, 0);
    timeit("writer + reader 32 WG, 8 samples, 8 loads in flight", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader<8>, 32, 256, 0, s, slab, G, gs, 256, out, st); }, 32);
    timeit("writer + reader 32 WG, 64 loads in flight", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader<64>, 32, 256, 0, s, slab, G, gs, 256, out, st); }, 32);
    timeit("writer + reader 64 WG x 128 thr, 64 in flight", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader<64>, 64, 128, 0, s, slab, G, gs, 128, out, st); }, 64);
    timeit("writer + reader 128 WG x 64 thr, 64 in flight", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader<64>, 128, 64, 0, s, slab, G, gs, 64, out, st); }, 128);
    timeit("reader only 32 WG (data resident, no writer)", [&] { hipLaunchKernelGGL(k_reader<64>, 32, 256, 0, s, slab, G, gs, 256, out, st); }, 32);
    timeit("writer + contiguous float4 reader 32 WG (48 KB each)", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader_contig, 32, 256, 0, s, (const float4*)slab, 12, out, st); }, 32);
    timeit("writer + contiguous float4 reader 128 WG (12 KB each)", [&] { hipLaunchKernelGGL(k_writer, G, 512, 0, s, slab, (int)gs); hipLaunchKernelGGL(k_reader_contig, 128, 256, 0, s, (const float4*)slab, 3, out, st); }, 128);
    return 0;
}
