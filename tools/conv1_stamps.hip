// Timeline of k_conv1_fwd_f32 workgroups (default: CIFAR conv1 + fused pool, 512 x 32 x 32 x 3 -> 32); CSV as tools/halo_stamps.hip.
// Stamps per workgroup: entry, then per item { staged, next loads issued, MFMAs issued, epilogue issued } (first 7 items), slot 29 = exit.
//   hipcc -O3 --offload-arch=gfx950 -DRCNX_STAMPS -o /tmp/c1 tools/conv1_stamps.hip && /tmp/c1 out.csv [wg_per_cu]
// MNIST shape (4096 x 28 x 28 x 1 -> 32): add -DC1_CIN=1 -DC1_N=4096 -DC1_HW=28.
#ifndef C1_CIN
#define C1_CIN 3
#define C1_N 512
#define C1_HW 32
#endif
#include "../mercer_research_amd/csrc/convnet_halo.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rcnx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    FILE* f = fopen(argc >= 2 ? argv[1] : "/tmp/conv1_stamps.csv", "w");
    ConvShape s{C1_N, C1_HW, C1_HW, C1_CIN, 32};
    const int tiles_w = (C1_HW + 15) / 16, tiles_h = (C1_HW + 7) / 8, items = tiles_w * tiles_h * s.N;
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_conv1_fwd_f32<C1_CIN, 16, 4>, kThreads, 0));
    if (argc >= 3 && atoi(argv[2]) > 0) per_cu = atoi(argv[2]);
    const int grid = std::min(items, per_cu * 256);
    const size_t nx = (size_t)s.N * s.H * s.W * C1_CIN, ny = (size_t)s.N * s.H * s.W * 32;
    float *X, *W, *B, *Y; uint8_t* idx; unsigned long long* st;
    CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&W, 27 * 32 * 4)); CK(hipMalloc(&B, 128)); CK(hipMalloc(&Y, ny)); CK(hipMalloc(&idx, ny / 4)); CK(hipMalloc(&st, (size_t)grid * 32 * 8));
    CK(hipMemset(X, 0, nx * 4)); CK(hipMemset(W, 0, 27 * 32 * 4)); CK(hipMemset(B, 0, 128));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rcnx_stamps), &st, sizeof(st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(st, 0, (size_t)grid * 32 * 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_conv1_fwd_f32<C1_CIN, 16, 4>), dim3(grid), dim3(kThreads), 0, 0, X, W, B, Y, s, tiles_w, tiles_h, items, idx);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; ++g) { t0 = std::min(t0, h[g * 32]); t1 = std::max(t1, h[g * 32 + 29]); }
    printf("conv1 fwd+pool: %d items on %d workgroups (%d per CU): %.1f us by events, %.2f us first start -> last end\n", items, grid, per_cu, ms * 1e3, (t1 - t0) / 100.0);
    fprintf(f, "# conv1 items %d grid %d per_cu %d nph 1 event_us %.1f\n", items, grid, per_cu, ms * 1e3);
    for (int g = 0; g < grid; ++g) {
        fprintf(f, "%d,%llu,%llu", g, h[g * 32 + 30], h[g * 32 + 31]);
        for (int k = 0; k < 30; ++k) fprintf(f, ",%lld", h[g * 32 + k] ? (long long)(h[g * 32 + k] - t0) : -1LL);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
