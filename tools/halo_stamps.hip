// Timeline of k_conv3x3_halo_f32 workgroups on one layer shape (default: CIFAR conv2 forward + fused pool, 512 x 16 x 16 x 32 -> 64):
// raw stamps (100 MHz wall clock, 10 ns) of every workgroup as CSV for tools/halo_stamps_report.py.  Stamp order per workgroup:
// entry, then per item { per phase { operands staged, next loads issued }, MFMAs issued, epilogue issued }, ..., slot 29 = exit.
//   hipcc -O3 --offload-arch=gfx950 -DRCNX_STAMPS -o /tmp/halo_stamps tools/halo_stamps.hip && /tmp/halo_stamps out.csv [N H W Cin Cout [wg_per_cu]]
#include "../mercer_research_amd/csrc/convnet_halo.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rcnx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int TW, int BN, int EPI, bool PIN = false> int run(ConvShape s, int slots_per_cu, FILE* f) {
    const int nimg = 16 / TW, tiles_w = (s.W + TW - 1) / TW, tiles_h = (s.H + 7) / 8;
    const int items = tiles_w * tiles_h * ((s.N + nimg - 1) / nimg) * (s.Cout / BN);
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_conv3x3_halo_f32<TW, BN, EPI, PIN>, kThreads, 0));
    if (slots_per_cu > 0) per_cu = slots_per_cu;
    const int grid = std::min(items, per_cu * 256);
    const size_t nx = (size_t)s.N * s.H * s.W * s.Cin, ny = (size_t)s.N * s.H * s.W * s.Cout, nw = (size_t)9 * s.Cin * s.Cout;
    float *X, *W, *B, *Y; uint8_t* idx; unsigned long long* st;
    CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&B, s.Cout * 4)); CK(hipMalloc(&Y, ny * 4)); CK(hipMalloc(&idx, ny)); CK(hipMalloc(&st, (size_t)grid * 32 * 8));
    CK(hipMemset(X, 0, nx * 4)); CK(hipMemset(W, 0, nw * 4)); CK(hipMemset(B, 0, s.Cout * 4)); CK(hipMemset(st, 0, (size_t)grid * 32 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rcnx_stamps), &st, sizeof(st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // PIN: the input is a pooled-resolution gradient (dP, P, arg-max) of a quarter of the pixels
    float *dP = nullptr, *Pp = nullptr; uint8_t* pidx = nullptr;
    if (PIN) { CK(hipMalloc(&dP, nx)); CK(hipMalloc(&Pp, nx)); CK(hipMalloc(&pidx, nx / 4)); CK(hipMemset(dP, 0, nx)); CK(hipMemset(Pp, 0, nx)); CK(hipMemset(pidx, 0, nx / 4)); }
    const PooledGrad pg{dP, Pp, pidx};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(st, 0, (size_t)grid * 32 * 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_conv3x3_halo_f32<TW, BN, EPI, PIN>), dim3(grid), dim3(kThreads), 0, 0, X, W, B, Y, s, tiles_w, tiles_h, items, idx, pg);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; ++g) { t0 = std::min(t0, h[g * 32]); t1 = std::max(t1, h[g * 32 + 29]); }
    printf("PIN %d TW %d BN %d EPI %d: %d items on %d workgroups (%d per CU), %d phases per item: %.1f us by events, %.2f us first start -> last end\n", (int)PIN, TW, BN, EPI, items, grid, per_cu,
           s.Cin / 32 * 3, ms * 1e3, (t1 - t0) / 100.0);
    fprintf(f, "# TW %d BN %d EPI %d items %d grid %d per_cu %d nph %d event_us %.1f\n", TW, BN, EPI, items, grid, per_cu, s.Cin / 32 * 3, ms * 1e3);
    for (int g = 0; g < grid; ++g) {
        fprintf(f, "%d,%llu,%llu", g, h[g * 32 + 30], h[g * 32 + 31]);
        for (int k = 0; k < 30; ++k) fprintf(f, ",%lld", h[g * 32 + k] ? (long long)(h[g * 32 + k] - t0) : -1LL);
        fprintf(f, "\n");
    }
    hipFree(X); hipFree(W); hipFree(B); hipFree(Y); hipFree(idx); hipFree(st);
    return 0;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argc >= 2 ? argv[1] : "/tmp/halo_stamps.csv", "w");
    if (!f) return 1;
    ConvShape s{512, 16, 16, 32, 64};
    if (argc >= 7) s = ConvShape{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6])};
    const int spc = argc >= 8 ? atoi(argv[7]) : 0;
    const int mode = argc >= 9 ? atoi(argv[8]) : 0;           // 0: forward + pool (EPI 4); 1: input gradient from a pooled gradient (PIN, EPI 0)
    const int bn = argc >= 10 ? atoi(argv[9]) : (s.Cout % 64 == 0 ? 64 : 32);
    if (mode == 0) {
        if (s.W % 16 == 0) { if (bn == 64) run<16, 64, 4>(s, spc, f); else run<16, 32, 4>(s, spc, f); }
        else { if (bn == 64) run<8, 64, 4>(s, spc, f); else run<8, 32, 4>(s, spc, f); }
    } else {
        if (s.W % 16 == 0) { if (bn == 64) run<16, 64, 0, true>(s, spc, f); else run<16, 32, 0, true>(s, spc, f); }
        else { if (bn == 64) run<8, 64, 0, true>(s, spc, f); else run<8, 32, 0, true>(s, spc, f); }
    }
    fclose(f);
    return 0;
}
