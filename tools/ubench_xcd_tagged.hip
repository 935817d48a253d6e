// The two hand-offs of a resident step on ONE XCD (tools/ubench_xcd_ring.hip) with SELF-VALIDATING payloads instead of flags:
// every 16-byte piece is {w0, w1, w2, iteration}, stored plain (stays in the XCD's L2), never drained, never announced; a consumer
// polls the pieces themselves with sc1 loads until each carries this iteration's tag.  What it would save in k_xcd_epoch: per hand-off
// the producer's drain (s_waitcnt vmcnt(0): ~0.4 us until the L2 has acknowledged the stores), its barrier and flag store, and the
// consumer's dependent second round trip (flag seen -> THEN payload loads).  What it costs: 4/3 the bytes (three payload words per
// piece), the polls re-read payload instead of one flag line, and it relies on a 16-byte store being seen whole by a 16-byte load
// (counted here: `torn` = pieces whose tag matched but whose words did not).
//   MODE 0: flags (the form k_xcd_epoch uses: plain stores, drain, barrier, plain flag, consumer polls flags then loads)
//   MODE 1: tagged pieces, consumers poll from the moment they arrive
// Same roles and sizes as the training step at B = 256: NA = 25 producers x NB = 32 consumers x 1 KB of payload, then 32 x 1 KB back
// to all 25 (+3 tail readers).
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/xt tools/ubench_xcd_tagged.hip && /tmp/xt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u4 = __attribute__((ext_vector_type(4))) unsigned;
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((ptr), 0, (int)(bytes), 0x00020000)
__device__ inline void st4(__amdgpu_buffer_rsrc_t r, int off, u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }
__device__ inline u4 ld4(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ inline void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline unsigned word(unsigned it, unsigned prod, unsigned idx) { return it * 0x9E3779B1u ^ (prod << 20) ^ idx; }

__device__ inline bool wait_flags(const unsigned* flags, int n, unsigned it) {
    const int lane = threadIdx.x & 63;
    for (long long guard = 0; guard < 4000000; ++guard) {
        const unsigned f = lane < n ? __hip_atomic_load(flags + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : it;
        if (__all((int)(f - it) >= 0)) return true;
    }
    return false;
}

constexpr int T = 512;
// pieces per 1 KB tile: flags 64 (four payload words each), tagged 86 (three payload words + tag)
template <int MODE> constexpr int PPT() { return MODE == 0 ? 64 : 86; }

template <int MODE>
__global__ __launch_bounds__(T) void k_ring(unsigned* slab, unsigned* dl, unsigned* flagA, unsigned* flagB, int NA, int NB, int NR, int iters,
                                            unsigned* xcc, unsigned long long* bad, unsigned long long* torn, long long* ticks) {
    if ((blockIdx.x & 7) != 0) return;
    const int w = blockIdx.x >> 3;
    const int NW = NB > NR ? NB : NR;
    if (w >= NW) return;
    constexpr int P = PPT<MODE>();
    const int tid = threadIdx.x;
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[w] = id & 0xf; }
    __shared__ int s_ok, s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const size_t slab_half = (size_t)NA * NB * P * 4, dl_half = (size_t)NB * P * 4;           // words
    const auto r_slab = RSRC(slab, 2 * slab_half * 4), r_dl = RSRC(dl, 2 * dl_half * 4);
    unsigned long long wrong = 0, tornw = 0;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        const int par = it & 1;
        if (w < NA) {                                                       // ---- phase 1: NB tiles of P pieces, 16 B per lane
            for (int e = tid; e < NB * P; e += T) {
                const int t = e / P, q = e % P;
                u4 v{word(it, w, 4 * e), word(it, w, 4 * e + 1), word(it, w, 4 * e + 2), MODE ? (unsigned)it : word(it, w, 4 * e + 3)};
                st4(r_slab, (int)(((size_t)par * slab_half + (((size_t)t * NA + w) * P + q) * 4) * 4), v);
            }
            if (MODE == 0) {
                drain();
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flagA + w * 32, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (w < NB) {                                                       // ---- phase 2: NA tiles for me
            if (MODE == 0) {
                if (tid < 64) { const bool ok = wait_flags(flagA, NA, it); if (tid == 0) s_ok = ok; }
                __syncthreads();
                if (!s_ok) { if (tid == 0) atomicAdd(bad, 1ull << 40); return; }
            }
            for (int e = tid; e < NA * P; e += T) {
                const int p = e / P, q = e % P;
                const int off = (int)(((size_t)par * slab_half + (((size_t)w * NA + p) * P + q) * 4) * 4);
                u4 v = ld4(r_slab, off);
                if (MODE) {
                    long long guard = 0;
                    while (v[3] != (unsigned)it && !s_fail) { v = ld4(r_slab, off); if (++guard > 4000000) { atomicAdd(bad, 1ull << 40); s_fail = 1; } }
                }
                const int pe = w * P + q;
                const unsigned long long m = (v[0] != word(it, p, 4 * pe)) + (v[1] != word(it, p, 4 * pe + 1)) + (v[2] != word(it, p, 4 * pe + 2)) +
                                             (MODE ? 0 : (v[3] != word(it, p, 4 * pe + 3)));
                wrong += m; tornw += MODE ? (m != 0) : 0;
            }
            if (MODE) { __syncthreads(); if (s_fail) return; }            // (the real step computes on the whole tile: everybody has to have his pieces)
            if (tid < P) {                                                  // my tile of "deltas"
                u4 v{word(it, 100 + w, 4 * tid), word(it, 100 + w, 4 * tid + 1), word(it, 100 + w, 4 * tid + 2), MODE ? (unsigned)it : word(it, 100 + w, 4 * tid + 3)};
                st4(r_dl, (int)(((size_t)par * dl_half + ((size_t)w * P + tid) * 4) * 4), v);
                if (MODE == 0) drain();
            }
            if (MODE == 0) {
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flagB + w * 32, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (w < NR) {                                                       // ---- phase 3: all NB tiles (feature workers and tail tiles)
            if (MODE == 0) {
                if (tid < 64) { const bool ok = wait_flags(flagB, NB, it); if (tid == 0) s_ok = ok; }
                __syncthreads();
                if (!s_ok) { if (tid == 0) atomicAdd(bad, 1ull << 40); return; }
            }
            for (int e = tid; e < NB * P; e += T) {
                const int t = e / P, q = e % P;
                const int off = (int)(((size_t)par * dl_half + ((size_t)t * P + q) * 4) * 4);
                u4 v = ld4(r_dl, off);
                if (MODE) {
                    long long guard = 0;
                    while (v[3] != (unsigned)it && !s_fail) { v = ld4(r_dl, off); if (++guard > 4000000) { atomicAdd(bad, 1ull << 40); s_fail = 1; } }
                }
                const unsigned long long m = (v[0] != word(it, 100 + t, 4 * q)) + (v[1] != word(it, 100 + t, 4 * q + 1)) + (v[2] != word(it, 100 + t, 4 * q + 2)) +
                                             (MODE ? 0 : (v[3] != word(it, 100 + t, 4 * q + 3)));
                wrong += m; tornw += MODE ? (m != 0) : 0;
            }
            if (MODE) { __syncthreads(); if (s_fail) return; }
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    if (tornw) atomicAdd(torn, tornw);
    if (w == 0 && tid == 0) *ticks = wall_clock64() - t0;
}

template <int MODE>
void run(const char* name, int NA, int NB, int NR, int iters) {
    const int NW = NB > NR ? NB : NR;
    constexpr int P = PPT<MODE>();
    unsigned *slab, *dl, *fa, *fb, *xcc; unsigned long long *bad, *torn; long long* ticks;
    const size_t sb = (size_t)2 * NA * NB * P * 16, db = (size_t)2 * NB * P * 16;
    hipMalloc(&slab, sb); hipMalloc(&dl, db);
    hipMalloc(&fa, 64 * 128); hipMalloc(&fb, 64 * 128); hipMalloc(&xcc, 256); hipMalloc(&bad, 8); hipMalloc(&torn, 8); hipMalloc(&ticks, 8);
    hipMemset(slab, 0, sb); hipMemset(dl, 0, db);
    hipMemset(fa, 0, 64 * 128); hipMemset(fb, 0, 64 * 128); hipMemset(xcc, 0xff, 256); hipMemset(bad, 0, 8); hipMemset(torn, 0, 8); hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL((k_ring<MODE>), dim3(8 * NW), dim3(T), 0, 0, slab, dl, fa, fb, NA, NB, NR, iters, xcc, bad, torn, ticks);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long hb = 0, htn = 0; long long ht = 0; std::vector<unsigned> hx(64);
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&htn, torn, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), xcc, 256, hipMemcpyDeviceToHost);
    int same = 1;
    for (int i = 1; i < NW; ++i) same &= hx[i] == hx[0];
    printf("%-34s NA=%2d NB=%2d NR=%2d: %6.2f us/iter   wrong words %llu  torn pieces %llu  timeouts %llu  xcc %s  %s\n", name, NA, NB, NR, ht * 0.01 / iters,
           hb & ((1ull << 40) - 1), htn, hb >> 40, same ? "all equal" : "MIXED", e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(slab); hipFree(dl); hipFree(fa); hipFree(fb); hipFree(xcc); hipFree(bad); hipFree(torn); hipFree(ticks);
}

int main() {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("flags (drain, barrier, flag; poll)", 25, 32, 28, iters);
        run<1>("tagged 16-byte pieces, no flags", 25, 32, 28, iters);
        run<0>("flags (drain, barrier, flag; poll)", 25, 4, 28, iters);       // the roles of a batch of 32: four sample groups
        run<1>("tagged 16-byte pieces, no flags", 25, 4, 28, iters);
    }
    return 0;
}
