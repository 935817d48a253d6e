// What does one step's communication cost when every workgroup of the step sits on ONE XCD (32 CUs, one coherent 4 MiB L2)?
//
// The two-kernel step pays two launch boundaries (~1.5 us each) and two cold hand-offs.  A resident kernel pays two in-launch
// hand-offs instead; round 1 measured those at 3.5 + 1.75 us with 85 workgroups spread over all eight XCDs (write-through stores,
// L2 misses on every consumer read).  Here: NW workgroups, all with blockIdx.x % 8 == 0 (blocks are dealt round-robin over the
// XCDs, so they share one; verified through XCC_ID), roles as in the training step:
//   phase 1   NA producers each write a 32 KB slab part (1 KB per consumer)                       -> flagA
//   phase 2   NB consumers wait for all flagA, read their NA x 1 KB, check it, write 1 KB of "deltas" -> flagB
//   phase 3   the NA producers wait for all flagB, read the whole 32 KB of deltas, check it
// with the payload stored PLAIN (stays dirty in the XCD's L2; legal only because every reader shares that L2) or write-through (sc1),
// and read with L1-bypassing (sc1) loads.  Reports us per iteration (= the communication floor of a resident step) and the number
// of stale / wrong words seen (must be 0).
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/xr tools/ubench_xcd_ring.hip && /tmp/xr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u4 = __attribute__((ext_vector_type(4))) unsigned;
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((ptr), 0, (int)(bytes), 0x00020000)

template <int AUX> __device__ inline void st4(__amdgpu_buffer_rsrc_t r, int off, u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX); }
template <int AUX> __device__ inline u4 ld4(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX); }
__device__ inline void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ inline unsigned word(unsigned it, unsigned prod, unsigned idx) { return it * 0x9E3779B1u ^ (prod << 20) ^ idx; }

__device__ inline bool wait_flags(const unsigned* flags, int n, unsigned it) {
    // one wave: lane i looks at flag i
    const int lane = threadIdx.x & 63;
    for (long long guard = 0; guard < 4000000; ++guard) {
        const unsigned f = lane < n ? __hip_atomic_load(flags + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : it;
        if (__all((int)(f - it) >= 0)) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int ST_AUX, int T, int FLAGP>
__global__ __launch_bounds__(T) void k_ring(unsigned* slab, unsigned* dl, unsigned* flagA, unsigned* flagB, int NA, int NB, int stride8, int iters,
                                            unsigned* xcc, unsigned long long* bad, long long* ticks) {
    if (stride8 && (blockIdx.x & 7) != 0) return;
    const int w = stride8 ? blockIdx.x >> 3 : blockIdx.x;
    const int NW = NA > NB ? NA : NB;
    if (w >= NW) return;
    const int tid = threadIdx.x;
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[w] = id & 0xf; }
    __shared__ int s_ok;
    const size_t slab_half = (size_t)NA * NB * 256, dl_half = (size_t)NB * 256;           // words
    const auto r_slab = RSRC(slab, 2 * slab_half * 4), r_dl = RSRC(dl, 2 * dl_half * 4);
    unsigned long long wrong = 0;
    const long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        const int par = it & 1;
        if (w < NA) {                                                       // ---- phase 1: my 32 KB (NB x 1 KB), 16 B per lane
            for (int e = tid; e < NB * 64; e += T) {
                const int t = e >> 6, q = e & 63;
                u4 v{word(it, w, 4 * e), word(it, w, 4 * e + 1), word(it, w, 4 * e + 2), word(it, w, 4 * e + 3)};
                st4<ST_AUX>(r_slab, (int)(((size_t)par * slab_half + ((size_t)t * NA + w) * 256 + 4 * q) * 4), v);
            }
            drain();
            __syncthreads();
            if (tid == 0) { if (FLAGP) { __hip_atomic_store(flagA + w * 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } else __hip_atomic_store(flagA + w * 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (w < NB) {                                                       // ---- phase 2: NA x 1 KB for me
            if (tid < 64) { const bool ok = wait_flags(flagA, NA, it); if (tid == 0) s_ok = ok; }
            __syncthreads();
            if (!s_ok) { if (tid == 0) atomicAdd(bad, 1ull << 40); return; }
            for (int e = tid; e < NA * 64; e += T) {
                const int p = e >> 6, q = e & 63;
                const u4 v = ld4<16>(r_slab, (int)(((size_t)par * slab_half + ((size_t)w * NA + p) * 256 + 4 * q) * 4));
                const int pe = w * 64 + q;
                wrong += (v[0] != word(it, p, 4 * pe)) + (v[1] != word(it, p, 4 * pe + 1)) + (v[2] != word(it, p, 4 * pe + 2)) + (v[3] != word(it, p, 4 * pe + 3));
            }
            if (tid < 64) {                                                 // my 1 KB of deltas
                u4 v{word(it, 100 + w, 4 * tid), word(it, 100 + w, 4 * tid + 1), word(it, 100 + w, 4 * tid + 2), word(it, 100 + w, 4 * tid + 3)};
                st4<ST_AUX>(r_dl, (int)(((size_t)par * dl_half + (size_t)w * 256 + 4 * tid) * 4), v);
                drain();
            }
            __syncthreads();
            if (tid == 0) { if (FLAGP) { __hip_atomic_store(flagB + w * 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } else __hip_atomic_store(flagB + w * 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (w < NA) {                                                       // ---- phase 3: all NB x 1 KB
            if (tid < 64) { const bool ok = wait_flags(flagB, NB, it); if (tid == 0) s_ok = ok; }
            __syncthreads();
            if (!s_ok) { if (tid == 0) atomicAdd(bad, 1ull << 40); return; }
            for (int e = tid; e < NB * 64; e += T) {
                const int t = e >> 6, q = e & 63;
                const u4 v = ld4<16>(r_dl, (int)(((size_t)par * dl_half + (size_t)t * 256 + 4 * q) * 4));
                wrong += (v[0] != word(it, 100 + t, 4 * q)) + (v[1] != word(it, 100 + t, 4 * q + 1)) + (v[2] != word(it, 100 + t, 4 * q + 2)) + (v[3] != word(it, 100 + t, 4 * q + 3));
            }
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    if (w == 0 && tid == 0) *ticks = wall_clock64() - t0;
}

template <int ST_AUX, int T, int FLAGP = 0>
void run(const char* name, int NA, int NB, int stride8, int iters) {
    const int NW = NA > NB ? NA : NB;
    unsigned *slab, *dl, *fa, *fb, *xcc; unsigned long long* bad; long long* ticks;
    hipMalloc(&slab, (size_t)2 * NA * NB * 1024); hipMalloc(&dl, (size_t)2 * NB * 1024);
    hipMalloc(&fa, 64 * 64); hipMalloc(&fb, 64 * 64); hipMalloc(&xcc, 256); hipMalloc(&bad, 8); hipMalloc(&ticks, 8);
    hipMemset(slab, 0, (size_t)2 * NA * NB * 1024); hipMemset(dl, 0, (size_t)2 * NB * 1024);
    hipMemset(fa, 0, 64 * 64); hipMemset(fb, 0, 64 * 64); hipMemset(xcc, 0xff, 256); hipMemset(bad, 0, 8); hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL((k_ring<ST_AUX, T, FLAGP>), dim3(stride8 ? 8 * NW : NW), dim3(T), 0, 0, slab, dl, fa, fb, NA, NB, stride8, iters, xcc, bad, ticks);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long hb = 0; long long ht = 0; std::vector<unsigned> hx(64);
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, 256, hipMemcpyDeviceToHost);
    int same = 1;
    for (int i = 1; i < NW; ++i) same &= hx[i] == hx[0];
    printf("%-26s T=%4d NA=%2d NB=%2d %-9s: %6.2f us/iter   wrong words %llu  timeouts %llu  xcc %s (first %u)  %s\n", name, T, NA, NB, stride8 ? "one XCD" : "spread",
           ht * 0.01 / iters, hb & ((1ull << 40) - 1), hb >> 40, same ? "all equal" : "MIXED", hx[0], e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(slab); hipFree(dl); hipFree(fa); hipFree(fb); hipFree(xcc); hipFree(bad); hipFree(ticks);
}

int main() {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 512>("plain stores, sc1 loads", 28, 32, 1, iters);
        run<16, 512>("sc1 stores, sc1 loads", 28, 32, 1, iters);
        run<16, 512>("sc1 stores, sc1 loads", 28, 32, 0, iters);
        run<0, 512>("plain stores (ILLEGAL)", 28, 32, 0, iters);
        run<0, 1024>("plain stores, sc1 loads", 28, 32, 1, iters);
        run<16, 1024>("sc1 stores, sc1 loads", 28, 32, 1, iters);
        run<0, 512, 1>("plain stores+flags, sc1 ld", 28, 32, 1, iters);
        run<0, 1024, 1>("plain stores+flags, sc1 ld", 28, 32, 1, iters);
        run<0, 512>("plain stores, sc1 loads", 16, 16, 1, iters);
        run<0, 512>("plain stores, sc1 loads", 32, 32, 1, iters);
    }
    return 0;
}
