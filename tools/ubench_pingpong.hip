// Ping-pong of one tagged 8-byte word between two workgroups of one kernel: how long does a store by one workgroup take to
// be seen by a polling load of another, as a function of where the two run (same XCD or not: workgroup i runs on XCD i % 8)
// and of the scope of the atomics.  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/pp tools/ubench_pingpong.hip && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
using u64 = unsigned long long;

template <int SCOPE>
__global__ void k_pp(u64* a, u64* b, int wgA, int wgB, int iters, long long* ticks) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != wgA && me != wgB) return;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (me == wgA) {
            __hip_atomic_store(a, (u64)i, __ATOMIC_RELAXED, SCOPE);
            long long guard = 0;
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, SCOPE) != (u64)i) if (++guard > 50000000) return;
        } else {
            long long guard = 0;
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, SCOPE) != (u64)i) if (++guard > 50000000) return;
            __hip_atomic_store(b, (u64)i, __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (me == wgA) *ticks = wall_clock64() - t0;
}

template <int SCOPE>
void run(const char* name, int wgA, int wgB) {
    u64 *a, *b; long long* t;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&t, 8);
    hipMemset(a, 0, 256); hipMemset(b, 0, 256); hipMemset(t, 0, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k_pp<SCOPE>), dim3(64), dim3(64), 0, 0, a, b + 16, wgA, wgB, iters, t);
    hipDeviceSynchronize();
    long long ticks = 0;
    hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s wg %2d <-> wg %2d : %7.1f ns per one-way hand-off\n", name, wgA, wgB, ticks * 10.0 / iters / 2);
    hipFree(a); hipFree(b); hipFree(t);
}

int main() {
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope", 0, 1);
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope", 0, 8);
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope", 0, 32);
    run<__HIP_MEMORY_SCOPE_SYSTEM>("system scope", 0, 1);
    run<__HIP_MEMORY_SCOPE_SYSTEM>("system scope", 0, 8);
    run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup scope (not a legal cross-WG scope)", 0, 8);
    run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup scope (not a legal cross-WG scope)", 0, 1);
    return 0;
}
