// One hop of the pushed exchange (csrc/dp_push.hpp), measured between two PROCESSES: a {value, step} word stored by one 8-byte
// system-scope store into the PEER's uncached, hipIpc-mapped memory, seen by the peer's poll of its OWN memory.  DESIGN section 7
// derives the N = 8 estimate from "0.7-1 us per posted-store hop"; this gives that figure a measured same-device lower bound (both
// processes share this box's one GPU -- the store crosses no xGMI link, everything else of the path is the product's: uncached
// allocation, IPC mapping, system scope, local poll).  Variants: hop between blocks on the SAME physical XCD and on DIFFERENT XCDs (a block leaves unless it sits on the XCD
// it was told to), and `burst` words per hop (the reduce-scatter pushes four words per lane).
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/push_hop tools/ubench_push_hop.hip && HSA_ENABLE_IPC_MODE_LEGACY=0 /tmp/push_hop
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
using u64 = unsigned long long;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

// 256 blocks so that every XCD gets 32; the ONE block that does the work is the first block (lowest blockIdx) on physical XCD `want`.
// rank 0 sends first.  words: [0..burst) the hop's payload; result: ticks of the 100 MHz clock for `iters` round trips.
__global__ void k_hop(u64* mine, u64* peer, int rank, int want_xcc, int iters, int burst, long long* ticks, unsigned* claim) {
    if (threadIdx.x != 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if ((int)(id & 0xfu) != want_xcc) return;
    if (atomicAdd(claim, 1u) != 0u) return;                               // one worker per process
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (rank == 0)
            for (int b = 0; b < burst; ++b) __hip_atomic_store(peer + b, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        long long guard = 0;
        for (int b = 0; b < burst; ++b)
            while (__hip_atomic_load(mine + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (u64)i)
                if (++guard > 20000000LL) { *ticks = -1; return; }
        if (rank == 1)
            for (int b = 0; b < burst; ++b) __hip_atomic_store(peer + b, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *ticks = wall_clock64() - t0;
}

static void xfer(int fd_out, int fd_in, const void* out, void* in, size_t n) {
    if (write(fd_out, out, n) != (ssize_t)n || read(fd_in, in, n) != (ssize_t)n) { perror("pipe"); exit(2); }
}

static int rank_main(int rank, int fd_out, int fd_in) {
    u64* mine = nullptr;
    CK(hipExtMallocWithFlags((void**)&mine, 1 << 16, hipDeviceMallocUncached));
    CK(hipMemset(mine, 0, 1 << 16));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h, hp;
    CK(hipIpcGetMemHandle(&h, mine));
    xfer(fd_out, fd_in, &h, &hp, sizeof h);
    u64* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    long long* ticks;
    unsigned* claim;
    CK(hipMalloc(&ticks, 8));
    CK(hipMalloc(&claim, 4));
    const int iters = 4000;
    struct { const char* name; int xcc0, xcc1, burst; } cases[] = {
        {"one word, XCD 0 <-> XCD 1", 0, 1, 1}, {"one word, both on XCD 2", 2, 2, 1}, {"four words, XCD 0 <-> XCD 1", 0, 1, 4}, {"four words, XCD 3 <-> XCD 6", 3, 6, 4}};
    int slot = 0;
    for (auto& cs : cases) {
        CK(hipMemset(ticks, 0, 8));
        CK(hipMemset(claim, 0, 4));
        CK(hipDeviceSynchronize());
        char go = 1, got = 0;
        xfer(fd_out, fd_in, &go, &got, 1);                                // both ranks launch together
        // every case on its own 256-byte slot of the two buffers (no word of an earlier case is ever looked at again)
        hipLaunchKernelGGL(k_hop, dim3(256), dim3(64), 0, 0, mine + 32 * slot, peer + 32 * slot, rank, rank == 0 ? cs.xcc0 : cs.xcc1, iters, cs.burst, ticks, claim);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        long long t = 0;
        CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
        if (rank == 0) {
            if (t <= 0) printf("%-34s : FAILED (ticks %lld)\n", cs.name, t);
            else printf("%-34s : %7.1f ns per hop (store -> seen by the peer's local poll; %d round trips / 2)\n", cs.name, t * 10.0 / iters / 2, iters);
            fflush(stdout);
        }
        ++slot;
    }
    char go = 1, got = 0;
    xfer(fd_out, fd_in, &go, &got, 1);                                    // nobody unmaps while the peer may still read
    CK(hipIpcCloseMemHandle(peer));
    return 0;
}

int main() {
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    int a[2], b[2];
    if (pipe(a) || pipe(b)) { perror("pipe"); return 2; }
    const pid_t pid = fork();                                             // before any HIP call in this process
    if (pid == 0) return rank_main(1, b[1], a[0]);
    const int rc = rank_main(0, a[1], b[0]);
    int st = 0;
    waitpid(pid, &st, 0);
    return rc ? rc : (WIFEXITED(st) ? WEXITSTATUS(st) : 3);
}
