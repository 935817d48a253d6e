// dp_p2p.hpp -- one-shot all-reduce of the flat gradient over xGMI peer reads, fused with the SGD update.
//
// Why not RCCL for this message: the data-parallel step exchanges P+1 values (95 KB for the default net) once per
// ~15 us of compute; a library collective costs several launches' worth of latency at that size.  xGMI is a
// point-to-point fabric -- every GPU can read every peer's HBM directly -- so for a message this small the fastest
// all-reduce is the trivial one: every rank reads all `world` gradient buffers and adds them itself.
//
// Protocol (one kernel per step, per rank; all buffers are exported / mapped once with hipIpc):
//   data  : rank r owns buf_r[2][stride] in ordinary device memory.  Step s uses slot s&1; the gradient kernel of step
//           s writes buf_r[s&1] and has COMPLETED (stream order; end-of-kernel release at system scope) before this
//           kernel starts.
//   flags : rank r owns flags_r[world] in UNCACHED device memory (fine-grained: remote stores are visible to local
//           polling, no stale L2 line).  At the start of step s, rank r stores seq = s+1 into flags_q[r] of every q
//           (system-scope release) -- "my slot s&1 is ready".
//   wait  : every workgroup polls its OWN flags_r[q] >= seq for all q (system-scope acquire), bounded by a wall-clock
//           timeout that raises a sticky error word instead of hanging the GPU.
//   reduce: element i = sum over q = 0..world-1, IN RANK ORDER on every rank, of buf_q[slot][i] read with
//           system-scope loads (never served from a stale local cache line).  Every rank therefore computes the
//           bit-identical sum and replicas stay identical without a broadcast.  The update p <- p - scale * sum
//           (rcn.rs:214,221 with the global batch length) is applied in the same pass; element P carries the loss.
//   reuse : no end-of-step barrier.  Rank r overwrites slot s&1 next at step s+2, which it can only reach after every
//           peer signalled step s+1, i.e. after every peer finished reading step s (flags are monotonic sequence
//           numbers, never reset).
#pragma once

#include "common.hpp"

namespace rcn {

constexpr int kP2PMaxWorld = 8;
constexpr int kP2PThreads = 256;

struct P2PDesc {
    int world, rank;
    void* buf[kP2PMaxWorld];          // rank q's double buffer, mapped into this process (own entry: the local pointer)
    unsigned* flags[kP2PMaxWorld];    // rank q's flag array [world]
};

template <typename T> struct P2PWord;
template <> struct P2PWord<float> { static constexpr int per = 2; };     // values per 8-byte system-scope load
template <> struct P2PWord<double> { static constexpr int per = 1; };

// mode 0: params <- params - scale * sum, loss_out <- sum[P];   mode 1 (self-test): out <- sum
template <typename T>
__global__ __launch_bounds__(kP2PThreads) void k_p2p_allreduce(P2PDesc d, unsigned seq, size_t stride, int P, T* __restrict__ params, T scale,
                                                               T* __restrict__ loss_out, T* __restrict__ raw_out, int mode,
                                                               unsigned* __restrict__ err, long long timeout_ticks) {
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = *err != 0u ? 2 : 0;               // a previous step failed: drain without touching anything
    __syncthreads();
    if (s_bad == 2) return;
    if (blockIdx.x == 0 && tid < d.world)
        __hip_atomic_store(d.flags[tid] + d.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < d.world && tid != d.rank) {                  // own slot: complete by stream order, nothing to wait for
        const unsigned* mine = d.flags[d.rank] + tid;
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (wall_clock64() - t0 > timeout_ticks) { s_bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) *err = 1u + (unsigned)d.rank;
        return;
    }
    constexpr int per = P2PWord<T>::per;
    const size_t words = stride / per;                     // stride is a multiple of 4 values
    const size_t slot_w = (size_t)(seq & 1u) * words;
    for (size_t w = (size_t)blockIdx.x * kP2PThreads + tid; w < words; w += (size_t)gridDim.x * kP2PThreads) {
        T acc[per];
#pragma unroll
        for (int e = 0; e < per; ++e) acc[e] = 0;
        unsigned long long raw[kP2PMaxWorld];
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q) {
            const int qc = q < d.world ? q : 0;                                    // unconditional loads, masked by value
            raw[q] = __hip_atomic_load((const unsigned long long*)d.buf[qc] + slot_w + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q) {
            if (q < d.world) {
                T v[per];
                __builtin_memcpy(v, &raw[q], 8);
#pragma unroll
                for (int e = 0; e < per; ++e) acc[e] += v[e];                       // rank order: identical on every rank
            }
        }
#pragma unroll
        for (int e = 0; e < per; ++e) {
            const size_t i = w * per + e;
            if (mode == 1) { raw_out[i] = acc[e]; continue; }
            if (i < (size_t)P) params[i] = params[i] - scale * acc[e];
            else if (i == (size_t)P && loss_out) *loss_out = acc[e];
        }
    }
}

// self-test pattern: small integers, exact in f32 for any summation order
__device__ inline float p2p_pattern(int rank, unsigned seq, size_t i) {
    const unsigned h = (unsigned)(i * 2654435761u) ^ (seq * 40503u) ^ ((unsigned)rank * 977u);
    return (float)((h >> 7) & 1023u) - 512.f;
}

template <typename T>
__global__ void k_p2p_fill(T* __restrict__ dst, size_t n, int rank, unsigned seq) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (T)p2p_pattern(rank, seq, i);
}

template <typename T>
__global__ void k_p2p_check(const T* __restrict__ got, size_t n, int world, unsigned seq, unsigned* __restrict__ mismatches) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T want = 0;
        for (int q = 0; q < world; ++q) want += (T)p2p_pattern(q, seq, i);
        if (got[i] != want) atomicAdd(mismatches, 1u);
    }
}

}  // namespace rcn
