// dense_p2_dp.hpp -- the feature-sliced pipeline (dense_p2.hpp) for the data-parallel step, on the peer-read exchange of
// dp_p2p.hpp.  Single-GPU step: k_p2_b, k_p2_a (update + next partials).  Data-parallel step, per rank:
//
//   k_p2_b            unchanged: slab -> a_1, layer 2, loss parts, delta_2, delta_1 of this rank's shard
//   k_p2_dp_grad      the U half of k_p2_a with the gradient going OUT: every feature slice's sum_s delta_1 (x) x (and the
//                     tail tiles db_0, d[W_1|b_1], and the shard's loss in element P) is written, unscaled, into this
//                     rank's exported slot -- at its parameter index, so the slot is laid out like the parameters
//   k_p2_dp_apply     the exchange + the rest of k_p2_a: signal "my slot is ready" (the data was written by the PREVIOUS
//                     kernel, so the end-of-kernel release already made it visible), wait for every peer's flag, then each
//                     thread adds ITS OWN parameter's gradient over all ranks in rank order (system-scope loads straight
//                     from the peers' HBM over xGMI), applies w <- w - eta/B_global * sum, keeps the updated W_0 slice in
//                     LDS and computes the next batch's partial z_1 from it -- so the all-reduce costs no kernel of its own
//                     and no rank ever materialises the reduced gradient.
//
// Same protocol, sequence numbers, double buffering and bounded waits as k_p2p_allreduce; the sums are in rank order on
// every rank, so replicas stay bit-identical.  The copies of k_p2_a's two loops below are deliberate: the single-GPU kernel
// is tuned at instruction level and is not touched.
#pragma once

#include "dense_p2.hpp"
#include "dp_p2p.hpp"

namespace rcn {

template <typename T> struct SysWord;
template <> struct SysWord<float> { using type = unsigned; };
template <> struct SysWord<double> { using type = unsigned long long; };

// one value of a (possibly remote) buffer, never served from a stale local cache line
template <typename T>
__device__ inline T load_sys(const T* p) {
    using W = typename SysWord<T>::type;
    const W w = __hip_atomic_load(reinterpret_cast<const W*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    T v;
    __builtin_memcpy(&v, &w, sizeof v);
    return v;
}

template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_p2_dp_grad(
    NetDesc nd, const T* __restrict__ Xp, int B, const T* __restrict__ a1, const T* __restrict__ d1, const T* __restrict__ d2,
    T* __restrict__ gbuf, size_t stride, const unsigned* __restrict__ seq_base, unsigned seq_off, int G, const T* __restrict__ loss_part,
    int n_loss, T loss_scale) {
    using acc_t = typename Mfma16<T>::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1];
    // step sequence number = *seq_base + seq_off: inside a replayed hipGraph the offset is baked in and the base is a device
    // word the host sets before every replay; launched eagerly, seq_base is null and seq_off is the number itself
    const unsigned seq = (seq_base ? *seq_base : 0u) + seq_off;
    T* __restrict__ gslot = gbuf + (size_t)(seq & 1u) * stride;
    if ((int)blockIdx.x >= G) {
        const int e = (int)blockIdx.x - G;
        if (e == 0 && tid == 0) finish_loss<T>(loss_part, n_loss, loss_scale, gslot + nd.P);     // this shard's part of the global cost
        if (e == 0) wgrad_tile_ld<T, false>(nd, 0, F, (T*)nullptr, gslot, (const T*)nullptr, 0, (const int*)nullptr, d1, kP2H, B, (T)0, red);
        else        wgrad_tile_ld<T, false>(nd, 1, (e - 1) * 16, (T*)nullptr, gslot, a1, kP2H, (const int*)nullptr, d2, kP2C, B, (T)0, red);
        return;
    }
    const int f0 = (int)blockIdx.x * 16;
    const int nf = F - f0 < 16 ? F - f0 : 16;
    const T* __restrict__ cp = Xp + (size_t)blockIdx.x * B * 16;
    const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8;
    const int m = mt * 16 + ml;
    const bool wvalid = m < H && cl < nf;
    acc_t acc[kMtp];
#pragma unroll
    for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
    const int kw = B >> 3;
    for (int kc = wave * kw; kc < (wave + 1) * kw; kc += 32) {
        const T* xb = cp + (size_t)(kc + g4) * 16 + n;
        const T* db = d1 + (size_t)(kc + g4) * kP2H + n;
        T bv[8], av[8][kMtp];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bv[q] = xb[q * 64];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) av[q][t] = db[q * 4 * kP2H + t * 16];
        }
        __builtin_amdgcn_sched_barrier(0);           // every load in flight before the first MFMA waits on one (dense_p2.hpp)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(av[q][t], bv[q], acc[t]);
    }
    store_partials<T>(red, wave, lane, acc);
    __syncthreads();
    const T gsum = sum_partials<T>(red, mt, cl, ml);
    if (wvalid) gslot[(size_t)nd.w_off[0] + (size_t)(f0 + cl) * H + m] = gsum;                    // rcn.rs:310 summed over the shard
}

template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_p2_dp_apply(
    NetDesc nd, T* __restrict__ params, const T* __restrict__ Xn, int B, T scale, T* __restrict__ slab, int G, T* __restrict__ loss_out,
    int do_fwd, P2PDesc d, const unsigned* __restrict__ seq_base, unsigned seq_off, size_t stride, unsigned* __restrict__ err,
    long long timeout_ticks) {
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    int& s_bad = *reinterpret_cast<int*>(red + kDenseWaves * kMtp * kRedTile + 16 * kP2H);      // in the slack behind the slice image
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1];
    const bool feat = (int)blockIdx.x < G;
    const unsigned seq = (seq_base ? *seq_base : 0u) + seq_off;
    const int f0 = (int)blockIdx.x * 16;
    const T* __restrict__ cn = Xn + (size_t)blockIdx.x * B * 16;

    // the next batch's rows do not depend on the exchange: issue them first
    vec4 xn[2];
    if (feat && do_fwd) {
#pragma unroll
        for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (wave + 8 * u) + n) * 16 + 4 * g4);
    }

    // ---- exchange: signal, then wait (dp_p2p.hpp)
    if (tid == 0) s_bad = *err != 0u ? 2 : 0;
    __syncthreads();
    if (s_bad == 2) return;
    if (blockIdx.x == 0 && tid < d.world) __hip_atomic_store(d.flags[tid] + d.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
    const size_t slot = (size_t)(seq & 1u) * stride;

    if (!feat) {
        // tail parameters b_0 | W_1 | b_1 (everything after W_0), spread over the tail workgroups; element P is the loss
        const int e = (int)blockIdx.x - G, ntw = (int)gridDim.x - G;
        const int t_begin = nd.w_off[0] + F * H;
        for (int p = t_begin + e * kDenseThreads + tid; p < nd.P; p += ntw * kDenseThreads) {
            T g = 0;
#pragma unroll
            for (int q = 0; q < kP2PMaxWorld; ++q) {
                const T v = load_sys<T>((const T*)d.buf[q < d.world ? q : 0] + slot + p);
                g += q < d.world ? v : (T)0;
            }
            params[p] = params[p] - scale * g;                                                    // rcn.rs:214,221
        }
        if (e == 0 && tid == 0 && loss_out) {
            T g = 0;
            for (int q = 0; q < d.world; ++q) g += load_sys<T>((const T*)d.buf[q] + slot + nd.P);
            *loss_out = g;
        }
        return;
    }

    // ---- this thread's element of the W_0 slice: gradient summed over ranks in rank order, update, keep the slice in LDS
    const int nf = F - f0 < 16 ? F - f0 : 16;
    T* wsl = red + kDenseWaves * kMtp * kRedTile;
    T* W0 = params + nd.w_off[0];
    {
        const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8;
        const int m = mt * 16 + ml;
        const bool wvalid = m < H && cl < nf;
        const size_t off = (size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0);
        const T wold = W0[off];
        T g = 0;
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q) {
            const T v = load_sys<T>((const T*)d.buf[q < d.world ? q : 0] + slot + nd.w_off[0] + off);   // unconditional, masked by value
            g += q < d.world ? v : (T)0;
        }
        const T w = wold - scale * g;
        if (wvalid) W0[off] = w;
        wsl[cl * kP2H + m] = wvalid ? w : (T)0;
    }
    __syncthreads();
    if (!do_fwd) return;

    // ---- F: partial z_1 of the new batch from the slice in LDS (k_p2_a's F half)
    const int ntile = B >> 4;
    T wf[4][kMtp];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < kMtp; ++t) wf[i][t] = wsl[(4 * g4 + i) * kP2H + t * 16 + n];
    for (int tb = 0; tb < ntile; tb += 16) {
        if (tb > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (tb + wave + 8 * u) + n) * 16 + 4 * g4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = 16 * (tb + wave + 8 * u) + n;
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(wf[i][t], xn[u][i], acc[t]);
            T* dst = slab + (((size_t)(s >> 3) * G + blockIdx.x) * kP2Ts + (s & 7)) * kP2H;
#pragma unroll
            for (int t = 0; t < kMtp; ++t) store4<T>(dst + t * 16, lane, acc[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two kernels per step: the exchange INSIDE the gradient kernel, on self-validating words.
//
// k_p2_dp_fused = k_p2_a with the peer exchange between its two halves.  Every thread owns one parameter; it needs from
// each peer exactly that parameter's partial gradient, produced by the peer's thread of the same index.  Flags plus
// fences would make that a release/acquire problem across GPUs inside a running kernel (tried: __threadfence_system
// costs an L2 write-back per workgroup, +8 us per step).  Instead each value travels WITH its step number in one 8-byte
// word -- {value bits, seq} for f32, two words {half, seq} for f64 -- written by one system-scope atomic store and polled
// by the consumer until the tag equals the step it is in (the "LL" idea of NCCL's low-latency protocol).  A word is
// single-copy atomic, so tag == seq implies the value is this step's; there is no flag, no fence and no barrier in the
// exchange, and its latency is one store propagation plus one load.  Slots alternate by step parity exactly like the
// kernel-boundary protocol (a word is next overwritten at step s+2, which its writer reaches only after every peer has
// produced step s+1, i.e. finished consuming step s), tags never repeat, every poll is bounded by the wall clock.
// rcn_hip_dp_init gives this form its own known-answer exchange (k_p2p_ll_selftest: same primitives) and its own vote.
using u64 = unsigned long long;
template <typename T> struct LLWords;
template <> struct LLWords<float> { static constexpr int n = 1; };
template <> struct LLWords<double> { static constexpr int n = 2; };

template <typename T>
__device__ inline u64* ll_region(void* rank_buf, size_t stride) { return reinterpret_cast<u64*>(reinterpret_cast<T*>(rank_buf) + 2 * stride); }   // behind the plain slots

__device__ inline void ll_store(u64* words, size_t idx, float v, unsigned seq) {
    unsigned b; __builtin_memcpy(&b, &v, 4);
    __hip_atomic_store(words + idx, ((u64)seq << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void ll_store(u64* words, size_t idx, double v, unsigned seq) {
    u64 b; __builtin_memcpy(&b, &v, 8);
    __hip_atomic_store(words + 2 * idx, ((u64)seq << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(words + 2 * idx + 1, ((u64)seq << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline bool ll_poll(const u64* words, size_t idx, unsigned seq, float& out) {
    const u64 w = __hip_atomic_load(words + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned b = (unsigned)w;
    __builtin_memcpy(&out, &b, 4);
    return (unsigned)(w >> 32) == seq;
}
__device__ inline bool ll_poll(const u64* words, size_t idx, unsigned seq, double& out) {
    const u64 lo = __hip_atomic_load(words + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const u64 hi = __hip_atomic_load(words + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const u64 b = (lo & 0xffffffffull) | (hi << 32);
    __builtin_memcpy(&out, &b, 8);
    return (unsigned)(lo >> 32) == seq && (unsigned)(hi >> 32) == seq;
}

// value `idx` of slot `seq & 1` summed over the ranks in rank order: own from the register, the peers' polled out of their
// HBM.  false: a peer's word did not arrive within the timeout.
// raw tagged word(s) of value idx: issued without looking at the answer, so that a round's loads are all in flight before the first use
template <typename T> struct LLRaw;
template <> struct LLRaw<float> { u64 w; };
template <> struct LLRaw<double> { u64 lo, hi; };
__device__ inline void ll_load(const u64* words, size_t idx, LLRaw<float>& r) { r.w = __hip_atomic_load(words + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void ll_load(const u64* words, size_t idx, LLRaw<double>& r) {
    r.lo = __hip_atomic_load(words + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    r.hi = __hip_atomic_load(words + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline bool ll_take(const LLRaw<float>& r, unsigned seq, float& out) {
    const unsigned b = (unsigned)r.w;
    __builtin_memcpy(&out, &b, 4);
    return (unsigned)(r.w >> 32) == seq;
}
__device__ inline bool ll_take(const LLRaw<double>& r, unsigned seq, double& out) {
    const u64 b = (r.lo & 0xffffffffull) | (r.hi << 32);
    __builtin_memcpy(&out, &b, 8);
    return (unsigned)(r.lo >> 32) == seq && (unsigned)(r.hi >> 32) == seq;
}

template <typename T>
__device__ inline bool ll_gather_sum(const P2PDesc& d, size_t stride, size_t idx, unsigned seq, T own, long long timeout_ticks, T& sum) {
    T v[kP2PMaxWorld];
    unsigned ready = 1u << d.rank;
    const unsigned all = (1u << d.world) - 1u;
    const size_t at = (size_t)(seq & 1u) * stride + idx;
    // A round has two phases: first the loads of every peer still missing, then the answers.  Written as one loop -- load a word,
    // look at it -- the compiler put an s_waitcnt vmcnt(0) behind every single load: up to seven xGMI round trips in a row per round
    // where one is enough (read off the ISA; invisible at a group of one).  Words that have arrived are not read again: every thread
    // of every rank polls, and re-reading everything every round saturates the memory of four ranks that share one GPU (measured:
    // time-outs) and would waste most of the link otherwise.
    // The clock is read only while a word is still missing, and then every 32nd round: s_memrealtime is a scalar memory access of the
    // better part of a microsecond -- read up front it sat between the publish and the first poll of EVERY exchange, and once per
    // round it would halve the polling rate.
    long long t0 = 0;
    for (unsigned it = 0; ready != all; ++it) {
        LLRaw<T> raw[kP2PMaxWorld];
        const unsigned miss = all & ~ready;
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q)                       // phase 1: the loads of the peers still missing, nothing looked at
            if ((miss >> q) & 1u) ll_load(ll_region<T>(d.buf[q], stride), at, raw[q]);
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q)                       // phase 2: the answers
            if ((miss >> q) & 1u) {
                T x;
                if (ll_take(raw[q], seq, x)) { v[q] = x; ready |= 1u << q; }
            }
        if (ready != all && (it & 31u) == 31u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout_ticks) return false;
        }
    }
    T g = 0;
#pragma unroll
    for (int q = 0; q < kP2PMaxWorld; ++q) g += q < d.world ? (q == d.rank ? own : v[q]) : (T)0;
    sum = g;
    return true;
}

// Four consecutive values idx0 .. idx0 + 3 per lane (the resident kernel's accumulator layout: dense_xcd.hpp) gathered TOGETHER: every round polls every
// outstanding (value, peer) pair, so the up-to 4 * (world - 1) xGMI reads of a lane are in flight at once and the exchange costs one
// peer round trip, not four.  want[i] false: value i is not exchanged (its sum is left alone).  Sums in rank order, as above.
__device__ inline bool ll_gather_sum4(const P2PDesc& d, size_t stride, size_t idx0, const bool (&want)[4], unsigned seq, float (&own)[4],
                                      long long timeout_ticks) {
    const size_t slot = (size_t)(seq & 1u) * stride + idx0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long long t0 = 0;                                                 // (read lazily: see ll_gather_sum)
    unsigned it = 0;
    const unsigned peers = ((1u << d.world) - 1u) & ~(1u << d.rank);
    // Two batches of four ranks, in rank order: up to sixteen words in flight per round (phase 1: the loads of the words still missing;
    // phase 2: the answers -- see ll_gather_sum); a batch is added, in rank order and with this rank's own values at its own place,
    // once all of its words carry this step's tag.  Every rank publishes at about the same time, so the second batch is normally
    // complete one round trip after the first.
#pragma unroll
    for (int q0 = 0; q0 < kP2PMaxWorld; q0 += 4) {
        if (q0 >= d.world) break;                                     // (uniform)
        float x[4][4];
        unsigned miss[4];                                             // per value: the ranks of this batch whose word is still missing
#pragma unroll
        for (int i = 0; i < 4; ++i) miss[i] = want[i] ? (peers >> q0) & 15u : 0u;
        for (; (miss[0] | miss[1] | miss[2] | miss[3]) != 0u; ++it) {
            u64 w[4][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const u64* words = ll_region<float>(d.buf[(q0 + qq) < d.world ? q0 + qq : 0], stride) + slot;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((miss[i] >> qq) & 1u) w[i][qq] = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (((miss[i] >> qq) & 1u) && (unsigned)(w[i][qq] >> 32) == seq) {
                        const unsigned b = (unsigned)w[i][qq];
                        __builtin_memcpy(&x[i][qq], &b, 4);
                        miss[i] &= ~(1u << qq);
                    }
            if ((miss[0] | miss[1] | miss[2] | miss[3]) != 0u && (it & 31u) == 31u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > timeout_ticks) return false;
            }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = q0 + qq;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += q < d.world ? (q == d.rank ? own[i] : x[i][qq]) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (want[i]) own[i] = acc[i];
    return true;
}

template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_p2_dp_fused(
    NetDesc nd, T* __restrict__ params, const T* __restrict__ Xp, const T* __restrict__ Xn, int B, const T* __restrict__ a1,
    const T* __restrict__ d1, const T* __restrict__ d2, T scale, T* __restrict__ slab, int G, const T* __restrict__ loss_part, int n_loss,
    T loss_scale, T* __restrict__ loss_out, int do_fwd, P2PDesc d, const unsigned* __restrict__ seq_base, unsigned seq_off, size_t stride,
    unsigned* __restrict__ err, long long timeout_ticks, T* __restrict__ tail_scratch, T* __restrict__ fragimg) {
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    int& s_bad = *reinterpret_cast<int*>(red + kDenseWaves * kMtp * kRedTile + 16 * kP2H);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1];
    const unsigned seq = (seq_base ? *seq_base : 0u) + seq_off;
    u64* __restrict__ mine = ll_region<T>(d.buf[d.rank], stride) + (size_t)(seq & 1u) * stride * LLWords<T>::n;
    if (tid == 0) s_bad = *err != 0u ? 2 : 0;
    __syncthreads();
    if (s_bad == 2) return;

    if ((int)blockIdx.x >= G) {
        // tail tile e: db_0 (e == 0) or 16 columns of [W_1 | b_1]; its parameters are one contiguous run.  The tile's sums go
        // through a local scratch image (wgrad_tile_ld ends on a barrier), from where each thread publishes and consumes its own
        const int e = (int)blockIdx.x - G;
        const int j = e == 0 ? 0 : 1, n0 = e == 0 ? F : (e - 1) * 16;
        if (e == 0) wgrad_tile_ld<T, false>(nd, 0, F, (T*)nullptr, tail_scratch, (const T*)nullptr, 0, (const int*)nullptr, d1, kP2H, B, (T)0, red);
        else        wgrad_tile_ld<T, false>(nd, 1, n0, (T*)nullptr, tail_scratch, a1, kP2H, (const int*)nullptr, d2, kP2C, B, (T)0, red);
        const int Kin = nd.dims[j], M = nd.dims[j + 1];
        const int c1 = n0 + 16 < Kin + 1 ? n0 + 16 : Kin + 1;                    // columns n0 .. c1-1 of [W_j | b_j] (bias = column Kin)
        const int p0 = nd.w_off[j] + n0 * M, cnt = (c1 - n0) * M;
        bool ok = true;
        for (int i = tid; i < cnt; i += kDenseThreads) ll_store(mine, (size_t)(p0 + i), tail_scratch[p0 + i], seq);
        T own_loss = 0;
        if (e == 0 && tid == 0) { finish_loss<T>(loss_part, n_loss, loss_scale, &own_loss); ll_store(mine, (size_t)nd.P, own_loss, seq); }
        for (int i = tid; i < cnt; i += kDenseThreads) {
            T g;
            if (ll_gather_sum<T>(d, stride, (size_t)(p0 + i), seq, tail_scratch[p0 + i], timeout_ticks, g)) {
                const T nv = params[p0 + i] - scale * g;                                                  // rcn.rs:214,221
                params[p0 + i] = nv;
                if (fragimg) p2_frag_scatter(j, n0 + i / M, i % M, H, nv, fragimg);                        // k_p2_b's operand image (dense.hpp)
            } else ok = false;
        }
        if (e == 0 && tid == 0) {
            T g;
            if (ll_gather_sum<T>(d, stride, (size_t)nd.P, seq, own_loss, timeout_ticks, g)) { if (loss_out) *loss_out = g; }
            else ok = false;
        }
        if (!ok) *err = 1u + (unsigned)d.rank;
        return;
    }

    const int f0 = (int)blockIdx.x * 16;
    const int nf = F - f0 < 16 ? F - f0 : 16;
    T* wsl = red + kDenseWaves * kMtp * kRedTile;
    T* W0 = params + nd.w_off[0];
    const T* __restrict__ cp = Xp + (size_t)blockIdx.x * B * 16;
    const T* __restrict__ cn = Xn + (size_t)blockIdx.x * B * 16;
    const int ntile = B >> 4;
    vec4 xn[2];
    if (do_fwd) {
#pragma unroll
        for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (wave + 8 * u) + n) * 16 + 4 * g4);
    }
    {
        // ---- U: this shard's dW_0[:, slice]                                                      rcn.rs:310
        const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8;
        const int m = mt * 16 + ml;
        const bool wvalid = m < H && cl < nf;
        const size_t off = (size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0);
        const T wold = W0[off];
        acc_t acc[kMtp];
#pragma unroll
        for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
        const int kw = B >> 3;
        for (int kc = wave * kw; kc < (wave + 1) * kw; kc += 32) {
            const T* xb = cp + (size_t)(kc + g4) * 16 + n;
            const T* db = d1 + (size_t)(kc + g4) * kP2H + n;
            T bv[8], av[8][kMtp];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bv[q] = xb[q * 64];
#pragma unroll
                for (int t = 0; t < kMtp; ++t) av[q][t] = db[q * 4 * kP2H + t * 16];
            }
            __builtin_amdgcn_sched_barrier(0);           // every load in flight before the first MFMA waits on one (dense_p2.hpp)
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(av[q][t], bv[q], acc[t]);
        }
        store_partials<T>(red, wave, lane, acc);
        __syncthreads();
        const T gsum = sum_partials<T>(red, mt, cl, ml);
        // ---- publish, gather, W_0 <- W_0 - (eta / B_global) * sum over ranks                        rcn.rs:214
        T g = gsum;
        bool ok = true;
        if (wvalid) {
            ll_store(mine, (size_t)nd.w_off[0] + off, gsum, seq);
            ok = ll_gather_sum<T>(d, stride, (size_t)nd.w_off[0] + off, seq, gsum, timeout_ticks, g);
        }
        if (!ok) { s_bad = 1; *err = 1u + (unsigned)d.rank; }
        __syncthreads();
        if (s_bad) return;
        const T w = wold - scale * g;
        if (wvalid) W0[off] = w;
        wsl[cl * kP2H + m] = wvalid ? w : (T)0;
    }
    __syncthreads();
    if (!do_fwd) return;
    // ---- F: partial z_1 of the new batch (k_p2_a's F half)
    T wf[4][kMtp];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < kMtp; ++t) wf[i][t] = wsl[(4 * g4 + i) * kP2H + t * 16 + n];
    for (int tb = 0; tb < ntile; tb += 16) {
        if (tb > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (tb + wave + 8 * u) + n) * 16 + 4 * g4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = 16 * (tb + wave + 8 * u) + n;
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(wf[i][t], xn[u][i], acc[t]);
            T* dst = slab + (((size_t)(s >> 3) * G + blockIdx.x) * kP2Ts + (s & 7)) * kP2H;
#pragma unroll
            for (int t = 0; t < kMtp; ++t) store4<T>(dst + t * 16, lane, acc[t]);
        }
    }
}

// known-answer exchange on exactly these primitives: every thread publishes one pattern value and gathers the sum
template <typename T>
__global__ __launch_bounds__(256) void k_p2p_ll_selftest(P2PDesc d, unsigned seq, size_t stride, unsigned* __restrict__ err,
                                                         long long timeout_ticks, unsigned* __restrict__ mismatches) {
    if (*err != 0u) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= stride) return;
    const T own = (T)p2p_pattern(d.rank, seq, i);
    ll_store(ll_region<T>(d.buf[d.rank], stride) + (size_t)(seq & 1u) * stride * LLWords<T>::n, i, own, seq);
    T got;
    if (!ll_gather_sum<T>(d, stride, i, seq, own, timeout_ticks, got)) { *err = 1u + (unsigned)d.rank; return; }
    T want = 0;
    for (int q = 0; q < d.world; ++q) want += (T)p2p_pattern(q, seq, i);
    if (got != want) atomicAdd(mismatches, 1u);
}

__global__ void k_set_u32(unsigned* p, unsigned v) { *p = v; }

}  // namespace rcn
