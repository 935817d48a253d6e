// dense_p2_persist.hpp -- the feature-sliced pipeline (dense_p2.hpp) as ONE kernel per epoch segment.
//
// k_p2_b / k_p2_a are bound by what surrounds their arithmetic: two dependent launches per step (1.6 us each inside a
// hipGraph) and W_0 leaving and re-entering the chip every step.  Here the same three kinds of workgroup stay resident for
// the whole segment and hand their results to each other through global memory, with the two in-launch hand-off forms of
// cdna_hip_programming.md (Guideline 16): big payloads (slab partials, deltas, activations) are stored write-through as
// plain 16-byte runs (buffer stores, sc1) and announced by ONE flag word per producer after its waves drained their stores,
// the consumer polls that word and then reads with sc1 loads; the small tail parameters travel as self-validating 8-byte
// words {value, step} that are their own flag.  (A first version sent everything as tagged words: correct, but 401 K
// uncoalesced 8-byte write-through stores per step queued up for 3-10 us before becoming visible.)  No launch, no fence
// and no grid barrier between steps; a slice of W_0 never leaves its workgroup's registers/LDS until the segment ends.
//
//   feature workgroup g (G of them)   owns W_0[:, 16g..16g+15].  Per step: poll delta_1 of the whole batch -> dW_0 slice
//                                     (MFMA, K-split over 8 waves) -> update in place -> partial z_1 of the NEXT batch
//                                     from the updated slice -> slab words.
//   sample workgroup t (B/8)          owns samples 8t..8t+7.  Per step: poll the G slab partials and the tail parameters ->
//                                     a_1, layer 2, loss part, delta_2, delta_1 -> words for the feature and tail groups.
//   tail workgroup e (1 + tiles)      owns db_0 or 16 columns of [W_1 | b_1].  Per step: poll a_1 / delta_2 / delta_1 of the
//                                     batch -> gradient tile -> update in registers -> publish the parameters as words.
//
// Every buffer has two halves selected by tag parity.  A word is rewritten two steps after it was written, and the chain
// slab(j) -> delta(j) -> slab(j+1) -> ... guarantees its last reader is done by then (dense_p2_dp.hpp spells the argument
// out).  Tags grow monotonically over the life of the context and never repeat.  All workgroups must be resident at once
// (85 for the default net on 256 CUs); every poll is bounded by the wall clock and raises a sticky error word that makes
// all workgroups leave at their next step, so a co-tenant that keeps some of them off the chip costs time, not a hang.
// f32 only (one value + tag = one naturally atomic 8-byte word); the f64 context keeps the two-kernel pipeline.
#pragma once

#include "dense_p2.hpp"

namespace rcn {

// diagnostic build only (-DRCN_STAMPS): phase times of the last-but-one step of a launch, per workgroup
#define PSTAMP(i) do { if (j == nb - 2) RCN_STAMP(0, i); } while (0)
#ifdef RCN_STAMPS
#define PSTAMP_W(i) do { if (j == nb - 2 && lane == 0 && blockIdx.x < 512) g_rcn_stamps[0][blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP_W(i) do { } while (0)
#endif

using pw_t = unsigned long long;

#ifndef RCN_PW_SLEEP
#define RCN_PW_SLEEP 8          // s_sleep units (64 clocks each) between two rounds of a poll: fewer rounds, less pressure on the memory side
#endif

__device__ inline void pw_store(pw_t* w, float v, unsigned tag) {
    unsigned b; __builtin_memcpy(&b, &v, 4);
    __hip_atomic_store(w, ((pw_t)tag << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline pw_t pw_load(const pw_t* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float pw_value(pw_t w) { const unsigned b = (unsigned)w; float v; __builtin_memcpy(&v, &b, 4); return v; }

// N words at p[i * stride], all polled until every tag matches; false after `timeout` ticks of the 100 MHz clock
// Bounded wait shared by the polls: looked at only every 1024 failed rounds, so the common case pays nothing for it.
// Gives up when the deadline passed or when another workgroup already raised the sticky error word.
__device__ inline bool pw_give_up(long long& t0, long long timeout, const unsigned* err) {
    const long long now = wall_clock64();
    if (t0 == 0) { t0 = now; return false; }
    return now - t0 > timeout || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

template <int N>
__device__ inline bool pw_poll(const pw_t* p, size_t stride, unsigned tag, float (&out)[N], long long timeout, const unsigned* err) {
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        pw_t w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = pw_load(p + i * stride);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && (unsigned)(w[i] >> 32) == tag;
        if (ok) {
#pragma unroll
            for (int i = 0; i < N; ++i) out[i] = pw_value(w[i]);
            return true;
        }
        if ((it & 1023u) == 1023u && pw_give_up(t0, timeout, err)) return false;
        __builtin_amdgcn_s_sleep(RCN_PW_SLEEP);
    }
}

// the same for N words at arbitrary offsets from one base: ONE round of loads in flight for all of them
template <int N>
__device__ inline bool pw_poll_at(const pw_t* base, const int (&off)[N], unsigned tag, float (&out)[N], long long timeout, const unsigned* err) {
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        pw_t w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = pw_load(base + off[i]);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && (unsigned)(w[i] >> 32) == tag;
        if (ok) {
#pragma unroll
            for (int i = 0; i < N; ++i) out[i] = pw_value(w[i]);
            return true;
        }
        if ((it & 1023u) == 1023u && pw_give_up(t0, timeout, err)) return false;
        __builtin_amdgcn_s_sleep(RCN_PW_SLEEP);
    }
}

struct PersistBufs {
    float* slab;       // [2][B/8][G][8][32]   partial z_1, written by the feature groups
    float* d1;         // [2][B][32]           written by the sample groups (as are a1, d2, loss)
    float* a1;         // [2][B][32]
    float* d2;         // [2][B][16]
    float* loss;       // [2][B/8]
    unsigned* sflag;   // [G]     step tag of the newest complete slab of feature group g
    unsigned* oflag;   // [B/8]   step tag of the newest complete outputs of sample group t
    pw_t* tail;        // [2][512] tail parameters [b_0 | W_1 | b_1] as tagged words
};

constexpr int kPersistThreads = 512, kPersistTailPad = 512;
// bytes: plain arrays, then the flags (padded), then the tagged words
inline size_t persist_bytes(size_t B, size_t G) {
    const size_t NS = B / kP2Ts;
    return 2 * (NS * G * kP2Ts * kP2H + B * kP2H + B * kP2H + B * kP2C + NS) * sizeof(float) + (64 + NS + 64) * sizeof(unsigned) +
           2 * kPersistTailPad * sizeof(pw_t) + 256;
}
inline int persist_grid(const NetDesc& nd, size_t B) { return pipe_slices(nd) + (int)(B / kP2Ts) + pipe_extra_wgs(nd); }
inline bool persist_supported(const NetDesc& nd, size_t B) {
    return p2_supported(nd, B) && nd.P - (nd.w_off[0] + nd.dims[0] * nd.dims[1]) <= kPersistTailPad && persist_grid(nd, B) <= 224;
}

using pu4 = __attribute__((ext_vector_type(4))) unsigned;
using pu2 = __attribute__((ext_vector_type(2))) unsigned;
// write-through / cache-bypassing accesses to the exchange arrays (aux 16 = sc1): byte offsets from the array's base
#define PW_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((ptr), 0, (int)(bytes), 0x00020000)
__device__ inline void px_store4(__amdgpu_buffer_rsrc_t r, int byte_off, float a, float b, float c, float d) {
    const float f[4] = {a, b, c, d};
    pu4 v;
    __builtin_memcpy(&v, f, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}
__device__ inline void px_store1(__amdgpu_buffer_rsrc_t r, int byte_off, float a) {
    unsigned v; __builtin_memcpy(&v, &a, 4);
    __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, 16);
}
__device__ inline float px_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16);
    float f; __builtin_memcpy(&f, &v, 4); return f;
}
// every storing wave drains its write-through stores before the flag that announces them is written
__device__ inline void px_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// one wave waits until the flag words its lanes look at (lanes >= nflags look at nothing) all carry `tag`
__device__ inline bool px_wait_flags(const unsigned* flags, int idx, bool active, unsigned tag, long long timeout, const unsigned* err) {
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        const unsigned f = active ? __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
        if (__all((int)(f - tag) >= 0)) return true;
        if ((it & 1023u) == 1023u && pw_give_up(t0, timeout, err)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(kPersistThreads) void k_p2_epoch(
    NetDesc nd, float* __restrict__ params, const float* __restrict__ xpack, const float* __restrict__ ypack, int B, int nb, int G, float scale,
    float loss_scale, float* __restrict__ loss_dev, PersistBufs bufs, unsigned base, unsigned* __restrict__ err, long long timeout) {
    using T = float;
    using acc_t = Mfma16<T>::acc_t;
    using vec4 = Vec4<T>::type;
    __shared__ __attribute__((aligned(16))) float smem[kDenseWaves * kMtp * kRedTile + 16 * kP2H + 64 + 512];
    __shared__ int s_abort;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1], C = nd.dims[2];
    const int NS = B / kP2Ts;
    const int tail_begin = nd.w_off[0] + F * H;
    const int slab_half = NS * G * kP2Ts * kP2H, dh_half = B * kP2H, d2_half = B * kP2C;      // values per parity half
    const auto r_slab = PW_RSRC(bufs.slab, (size_t)2 * slab_half * 4);
    const auto r_d1 = PW_RSRC(bufs.d1, (size_t)2 * dh_half * 4), r_a1 = PW_RSRC(bufs.a1, (size_t)2 * dh_half * 4);
    const auto r_d2 = PW_RSRC(bufs.d2, (size_t)2 * d2_half * 4), r_loss = PW_RSRC(bufs.loss, (size_t)2 * NS * 4);
    bool bad = false;
    // A failed wait raises the sticky error word (which makes every other workgroup's next long wait give up too) and this
    // workgroup's LDS flag; a workgroup looks at its flag right after a barrier it has anyway, and leaves.
    if (tid == 0) s_abort = 0;
    __syncthreads();
    auto step_begin = [&]() -> bool { return s_abort == 0; };
    auto fail = [&]() { bad = true; s_abort = 1; __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    if ((int)blockIdx.x < G) {
        // =============================================================== feature workgroup: slice g of W_0
        const int g = blockIdx.x, f0 = g * 16;
        const int nf = F - f0 < 16 ? F - f0 : 16;
        float* red = smem;
        float* wsl = smem + kDenseWaves * kMtp * kRedTile;
        float* W0 = params + nd.w_off[0];
        const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8, m = mt * 16 + ml;
        const bool wvalid = m < H && cl < nf;
        const size_t off = (size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0);
        float w = W0[off];
        w = wvalid ? w : 0.f;
        wsl[cl * kP2H + m] = w;
        __syncthreads();
        const int ntile = B >> 4, kw = B >> 3;
        // partial z_1 of batch jn from the slice in LDS -> slab words tagged `tag`
        auto forward = [&](int jn, unsigned tag) {
            const float* __restrict__ cn = xpack + ((size_t)jn * G + g) * B * 16;
            const int sl = (int)(tag & 1u) * slab_half;
            float wf[4][kMtp];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) wf[i][t] = wsl[(4 * g4 + i) * kP2H + t * 16 + n];
            for (int tb = 0; tb < ntile; tb += 16) {
                vec4 xn[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (tb + wave + 8 * u) + n) * 16 + 4 * g4);
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
                    // rows 4 g4 .. 4 g4 + 3 of each 16-row tile: 16 contiguous bytes per lane, 64 per sample and tile
                    const int dst = sl + ((((s >> 3) * G + g) * kP2Ts + (s & 7)) * kP2H + 4 * g4);
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) px_store4(r_slab, (dst + t * 16) * 4, acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                }
            }
            px_drain();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(bufs.sflag + g, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        forward(0, base + 1);
        for (int j = 0; j < nb; ++j) {
            if (!step_begin()) return;
            PSTAMP(0);
            const unsigned tag = base + (unsigned)j + 1;
            const float* __restrict__ cp = xpack + ((size_t)j * G + g) * B * 16;
            const int dl = (int)(tag & 1u) * dh_half;
            // ---- U: dW_0[:, slice] = sum_s delta_1[s] (x) x_s[slice]                                   rcn.rs:310
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
            for (int kc = wave * kw; kc < (wave + 1) * kw; kc += 32) {
                const float* xb = cp + (size_t)(kc + g4) * 16 + n;
                float bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) bv[q] = xb[q * 64];
                // samples kc .. kc + 31 belong to sample groups kc/8 .. kc/8 + 3: wait for their flags, then read delta_1
                if (!px_wait_flags(bufs.oflag, kc / kP2Ts + (lane & 3), lane < 4, tag, timeout, err)) fail();
                PSTAMP(1);
                float av[16];                                           // delta_1 rows n (0..7) and 16 + n (8..15) of samples kc + g4 + 4 q
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int at = dl + (kc + g4 + 4 * q) * kP2H + n;
                    av[q] = px_load1(r_d1, at * 4);
                    av[8 + q] = px_load1(r_d1, (at + 16) * 4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc[0] = Mfma16<T>::mfma(av[q], bv[q], acc[0]);
                    acc[1] = Mfma16<T>::mfma(av[8 + q], bv[q], acc[1]);
                }
            }
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            PSTAMP(2);
            w = w - scale * sum_partials<T>(red, mt, cl, ml);            // rcn.rs:214
            w = wvalid ? w : 0.f;
            wsl[cl * kP2H + m] = w;
            __syncthreads();
            PSTAMP(3);
            if (j + 1 < nb) forward(j + 1, tag + 1);
            PSTAMP(4);
        }
        if (wvalid && !bad) W0[off] = w;
        return;
    }

    if ((int)blockIdx.x < G + NS) {
        // =============================================================== sample workgroup: samples 8 t .. 8 t + 7
        constexpr int kFrag = 28;                                     // fragment words per lane: wz 8, wd 8, b1 4, y 4, b0 4
        constexpr int kPer = kP2MaxSlices / kP2BWaves;
        const int t = blockIdx.x - G, s0 = t * kP2Ts;
        vec4* zred = reinterpret_cast<vec4*>(smem);
        float* a1s = smem + kP2BWaves * 64 * 4;
        float* d2s = a1s + kP2H * kLd;
        float* frag = d2s + kP2C * kLd;
        const int w1_t = nd.w_off[1] - tail_begin;                    // offsets inside the tail parameter block
        const int b1_t = w1_t + C * H, b0_t = 0;
        for (int j = 0; j < nb; ++j) {
            if (!step_begin()) return;
            PSTAMP(0);
            const unsigned tag = base + (unsigned)j + 1;
            const pw_t* tl = bufs.tail + (size_t)(tag & 1u) * kPersistTailPad;
            // the tail's operands as ready-made MFMA fragments: each lane of waves 1-5 and 7 needs four tail parameters -- from
            // plain memory before the first update, afterwards from the words the tail workgroups published (one round of loads)
            int ti[4];
            bool tv[4];
            int fw = -1;                                              // first fragment word this lane fills (wave-uniform), -1: none
            if (wave == 1 || wave == 2) {                             // z_2 = W_1 a_1:   A[m = c][k = h]
                fw = 4 * (wave - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = 4 * (fw + q) + g4;
                    ti[q] = w1_t + (h < H ? h : 0) * C + (n < C ? n : 0);
                    tv[q] = h < H && n < C;
                }
            } else if (wave == 3 || wave == 4) {                      // W_1^T delta_2:  A[m = h][k = c]
                const int mt = wave - 3;
                fw = 8 + mt * 4;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int h = mt * 16 + n, c = 4 * ks + g4;
                    ti[ks] = w1_t + (h < H ? h : 0) * C + (c < C ? c : 0);
                    tv[ks] = h < H && c < C;
                }
            } else if (wave == 5) {                                   // b_1 per accumulator element
                fw = 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = Mfma16<T>::row(lane, i);
                    ti[i] = b1_t + (c < C ? c : 0);
                    tv[i] = c < C;
                }
            } else if (wave == 7) {                                   // b_0 per slab float4 element
                fw = 24;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int h = 4 * (lane & 7) + i;
                    ti[i] = b0_t + (h < H ? h : 0);
                    tv[i] = h < H;
                }
            } else if (wave == 6) {                                   // targets per accumulator element (plain memory)
                const float* Ys = ypack + (size_t)j * B * C;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = Mfma16<T>::row(lane, i);
                    frag[(20 + i) * 64 + lane] = Ys[(size_t)(s0 + (n & 7)) * C + (c < C ? c : 0)];
                }
            }
            if (fw >= 0) {
                float tp[4];
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) tp[i] = params[tail_begin + ti[i]];
                } else if (!pw_poll_at<4>(tl, ti, tag, tp, timeout, err)) fail();
#pragma unroll
                for (int i = 0; i < 4; ++i) frag[(fw + i) * 64 + lane] = tv[i] ? tp[i] : 0.f;
            }
            if (wave == 7) PSTAMP_W(1);
            // ---- slab partials: slices wave, wave + 8, ...; lane <- elements 4 lane .. 4 lane + 3 of the [8][32] tile
            // this wave's slices are wave, wave + 8, ...: wait for their producers' flags (lane q looks at slice wave + 8 q), then
            // every lane reads its 16 bytes of each [8][32] tile
            if (!px_wait_flags(bufs.sflag, wave + kP2BWaves * (lane & 7), lane < kPer && wave + kP2BWaves * lane < G, tag, timeout, err)) fail();
            const int sp = ((int)(tag & 1u) * slab_half + t * G * (kP2Ts * kP2H) + 4 * lane) * 4;
            vec4 z = vec4{0, 0, 0, 0};
            pu4 raw[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int gq = wave + kP2BWaves * q;
                raw[q] = __builtin_amdgcn_raw_buffer_load_b128(r_slab, sp + (gq < G ? gq : wave) * (kP2Ts * kP2H * 4), 0, 16);
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q)
                if (wave + kP2BWaves * q < G) {
                    vec4 v;
                    __builtin_memcpy(&v, &raw[q], 16);
                    z += v;
                }
            zred[wave * 64 + lane] = z;
            if (wave == 0) PSTAMP(2);
            __syncthreads();
            PSTAMP(3);
            if (wave == 0) {
                z = ((zred[lane] + zred[64 + lane]) + (zred[128 + lane] + zred[192 + lane])) +
                    ((zred[256 + lane] + zred[320 + lane]) + (zred[384 + lane] + zred[448 + lane]));
                {   // a_1 = sigmoid(z_1 + b_0); lane <- sample lane>>3, hidden 4*(lane&7)+i                rcn.rs:287-289
                    const int s = lane >> 3, h0 = 4 * (lane & 7);
                    float a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto sg = sigmoid_fast(z[i] + frag[(24 + i) * 64 + lane]);   // unconditionally: as `c ? f(x) : 0` this is four branches
                        a[i] = (h0 + i < H) ? sg : 0.f;
                        a1s[(h0 + i) * kLd + s] = a[i];
                    }
                    px_store4(r_a1, ((int)(tag & 1u) * dh_half + (s0 + s) * kP2H + h0) * 4, a[0], a[1], a[2], a[3]);
                }
                acc_t acc = acc_t{0, 0, 0, 0};                        // z_2 = W_1 a_1 + b_1
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const float bv = a1s[(4 * ks + g4) * kLd + (n & 7)];
                    acc = Mfma16<T>::mfma(frag[ks * 64 + lane], n < kP2Ts ? bv : 0.f, acc);
                }
                float lsum = 0.f;
                acc_t dv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = Mfma16<T>::row(lane, i);
                    const float a2 = sigmoid_fast(acc[i] + frag[(16 + i) * 64 + lane]);
                    const float diff = a2 - frag[(20 + i) * 64 + lane];
                    const bool ok = c < C && n < kP2Ts;
                    dv[i] = ok ? diff * (a2 * (1.f - a2)) : 0.f;      // rcn.rs:299
                    lsum += ok ? diff * diff : 0.f;
                    d2s[c * kLd + n] = dv[i];
                }
                if (n < kP2Ts) px_store4(r_d2, ((int)(tag & 1u) * d2_half + (s0 + n) * kP2C + 4 * g4) * 4, dv[0], dv[1], dv[2], dv[3]);
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt) {                   // delta_1 = (W_1^T delta_2) (*) a_1 (1 - a_1)   rcn.rs:305-309
                    acc_t ad = acc_t{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) ad = Mfma16<T>::mfma(frag[(8 + mt * 4 + ks) * 64 + lane], d2s[(4 * ks + g4) * kLd + n], ad);
                    float o4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float a = a1s[(mt * 16 + Mfma16<T>::row(lane, i)) * kLd + (n & 7)];
                        o4[i] = ad[i] * (a * (1.f - a));                  // padded hidden rows hold a = 0 -> delta 0
                    }
                    if (n < kP2Ts) px_store4(r_d1, ((int)(tag & 1u) * dh_half + (s0 + n) * kP2H + mt * 16 + 4 * g4) * 4, o4[0], o4[1], o4[2], o4[3]);
                }
                lsum = wave_sum_lane0(lsum);
                if (lane == 0) px_store1(r_loss, ((int)(tag & 1u) * NS + t) * 4, lsum);
                px_drain();                                           // only this wave stored: drain, then announce
                if (lane == 0) __hip_atomic_store(bufs.oflag + t, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PSTAMP(4);
            }
            __syncthreads();                                          // wave 0 is done with frag / zred before they are refilled
        }
        return;
    }

    {
        // =============================================================== tail workgroup e: db_0 or 16 columns of [W_1 | b_1]
        const int e = blockIdx.x - G - NS;
        const int jl = e == 0 ? 0 : 1, n0 = e == 0 ? F : (e - 1) * 16;
        const int Kin = nd.dims[jl], M = nd.dims[jl + 1];
        float* red = smem;
        const int c = n0 + n;                                          // this lane's column of [W | b]
        const int mt_u = tid >> 8, o_u = tid & 255, cl_u = o_u >> 4, ml_u = o_u & 15;
        const int m_u = mt_u * 16 + ml_u, cc_u = n0 + cl_u;
        const bool pvalid = m_u < M && cc_u <= Kin;
        const int p_u = nd.w_off[jl] + (pvalid ? cc_u * M + m_u : 0);
        float pv = params[p_u];
        const int kw = B >> 3;                                         // samples per wave (a multiple of 32)
        for (int j = 0; j < nb; ++j) {
            if (!step_begin()) return;
            PSTAMP(0);
            const unsigned tag = base + (unsigned)j + 1;
            const int d_off = (int)(tag & 1u) * (e == 0 ? dh_half : d2_half), a_off = (int)(tag & 1u) * dh_half;
            const int ldD = e == 0 ? kP2H : kP2C;
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
            for (int kc = wave * kw; kc < (wave + 1) * kw; kc += 32) {
                if (!px_wait_flags(bufs.oflag, kc / kP2Ts + (lane & 3), lane < 4, tag, timeout, err)) fail();
                PSTAMP(1);
                // delta rows n and 16 + n, and (tiles of W_1) column c of a_1, of samples kc + g4 + 4 q
                const int r0 = n < M ? n : M - 1, r1 = 16 + n < M ? 16 + n : M - 1, ca = c < Kin ? c : Kin - 1;
                float v[24];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int smp = kc + g4 + 4 * q;
                    v[q] = px_load1(e == 0 ? r_d1 : r_d2, (d_off + smp * ldD + r0) * 4);
                    v[8 + q] = px_load1(e == 0 ? r_d1 : r_d2, (d_off + smp * ldD + r1) * 4);
                    v[16 + q] = e != 0 ? px_load1(r_a1, (a_off + smp * kP2H + ca) * 4) : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    // bias column: the activation is the constant 1, so db = sum_s delta              rcn.rs:302,309
                    const float b = e != 0 && c < Kin ? v[16 + q] : (c == Kin ? 1.f : 0.f);
                    acc[0] = Mfma16<T>::mfma(n < M ? v[q] : 0.f, b, acc[0]);
                    acc[1] = Mfma16<T>::mfma(16 + n < M ? v[8 + q] : 0.f, b, acc[1]);
                }
            }
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            PSTAMP(2);
            if (pvalid) {
                pv = pv - scale * sum_partials<T>(red, mt_u, cl_u, ml_u);                              // rcn.rs:214,221
                pw_store(bufs.tail + (size_t)((tag + 1) & 1u) * kPersistTailPad + (p_u - tail_begin), pv, tag + 1);
            }
            if (e == 0 && wave == 7 && loss_dev) {                     // cost of this batch: lane t fetches sample group t's part, lane 0 adds in order
                float tot = 0.f;
                for (int t0 = 0; t0 < NS; t0 += 64) {
                    const int t = t0 + lane;
                    if (!px_wait_flags(bufs.oflag, t < NS ? t : NS - 1, t < NS, tag, timeout, err)) fail();
                    const float part = px_load1(r_loss, ((int)(tag & 1u) * NS + (t < NS ? t : NS - 1)) * 4);
                    for (int u = 0; u < 64 && t0 + u < NS; ++u) tot += __shfl(part, u, 64);
                }
                if (lane == 0) loss_dev[j] = tot * loss_scale;
            }
            __syncthreads();
        }
        if (pvalid && !bad) params[p_u] = pv;
    }
}

}  // namespace rcn
