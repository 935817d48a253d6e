// dense_p2.hpp -- the feature-sliced pipeline of dense_pipe.hpp, specialised for the shape class of the reference's
// own network (rcn/src/main.rs:53-59: ONE hidden layer): hidden <= 32, classes <= 16, batch a multiple of 256.
//
// Same algorithm and the same packed slice-major batch image (k_pack_epoch); what changes is the instruction count.
// At this problem size a kernel runs at 1-2 waves per SIMD, so every instruction costs its full issue latency:
// the generic kernels spend ~2000 instructions per wave on runtime div/mod, 64-bit clamped addressing and scalar LDS
// loops.  Here every internal image has power-of-two padded rows (hidden -> 32, classes -> 16; pads are exact zeros),
// so index math is shifts and immediates, no load needs a clamp, the tail layer runs on MFMA from fragments that are
// loaded straight from global memory at kernel entry, and the slab sum is spread over all four waves.
//
//   internal images (device only; the parameter layout stays the reference's column-major W|b):
//     a1 [B][32]  hidden activations      d1 [B][32]  delta of the hidden layer      d2 [B][16]  output delta
//     slab [B/8][G][8][32]                partial z_1 per 16-feature slice, contiguous per k_p2_b workgroup
#pragma once

#include "common.hpp"
#include "dense.hpp"
#include "dense_pipe.hpp"

namespace rcn {

constexpr int kP2H = 32, kP2C = 16, kP2Ts = 8, kP2BThreads = 512, kP2BWaves = kP2BThreads / 64, kP2MaxSlices = 64;

inline bool p2_supported(const NetDesc& nd, size_t B) {
    return nd.L == 2 && nd.dims[1] <= kP2H && nd.dims[2] <= kP2C && B % 256 == 0 && B >= 256 && pipe_slices(nd) <= kP2MaxSlices;
}
inline size_t p2_a_lds_elems() { return (size_t)kDenseWaves * kMtp * kRedTile + 16 * kP2H + 64; }

template <typename T>
__device__ inline void store4(T* dst, int lane, const typename Mfma16<T>::acc_t& v) {
    // rows held by a lane: f32 4*(lane>>4)+i (contiguous -> one 16-byte store), f64 (lane>>4)+4i (strided)
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<typename Vec4<T>::type*>(dst + 4 * (lane >> 4)) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[(lane >> 4) + 4 * i] = v[i];
    }
}

template <typename T>
__device__ __forceinline__ void p2_a_body(
    const NetDesc& nd, T* __restrict__ params, const T* __restrict__ Xp, const T* __restrict__ Xn, int B,
    const T* __restrict__ a1, const T* __restrict__ d1, const T* __restrict__ d2, T scale, T* __restrict__ slab, int G,
    const T* __restrict__ loss_part, int n_loss, T loss_scale, T* __restrict__ loss_out, int do_update, int do_fwd, unsigned char* smem_raw,
    T* __restrict__ fragimg = nullptr) {
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    T* red = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1];
    RCN_STAMP(0, 0);

    if ((int)blockIdx.x >= G) {
        // e == 0: bias column of W_0 (db_0 = sum_s delta_1);  e >= 1: 16-column tiles of [W_1 | b_1]
        if (!do_update) return;
        const int e = (int)blockIdx.x - G;
        if (e == 0 && tid == 0 && loss_out) finish_loss<T>(loss_part, n_loss, loss_scale, loss_out);
        if (e == 0) wgrad_tile_ld<T, true>(nd, 0, F, params, (T*)nullptr, (const T*)nullptr, 0, (const int*)nullptr, d1, kP2H, B, scale, red, fragimg);
        else        wgrad_tile_ld<T, true>(nd, 1, (e - 1) * 16, params, (T*)nullptr, a1, kP2H, (const int*)nullptr, d2, kP2C, B, scale, red, fragimg);
        RCN_STAMP(0, 6);
        return;
    }

    const int f0 = (int)blockIdx.x * 16;
    const int nf = F - f0 < 16 ? F - f0 : 16;
    T* wsl = red + kDenseWaves * kMtp * kRedTile;                   // this slice of W_0: [feature 0..15][32]
    T* W0 = params + nd.w_off[0];
    const T* __restrict__ cp = Xp + (size_t)blockIdx.x * B * 16;
    const T* __restrict__ cn = Xn + (size_t)blockIdx.x * B * 16;
    const int ntile = B >> 4;                                        // B % 256 == 0: 2*k tiles per wave, no remainder

    // new batch first (needed last): lane (n, g4) <- features 4g4..4g4+3 of sample 16t+n
    vec4 xn[2];
    if (do_fwd) {
#pragma unroll
        for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (wave + 8 * u) + n) * 16 + 4 * g4);
    }

    if (do_update) {
        // ---- U: dW_0[:, slice] = sum_s delta_1[s] (x) x_s[slice]; W_0 <- W_0 - (eta/B) dW_0          rcn.rs:310, 214
        const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8;
        const int m = mt * 16 + ml;
        const bool wvalid = m < H && cl < nf;
        const T wold = W0[(size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0)];
        acc_t acc[kMtp];
#pragma unroll
        for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
        const int kw = B >> 3;                                       // samples per wave, a multiple of 32
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
            // every load of the chunk is in flight before the first MFMA waits on one: without this fence the scheduler starts
            // the MFMA chain after 16 of the 24 loads and issues the last 8 behind the first wait -- a second memory round trip
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(av[q][t], bv[q], acc[t]);
        }
        RCN_STAMP(0, 1);
        store_partials<T>(red, wave, lane, acc);
        __syncthreads();
        RCN_STAMP(0, 2);
        {
            const T w = wold - scale * sum_partials<T>(red, mt, cl, ml);
            if (wvalid) W0[(size_t)(f0 + cl) * H + m] = w;
            wsl[cl * kP2H + m] = wvalid ? w : (T)0;
        }
        __syncthreads();
    } else {
        for (int e = tid; e < 16 * kP2H; e += kDenseThreads) {
            const int cl = e >> 5, m = e & 31;
            const bool ok = m < H && cl < nf;
            const T w = W0[(size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0)];
            wsl[e] = ok ? w : (T)0;
        }
        __syncthreads();
    }
    RCN_STAMP(0, 3);
    if (!do_fwd) return;

    // ---- F: partial z_1 for all samples of the new batch from the slice that is still in LDS
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
    RCN_STAMP(0, 5);
}

// One workgroup per 8 samples, 8 waves.  All waves sum slabs (slices w, w+8, ...).  Waves 1-7 additionally fetch and
// mask the tail's operands into LDS as ready-made MFMA fragments (a few loads each), so that wave 0 -- the critical
// path, ~5.6 cycles per instruction when a wave runs alone on its SIMD -- only reads fragments, issues 16 MFMAs and stores.
constexpr int kP2BFrag = 28;                                      // fragment words per lane: wz 8, wd 8, b1 4, y 4, b0 4
inline size_t p2_b_lds_elems() { return (size_t)kP2BWaves * 64 * 4 + kP2H * kLd + kP2C * kLd + kP2BFrag * 64; }

template <typename T>
__device__ __forceinline__ void p2_b_body(
    const NetDesc& nd, const T* __restrict__ params, const T* __restrict__ slab, int G, const T* __restrict__ Ys, int B,
    T* __restrict__ a1g, T* __restrict__ d1g, T* __restrict__ d2g, T* __restrict__ loss_part, unsigned char* smem_raw,
    const T* __restrict__ fragimg = nullptr) {
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    constexpr int kFrag = kP2BFrag;
    constexpr int kPer = kP2MaxSlices / kP2BWaves;                // slab slices per wave (8)
    T* smem = reinterpret_cast<T*>(smem_raw);
    vec4* zred = reinterpret_cast<vec4*>(smem);                   // [8 waves][64 lanes]
    T* a1s = smem + kP2BWaves * 64 * 4;                           // a_1 tile  [hidden 32][kLd]  (MFMA B operand image)
    T* d2s = a1s + kP2H * kLd;                                    // delta_2   [class 16][kLd]
    T* frag = d2s + kP2C * kLd;                                   // [word][lane]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1], C = nd.dims[2];
    const int s0 = blockIdx.x * kP2Ts;
    RCN_STAMP(1, 0);

    // ---- slab loads: 8 slices per wave in flight (G <= 64); lane <- float4 `lane` of the [8][32] slice tile
    const vec4* sp = reinterpret_cast<const vec4*>(slab + (size_t)blockIdx.x * G * kP2Ts * kP2H) + lane;
    vec4 t[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int g = wave + kP2BWaves * q;
        t[q] = sp[(size_t)(g < G ? g : wave) * 64];
    }
    // With a fragment image (kept current by k_p2_a's tail tiles, dense.hpp: p2_frag_scatter) the finishing wave fetches its
    // 24 parameter words itself, 256 contiguous bytes per load; only the targets still go through LDS.  Without one, waves
    // 1-5 and 7 gather them from the parameter vector: ~24 loads of 64 scattered addresses per workgroup, which the CU's
    // address unit works through for ~0.35 us behind the slab stream -- measured, that is when those waves reach the barrier.
    T fr[kFrag];
    if (fragimg && wave == 0) {
#pragma unroll
        for (int q = 0; q < kFrag; ++q) fr[q] = (q >= 20 && q < 24) ? (T)0 : fragimg[q * 64 + lane];
    }
    const bool gather = fragimg == nullptr;
    const T* W1 = params + nd.w_off[1];                           // C x H column-major: (c, h) at h*C + c
    if (!gather && wave != 6) {
    } else if (wave == 1 || wave == 2) {                                 // z_2 = W_1 a_1:   A[m = c][k = h]; 4 k-steps each
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ks = 4 * (wave - 1) + q, h = 4 * ks + g4;
            const T w = W1[(size_t)(h < H ? h : 0) * C + (n < C ? n : 0)];
            frag[ks * 64 + lane] = (h < H && n < C) ? w : (T)0;
        }
    } else if (wave == 3 || wave == 4) {                          // W_1^T delta_2:  A[m = h][k = c]; one M-tile each
        const int mt = wave - 3;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int h = mt * 16 + n, c = 4 * ks + g4;
            const T w = W1[(size_t)(h < H ? h : 0) * C + (c < C ? c : 0)];
            frag[(8 + mt * 4 + ks) * 64 + lane] = (h < H && c < C) ? w : (T)0;
        }
    } else if (wave == 5) {                                       // b_1 per accumulator element
        const T* b1 = W1 + (size_t)C * H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = Mfma16<T>::row(lane, i);
            const T bb = b1[c < C ? c : 0];
            frag[(16 + i) * 64 + lane] = c < C ? bb : (T)0;
        }
    } else if (wave == 6) {                                       // targets per accumulator element
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = Mfma16<T>::row(lane, i);
            frag[(20 + i) * 64 + lane] = Ys[(size_t)(s0 + (n & 7)) * C + (c < C ? c : 0)];
        }
    } else if (wave == 7) {                                       // b_0 per slab float4 element
        const T* b0 = params + nd.w_off[0] + (size_t)H * F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = 4 * (lane & 7) + i;
            const T b = b0[h < H ? h : 0];
            frag[(24 + i) * 64 + lane] = h < H ? b : (T)0;
        }
    }
    vec4 z = vec4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < kPer; ++q)
        if (wave + kP2BWaves * q < G) z += t[q];                  // slice order within a wave ...
    zred[wave * 64 + lane] = z;
    RCN_STAMP(1, 1);
    __syncthreads();
    RCN_STAMP(1, 2);
    if (wave != 0) return;
    {   // ... and a fixed order across waves.  Written on the two halves of each 16-byte word so that it compiles to packed adds on the
        // register pairs the reads delivered (left to itself the compiler packs ACROSS the eight operands and spends ~40 moves on it)
        typedef T h2 __attribute__((ext_vector_type(2)));
        vec4 r[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) r[w] = zred[w * 64 + lane];
        h2 lo[8], hi[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) { lo[w] = h2{r[w][0], r[w][1]}; hi[w] = h2{r[w][2], r[w][3]}; }
        const h2 zl = ((lo[0] + lo[1]) + (lo[2] + lo[3])) + ((lo[4] + lo[5]) + (lo[6] + lo[7]));
        const h2 zh = ((hi[0] + hi[1]) + (hi[2] + hi[3])) + ((hi[4] + hi[5]) + (hi[6] + hi[7]));
        z = vec4{zl[0], zl[1], zh[0], zh[1]};
    }

    // every fragment word this wave will use, in ONE round of LDS reads right behind the barrier (read where they are used,
    // each group costs the chain another LDS round trip: the compiler may not hoist them over the a_1 / delta_2 tile writes)
    if (gather) {
#pragma unroll
        for (int q = 0; q < kFrag; ++q) fr[q] = frag[q * 64 + lane];
    } else {
#pragma unroll
        for (int q = 20; q < 24; ++q) fr[q] = frag[q * 64 + lane];
    }

    // ---- a_1 = sigmoid(z_1 + b_0); lane <- sample lane>>3, hidden 4*(lane&7)+i                       rcn.rs:287-289
    {
        const int s = lane >> 3, h0 = 4 * (lane & 7);
        vec4 a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto sg = sigmoid_fast(z[i] + fr[24 + i]);   // unconditionally: as `c ? f(x) : 0` this is four branches
            a[i] = (h0 + i < H) ? sg : (T)0;
            a1s[(h0 + i) * kLd + s] = a[i];
        }
        *reinterpret_cast<vec4*>(a1g + (size_t)(s0 + s) * kP2H + h0) = a;
    }
    RCN_STAMP(1, 3);
    // ---- z_2 = W_1 a_1 + b_1, a_2 = sigmoid, delta_2 = (a_2 - y)(*)a_2(1-a_2)                    rcn.rs:287-289, 299
    acc_t acc = acc_t{0, 0, 0, 0};
    {
        T bv[8];                                                  // all eight operands in one round of LDS reads, then the MFMA chain
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bv[ks] = a1s[(4 * ks + g4) * kLd + (n & 7)];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = Mfma16<T>::mfma(fr[ks], n < kP2Ts ? bv[ks] : (T)0, acc);
    }
    T lsum = 0;
    acc_t dv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = Mfma16<T>::row(lane, i);
        const T a2 = sigmoid_fast(acc[i] + fr[16 + i]);
        const T diff = a2 - fr[20 + i];
        const bool ok = c < C && n < kP2Ts;
        dv[i] = ok ? diff * (a2 * ((T)1 - a2)) : (T)0;
        lsum += ok ? diff * diff : (T)0;
        d2s[c * kLd + n] = dv[i];
    }
    if (n < kP2Ts) store4<T>(d2g + (size_t)(s0 + n) * kP2C, lane, dv);
    RCN_STAMP(1, 4);
    // ---- delta_1 = (W_1^T delta_2) (*) a_1 (1 - a_1)                                                rcn.rs:305-309
    {
        T dvv[4], av[kMtp][4];                                    // again one round of LDS reads for both M-tiles
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dvv[ks] = d2s[(4 * ks + g4) * kLd + n];
#pragma unroll
        for (int mt = 0; mt < kMtp; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) av[mt][i] = a1s[(mt * 16 + Mfma16<T>::row(lane, i)) * kLd + (n & 7)];
#pragma unroll
        for (int mt = 0; mt < kMtp; ++mt) {
            acc_t ad = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ad = Mfma16<T>::mfma(fr[8 + mt * 4 + ks], dvv[ks], ad);
            acc_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ad[i] * (av[mt][i] * ((T)1 - av[mt][i]));      // padded hidden rows hold a = 0 -> delta 0
            if (n < kP2Ts) store4<T>(d1g + (size_t)(s0 + n) * kP2H + mt * 16, lane, o);
        }
    }
    lsum = wave_sum_lane0(lsum);
    if (lane == 0 && loss_part) loss_part[blockIdx.x] = lsum;
    RCN_STAMP(1, 6);
}

// the whole fragment image from the parameter vector (once per train_epoch / train_batch call; afterwards k_p2_a keeps it current)
__global__ __launch_bounds__(512) void k_p2_fragimg(NetDesc nd, const float* __restrict__ params, float* __restrict__ img) {
    const int F = nd.dims[0], H = nd.dims[1], C = nd.dims[2];
    for (int e = threadIdx.x; e < kP2BFrag * 64; e += 512) img[e] = 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < (H + 1) * C; e += 512) p2_frag_scatter(1, e / C, e % C, H, params[nd.w_off[1] + e], img);
    for (int e = threadIdx.x; e < H; e += 512) p2_frag_scatter(0, F, e, H, params[nd.w_off[0] + H * F + e], img);
}

// ---- the two kernels of a step, and both in ONE kernel object -------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_p2_a(
    NetDesc nd, T* __restrict__ params, const T* __restrict__ Xp, const T* __restrict__ Xn, int B,
    const T* __restrict__ a1, const T* __restrict__ d1, const T* __restrict__ d2, T scale, T* __restrict__ slab, int G,
    const T* __restrict__ loss_part, int n_loss, T loss_scale, T* __restrict__ loss_out, int do_update, int do_fwd, T* __restrict__ fragimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    p2_a_body<T>(nd, params, Xp, Xn, B, a1, d1, d2, scale, slab, G, loss_part, n_loss, loss_scale, loss_out, do_update, do_fwd, smem_dyn, fragimg);
}

template <typename T>
__global__ __launch_bounds__(kP2BThreads) void k_p2_b(
    NetDesc nd, const T* __restrict__ params, const T* __restrict__ slab, int G, const T* __restrict__ Ys, int B,
    T* __restrict__ a1g, T* __restrict__ d1g, T* __restrict__ d2g, T* __restrict__ loss_part, const T* __restrict__ fragimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    p2_b_body<T>(nd, params, slab, G, Ys, B, a1g, d1g, d2g, loss_part, smem_dyn, fragimg);
}

#ifdef RCN_HIP_EXPERIMENTS
// The epoch loop alternates the two strictly, and the pair costs ~0.9 us more than the two back to back with themselves
// (9.2 us vs 3.85 + 4.45).  Hypothesis tested here: the switch of kernel object (code, descriptor, LDS / register allocation)
// between launches.  As two ROLES of one kernel object the step is 2 % SLOWER (9.75 vs 9.55 us), so that is not it; kept
// behind RCN_HIP_P2_ONE_OBJECT=1 as the record of the experiment.
static_assert(kDenseThreads == kP2BThreads, "one launch shape for both roles");
inline size_t p2_ab_lds_elems() { return p2_a_lds_elems() > p2_b_lds_elems() ? p2_a_lds_elems() : p2_b_lds_elems(); }

template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_p2_ab(
    int role, NetDesc nd, T* __restrict__ params, const T* __restrict__ Xp, const T* __restrict__ Xn, const T* __restrict__ Ys, int B,
    T* __restrict__ a1, T* __restrict__ d1, T* __restrict__ d2, T scale, T* __restrict__ slab, int G, T* __restrict__ loss_part, int n_loss,
    T loss_scale, T* __restrict__ loss_out, int do_update, int do_fwd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    if (role == 0) p2_b_body<T>(nd, params, slab, G, Ys, B, a1, d1, d2, loss_part, smem_dyn);
    else p2_a_body<T>(nd, params, Xp, Xn, B, a1, d1, d2, scale, slab, G, loss_part, n_loss, loss_scale, loss_out, do_update, do_fwd, smem_dyn);
}

#endif  // RCN_HIP_EXPERIMENTS

}  // namespace rcn
