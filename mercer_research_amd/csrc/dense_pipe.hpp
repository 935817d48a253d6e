// dense_pipe.hpp -- feature-sliced ("split-K") pipelined train_batch for small batches (rcn.rs:176-314).
//
// A train_batch at B = 256 moves ~1 MB and ~25 MFLOP: it is bound by how many bytes each CU has to pull
// through its own memory pipe and by the number of dependent phases, not by HBM or MFMA peak.  The sample-tile
// kernels of dense.hpp put the whole first layer of a 16-sample tile on one CU (16 workgroups x ~147 KB at
// B = 256).  This path instead cuts the first layer along its INPUT features:
//
//   k_pipe_a  one workgroup per 16-feature slice of W_0 (49 for F = 784), plus a few workgroups for the bias
//             column and the small tail layers.  In one launch a slice workgroup
//               U) finishes step i-1:  dW_0[:, slice] = Delta_1^T . X_{i-1}[:, slice] over the whole batch
//                  (MFMA, contraction = sample) and applies W_0 <- W_0 - (eta/B) dW_0   (rcn.rs:210-222),
//               F) starts step i:      partial Z_1 = W_0[:, slice] . X_i[:, slice]^T for all samples with the
//                  slice it has just updated still in LDS, written as one slab per slice.
//             Both contractions of the big layer stay local to the slice; nothing else of W_0 is touched.
//             (Xp / idx_prev address the batch of the step being finished, X / idx_new the batch being started.)
//   k_pipe_b  one workgroup per 8 samples: sums the slabs in slice order (fixed order => bit-reproducible),
//             adds b_0, sigmoid, runs the small tail layers forward, the output delta (rcn.rs:299) and the
//             back-propagated deltas (rcn.rs:305-309) in plain FMAs in the reference's own summation order,
//             and stores activations / deltas for the next k_pipe_a.
//
//   k_pack_epoch  once per epoch (or per single step): gathers the shuffled batches (rcn.rs:146-147) into a slice-major
//             image Xs[batch][slice][sample][16 features] (+ targets Ys[batch][sample][classes]), so that every read
//             in the per-step kernels is one contiguous region per workgroup, 16 bytes per lane, with no index
//             indirection in front of it.  (Measured on MI355X: a 47 KB hand-off read costs 0.8 us contiguous vs 2.3 us
//             as 4-byte loads at a 32 KB stride; every dependent global round trip costs >= 1.2 us.)
//
// Per step: one k_pipe_b and one k_pipe_a; the only cross-CU exchange is the slab sum.  The data-parallel
// (gradient-out) form and layer stacks whose tail does not fit LDS use the kernels of dense.hpp.
#pragma once

#include "common.hpp"
#include "dense.hpp"

namespace rcn {

constexpr int kPipeTs = 8;            // samples per k_pipe_b workgroup
constexpr int kPipeBThreads = 256;
constexpr int kPipeMaxDim = 128;      // widest tail layer handled by k_pipe_b

inline int pipe_slices(const NetDesc& nd) { return (nd.dims[0] + 15) / 16; }
inline int pipe_mp(const NetDesc& nd) { return (nd.dims[1] + 3) & ~3; }                 // padded slab row
inline int pipe_extra_wgs(const NetDesc& nd) { return 1 + nd.tile_start[nd.L] - nd.tile_start[1]; }
inline bool pipe_supported(const NetDesc& nd) {
    if (nd.L < 2 || !dense_tail_staged(nd)) return false;
    for (int j = 1; j <= nd.L; ++j)
        if (nd.dims[j] > kPipeMaxDim) return false;
    return true;
}
inline size_t pipe_a_lds_elems(const NetDesc& nd) { return (size_t)kDenseWaves * kMtp * kRedTile + 16 * (size_t)(((nd.dims[1] + 31) / 32) * 32) + 64; }
inline size_t pipe_b_lds_elems(const NetDesc& nd) {
    int sumd = 0, maxd = 0;
    for (int j = 1; j <= nd.L; ++j) { sumd += nd.dims[j]; if (nd.dims[j] > maxd) maxd = nd.dims[j]; }
    return (size_t)((dense_tail_params(nd) + nd.dims[1] + 3) & ~3) + (size_t)kPipeTs * (sumd + 2 * maxd) + 64 + (size_t)kPipeBThreads * 4;
}

// Xp / X: packed slice-major images of the batch being finished / started: chunk(g) = base + g * B * 16 elements, element
// (sample s, feature 16 g + f) at chunk[s * 16 + f]; features past F are stored as zeros by k_pack_epoch.
template <typename T>
__global__ __launch_bounds__(kDenseThreads) void k_pipe_a(
    NetDesc nd, T* __restrict__ params, const T* __restrict__ Xp, const T* __restrict__ X, int B,
    const T* __restrict__ acts, const T* __restrict__ deltas, T scale,
    T* __restrict__ slab, int G, const T* __restrict__ loss_part, int n_loss, T loss_scale,
    T* __restrict__ loss_out, int do_update, int do_fwd) {
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    RCN_STAMP(0, 0);

    if ((int)blockIdx.x >= G) {
        // bias column of layer 0 and every tile of the tail layers: the generic whole-batch wgrad tile
        if (!do_update) return;
        const int e = (int)blockIdx.x - G;
        if (e == 0 && tid == 0 && loss_out) finish_loss<T>(loss_part, n_loss, loss_scale, loss_out);
        int j = 0, n0 = nd.dims[0];                      // e == 0: the tile that starts at the bias column of W_0
        if (e > 0) {
            const int t = nd.tile_start[1] + e - 1;
            j = 1;
            while (j + 1 < nd.L && t >= nd.tile_start[j + 1]) ++j;
            n0 = (t - nd.tile_start[j]) * 16;
        }
        wgrad_tile<T, true>(nd, j, n0, params, (T*)nullptr, (const T*)nullptr, (const int*)nullptr, acts, deltas, B, scale, red);
        return;
    }

    const int F = nd.dims[0], M = nd.dims[1];
    const int Mp = (M + 3) & ~3, Mw = ((M + 31) / 32) * 32;
    const int f0 = (int)blockIdx.x * 16;
    const int nf = F - f0 < 16 ? F - f0 : 16;
    T* wsl = red + kDenseWaves * kMtp * kRedTile;        // this slice of W_0, [feature 0..15][Mw]
    T* W0 = params + nd.w_off[0];
    const T* __restrict__ cp = Xp + (size_t)blockIdx.x * B * 16;      // this slice's chunk of the finished batch
    const T* __restrict__ cn = X + (size_t)blockIdx.x * B * 16;       // ... of the batch being started
    const int ntile = (B + 15) / 16;

    // Issue the new batch's loads first: nothing below depends on them until part F, so their latency hides behind U.
    // lane (n, g4) holds features 4g4 .. 4g4+3 of sample 16t+n; MFMA i contracts feature 4g4+i.
    constexpr int kFt = 4;                                // sample tiles per wave held in registers (B <= 512)
    vec4 xn[kFt];
    if (do_fwd) {
#pragma unroll
        for (int u = 0; u < kFt; ++u) {
            const int s = 16 * (wave + u * kDenseWaves) + n;
            xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(s < B ? s : B - 1) * 16 + 4 * g4);
        }
    }

    if (do_update) {
        // ---- U: dW_0[:, f0..f0+nf) = sum_s delta_1[s] (x) x_s[f0..]   (rcn.rs:310 summed as in :190-205) + SGD
        const T* __restrict__ D = deltas + (size_t)B * nd.act_off[1];
        int kb, ke;
        wave_k_range(B, wave, kb, ke);
        for (int mbase = 0; mbase < M; mbase += 16 * kMtp) {
            acc_t acc[kMtp];
#pragma unroll
            for (int mt = 0; mt < kMtp; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
            // the old parameters of this tile: issued with the operand loads, consumed in the epilogue
            T wold = 0;
            {
                const int mt = tid >> 8, o = tid & 255, cl = o >> 4, ml = o & 15;
                const int m = mbase + mt * 16 + ml;
                wold = W0[(size_t)(f0 + (cl < nf ? cl : 0)) * M + (m < M ? m : M - 1)];
            }
            for (int kc = kb; kc < ke; kc += 32) {
                T bv[8], av[8][kMtp];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int s = kc + 4 * q + g4;
                    const int sc = s < ke ? s : kb;            // clamp: unconditional loads, masked by value below
                    bv[q] = cp[(size_t)sc * 16 + n];
#pragma unroll
                    for (int mt = 0; mt < kMtp; ++mt) {
                        const int row = mbase + mt * 16 + n;
                        av[q][mt] = D[(size_t)sc * M + (row < M ? row : M - 1)];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool sv = kc + 4 * q + g4 < ke;
#pragma unroll
                    for (int mt = 0; mt < kMtp; ++mt)
                        acc[mt] = Mfma16<T>::mfma((sv && mbase + mt * 16 + n < M) ? av[q][mt] : (T)0, sv ? bv[q] : (T)0, acc[mt]);
                }
            }
            RCN_STAMP(0, 1);
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            RCN_STAMP(0, 2);
            {
                const int mt = tid >> 8, o = tid & 255, cl = o >> 4, ml = o & 15;
                const int m = mbase + mt * 16 + ml;
                if (m < M && cl < nf) {
                    const T w = wold - scale * sum_partials<T>(red, mt, cl, ml);      // rcn.rs:214
                    W0[(size_t)(f0 + cl) * M + m] = w;
                    wsl[cl * Mw + m] = w;
                }
            }
            __syncthreads();
        }
    } else {
        for (int e = tid; e < nf * M; e += kDenseThreads) wsl[(e / M) * Mw + (e % M)] = W0[(size_t)f0 * M + e];
        __syncthreads();
    }
    RCN_STAMP(0, 3);
    if (!do_fwd) return;

    // ---- F: partial Z_1[m][s] = sum_{k in slice} W_0[m][k] x_s[k] for every sample of the new batch, one slab per
    // slice laid out [8-sample tile][slice][sample][Mp] so that each k_pipe_b workgroup reads ONE contiguous region.
    for (int tb = 0; tb < ntile; tb += kFt * kDenseWaves) {
        if (tb > 0) {
#pragma unroll
            for (int u = 0; u < kFt; ++u) {
                const int s = 16 * (tb + wave + u * kDenseWaves) + n;
                xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(s < B ? s : B - 1) * 16 + 4 * g4);
            }
        }
        RCN_STAMP(0, 4);
#pragma unroll
        for (int u = 0; u < kFt; ++u) {
            const int t = tb + wave + u * kDenseWaves;
            if (t >= ntile) break;
            const int s = 16 * t + n;
            for (int mbase = 0; mbase < M; mbase += 16 * kMtp) {
                acc_t acc[kMtp];
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < kMtp; ++mt) {
                        const int row = mbase + mt * 16 + n;
                        const T wv = wsl[(4 * g4 + i) * Mw + (row < M ? row : M - 1)];
                        acc[mt] = Mfma16<T>::mfma((row < M && 4 * g4 + i < nf) ? wv : (T)0, s < B ? xn[u][i] : (T)0, acc[mt]);
                    }
                T* dst = slab + (((size_t)(s >> 3) * G + blockIdx.x) * kPipeTs + (s & 7)) * Mp;
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = mbase + mt * 16 + Mfma16<T>::row(lane, i);
                        if (row < M) dst[row] = acc[mt][i];
                    }
            }
        }
    }
    RCN_STAMP(0, 5);
}

// training_set.shuffle (rcn.rs:146) on the device: `passes` independent pseudo-random permutations of 0..n-1, pass p at
// perm[p*n ..].  Each is a keyed 4-round balanced Feistel network over the next even-bit power of two >= n (a bijection),
// cycle-walked into [0, n); the key is a hash of (seed, pass).  One thread per element, no sort, no atomics.
__device__ inline unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void k_shuffle_indices(int* __restrict__ perm, unsigned n, unsigned passes, unsigned long long seed, int half_bits) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (unsigned long long)n * passes) return;
    const unsigned p = (unsigned)(gid / n), i = (unsigned)(gid - (unsigned long long)p * n);
    const unsigned mask = (1u << half_bits) - 1u;
    unsigned key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) key[r] = mix32((unsigned)(seed >> (16 * (r & 1))) ^ mix32(p * 0x9e3779b9U + r * 0x85ebca6bU + (unsigned)(seed >> 32)));
    unsigned x = i;
    do {
        unsigned L = x >> half_bits, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned t = L ^ (mix32(R ^ key[r]) & mask);
            L = R; R = t;
        }
        x = (L << half_bits) | R;
    } while (x >= n);
    perm[gid] = (int)x;
}

// Gathers `nb` shuffled batches into the slice-major image described above.  Grid: ((G + 1) / 2 + 1, nb): blockIdx.x < (G + 1) / 2
// packs TWO neighbouring 16-feature slices of one batch -- 128 contiguous bytes of every (f32) sample row, a whole cache line
// per gathered row instead of half of one (as one slice per workgroup the other half was fetched again by the neighbouring
// slice's workgroup, on another XCD) -- the last blockIdx.x packs that batch's targets.  perm == NULL: identity
// (chunks_exact over the set as stored).
inline int pack_grid_x(int G) { return (G + 1) / 2 + 1; }

template <typename T, bool VECX>
__global__ __launch_bounds__(256) void k_pack_epoch(const T* __restrict__ X, const T* __restrict__ Y, const int* __restrict__ perm,
                                                    int B, int F, int C, int G, T* __restrict__ Xs, T* __restrict__ Ys) {
    using vec4 = typename Vec4<T>::type;
    const int j = blockIdx.y, gp = blockIdx.x;
    const int* pj = perm ? perm + (size_t)j * B : nullptr;
    if (gp == (G + 1) / 2) {
        T* dst = Ys + (size_t)j * B * C;
        for (int e = threadIdx.x; e < B * C; e += blockDim.x) {
            const int s = e / C, m = e - s * C;
            const long long r = pj ? (long long)pj[s] : (long long)j * B + s;
            dst[e] = Y[r * (long long)C + m];
        }
        return;
    }
    T* dst = Xs + ((size_t)j * G + 2 * gp) * B * 16;                   // chunk of slice 2 gp; slice 2 gp + 1 follows B * 16 elements later
    const int f0 = 32 * gp;
    const bool two = 2 * gp + 1 < G;
    for (int e = threadIdx.x; e < B * 8; e += blockDim.x) {            // one 4-feature group per thread-iteration
        const int s = e >> 3, q = e & 7, f = f0 + 4 * q;
        if (q >= 4 && !two) continue;
        const long long r = pj ? (long long)pj[s] : (long long)j * B + s;
        vec4 v = vec4{0, 0, 0, 0};
        if (VECX) {
            if (f < F) v = *reinterpret_cast<const vec4*>(X + r * (long long)F + f);      // F % 4 == 0: whole groups only
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (f + i < F) ? X[r * (long long)F + f + i] : (T)0;
        }
        *reinterpret_cast<vec4*>(dst + (size_t)(q >> 2) * B * 16 + (size_t)s * 16 + 4 * (q & 3)) = v;
    }
}

// One workgroup per 8 samples.  Ys: this batch's targets, packed [sample][classes].
// (f32: sigmoid_fast -- v_exp_f32 / v_rcp_f32, ~1e-7 relative, inside the 2e-6 activation tolerance and what k_p2_b uses; the
// libm-grade expf and the IEEE division of sigmoid_ref cost ~60 instructions per value, and the waves that evaluate it run alone on
// their SIMD at ~5.6 cycles per instruction: by the stamps 0.6 us of this kernel's 1.0 us a_1 phase.  f64: unchanged.)
template <typename T>
__global__ __launch_bounds__(kPipeBThreads) void k_pipe_b(
    NetDesc nd, const T* __restrict__ params, const T* __restrict__ slab, int G, const T* __restrict__ Ys, int B,
    T* __restrict__ acts, T* __restrict__ deltas, T* __restrict__ loss_part) {
    using vec4 = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, L = nd.L;
    const int F = nd.dims[0], M = nd.dims[1], Mp = (M + 3) & ~3, C = nd.dims[L];
    const int s0 = blockIdx.x * kPipeTs;
    int sumd = 0, maxd = 0;
    for (int j = 1; j <= L; ++j) { sumd += nd.dims[j]; maxd = nd.dims[j] > maxd ? nd.dims[j] : maxd; }
    const int tail0 = nd.w_off[1], ntail = nd.P - tail0;

    T* wsm = smem;                                        // [tail parameters | b_0]
    T* b0 = wsm + ntail;
    T* act = smem + ((ntail + M + 3) & ~3);               // a_j[s][d_j] at kPipeTs * act_off[j]
    T* dA = act + kPipeTs * sumd;                         // delta ping  [s][maxd]
    T* dB = dA + kPipeTs * maxd;                          // delta pong
    T* lred = dB + kPipeTs * maxd;                        // 4 wave partials of the loss
    vec4* zpart = reinterpret_cast<vec4*>(smem + (((ntail + M + 3) & ~3) + kPipeTs * (sumd + 2 * maxd) + 64));   // [slice group][float4 group]
    RCN_STAMP(1, 0);

    // ---- every global read of the kernel is issued here, before the first wait: the slab (one contiguous region of
    // G * 8 * Mp elements, 16 bytes per lane), the targets, and the small parameters.
    // The tile holds nq = 8 * Mp / 4 float4 groups per slice -- 24 for the reference's ten-unit test net -- so handing one group to a
    // thread would leave 232 of the 256 threads without a load and the 24 others with all G of them, four dependent round trips
    // deep (measured: 8.8 us for this kernel against 3.4 us for its specialised sibling).  Instead the slices are dealt over
    // NG = 256 / nq thread groups: group r sums slices r, r + NG, r + 2 NG, ... in that order (one round of loads in flight), and the
    // groups' partials meet in LDS and are added in group order -- a fixed order, so the result is bit-reproducible.
    // the small parameters first (into registers: up to four values per thread, the usual case), so that their round trip runs
    // under the slab's instead of behind it
    constexpr int kPre = 4;
    T pre[kPre], preb = 0;
    const bool pre_ok = ntail <= kPre * kPipeBThreads && M <= kPipeBThreads;
    if (pre_ok) {
#pragma unroll
        for (int q = 0; q < kPre; ++q) { const int e = tid + q * kPipeBThreads; pre[q] = params[tail0 + (e < ntail ? e : 0)]; }
        preb = params[nd.w_off[0] + (size_t)M * F + (tid < M ? tid : 0)];
    }
    const int nq = kPipeTs * Mp / 4;                      // float4 groups per slice in this tile
    const int NG = kPipeBThreads / nq > 0 ? kPipeBThreads / nq : 1;
    {
        const int grp = tid / nq, col = tid - grp * nq;
        vec4 zp = vec4{0, 0, 0, 0};
        if (grp < NG) {
            const vec4* p = reinterpret_cast<const vec4*>(slab + (size_t)blockIdx.x * G * kPipeTs * Mp) + col;
            for (int g0 = grp; g0 < G; g0 += 8 * NG) {
                vec4 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = p[(size_t)(g0 + q * NG < G ? g0 + q * NG : grp) * nq];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (g0 + q * NG < G) zp += t[q];
            }
            zpart[grp * nq + col] = zp;
        }
    }
    T yv[(kPipeTs * kPipeMaxDim + kPipeBThreads - 1) / kPipeBThreads];
    {
        int u = 0;
        for (int e = tid; e < kPipeTs * C; e += kPipeBThreads, ++u) {
            const int s = e / C;
            yv[u] = Ys[(size_t)(s0 + s < B ? s0 + s : B - 1) * C + (e - s * C)];
        }
    }
    if (pre_ok) {
#pragma unroll
        for (int q = 0; q < kPre; ++q) { const int e = tid + q * kPipeBThreads; if (e < ntail) wsm[e] = pre[q]; }
        if (tid < M) b0[tid] = preb;
    } else {
        for (int e = tid; e < ntail; e += kPipeBThreads) wsm[e] = params[tail0 + e];
        for (int e = tid; e < M; e += kPipeBThreads) b0[e] = params[nd.w_off[0] + (size_t)M * F + e];
    }
    RCN_STAMP(1, 1);
    __syncthreads();                                      // staged parameters visible
    RCN_STAMP(1, 2);

    // ---- a_1 = sigmoid(z_1 + b_0)                                                                 rcn.rs:287-289
    if (tid < nq) {
        vec4 z = zpart[tid];
        for (int r0 = 1; r0 < NG; r0 += 8) {             // group order, eight partials per round of LDS reads
            vec4 t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = zpart[(r0 + q < NG ? r0 + q : 0) * nq + tid];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (r0 + q < NG) z += t[q];
        }
        const int s = tid / (Mp / 4), m4 = (tid - s * (Mp / 4)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m4 + i;
            if (m < M) {
                const T a = sigmoid_fast(z[i] + b0[m]);
                act[s * M + m] = a;                       // act_off[1] == 0
                if (s0 + s < B) acts[(size_t)B * nd.act_off[1] + (size_t)(s0 + s) * M + m] = a;
            }
        }
    }
    __syncthreads();
    RCN_STAMP(1, 3);

    // ---- tail layers forward: z = W a + b accumulated column by column like nalgebra's gemv           rcn.rs:287
    // (LDS operands are fetched 8 k-steps at a time so that the LDS latency is paid per group, not per term)
    for (int j = 1; j < L; ++j) {
        const int dk = nd.dims[j], dn = nd.dims[j + 1];
        const T* W = wsm + (nd.w_off[j] - tail0);
        const T* bj = W + (size_t)dn * dk;
        const T* aj = act + kPipeTs * nd.act_off[j];
        T* an = act + kPipeTs * nd.act_off[j + 1];
        for (int e = tid; e < kPipeTs * dn; e += kPipeBThreads) {
            const int s = e / dn, m = e - s * dn;
            T zz = 0;
            for (int k0 = 0; k0 < dk; k0 += 8) {
                T w[8], a[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = k0 + q < dk ? k0 + q : dk - 1;
                    w[q] = W[(size_t)k * dn + m];
                    a[q] = aj[s * dk + k];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < dk) zz = (k0 + q == 0) ? w[q] * a[q] : w[q] * a[q] + zz;
            }
            const T a = sigmoid_fast(zz + bj[m]);
            an[s * dn + m] = a;
            if (j + 1 < L && s0 + s < B) acts[(size_t)B * nd.act_off[j + 1] + (size_t)(s0 + s) * dn + m] = a;
        }
        __syncthreads();
    }
    RCN_STAMP(1, 4);

    // ---- output delta (a_L - y) (*) s'(z_L)                                                      rcn.rs:299
    T lsum = 0;
    {
        const T* aL = act + kPipeTs * nd.act_off[L];
        int u = 0;
        for (int e = tid; e < kPipeTs * C; e += kPipeBThreads, ++u) {
            const int s = e / C, m = e - s * C;
            T d = 0;
            if (s0 + s < B) {
                const T a = aL[s * C + m];
                const T diff = a - yv[u];
                d = diff * (a * ((T)1 - a));
                deltas[(size_t)B * nd.act_off[L] + (size_t)(s0 + s) * C + m] = d;
                lsum += diff * diff;
            }
            dA[s * C + m] = d;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if ((tid & 63) == 0) lred[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0 && loss_part) loss_part[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
    RCN_STAMP(1, 5);

    // ---- hidden deltas: (W_j^T delta_{j+1}) (*) s'(z_j), accumulated row by row like gemv on W^T    rcn.rs:305-309
    T* dCur = dA;
    T* dNxt = dB;
    for (int j = L - 1; j >= 1; --j) {
        const int dm = nd.dims[j], dk = nd.dims[j + 1];
        const T* W = wsm + (nd.w_off[j] - tail0);          // dk x dm column-major: (k, m) at m*dk + k
        const T* aj = act + kPipeTs * nd.act_off[j];
        for (int e = tid; e < kPipeTs * dm; e += kPipeBThreads) {
            const int s = e / dm, m = e - s * dm;
            T v = 0;
            for (int k0 = 0; k0 < dk; k0 += 8) {
                T w[8], dd[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = k0 + q < dk ? k0 + q : dk - 1;
                    w[q] = W[(size_t)m * dk + k];
                    dd[q] = dCur[s * dk + k];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < dk) v = (k0 + q == 0) ? w[q] * dd[q] : w[q] * dd[q] + v;
            }
            const T a = aj[s * dm + m];
            const T d = (s0 + s < B) ? v * (a * ((T)1 - a)) : (T)0;
            if (s0 + s < B) deltas[(size_t)B * nd.act_off[j] + (size_t)(s0 + s) * dm + m] = d;
            dNxt[s * dm + m] = d;
        }
        __syncthreads();
        T* t = dCur; dCur = dNxt; dNxt = t;
    }
    RCN_STAMP(1, 6);
}

}  // namespace rcn
