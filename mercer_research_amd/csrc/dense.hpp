// dense.hpp -- gfx950 kernels for rcn's dense sigmoid/MSE network (rcn.rs:105-116, 176-314).
//
// The reference walks the batch sample by sample (gemv + outer product, rcn.rs:260-314) and sums the
// per-sample gradients under a mutex (rcn.rs:190-205).  Here the batch is the N dimension of
// 16x16x4 MFMA tiles (f32 or f64 inputs, exact fused-multiply-add chains in that type):
//
//   k_dense_fwd   one workgroup per 16-sample tile: streams the tile's feature rows through LDS,
//                 Z = W.A + b and sigmoid for every layer, then (TRAIN) the output delta
//                 (a_L - y) * s'(z_L) (rcn.rs:299) and the back-propagated deltas (rcn.rs:305-309).
//                 K of every small GEMM is split over the 8 waves and combined through LDS in a
//                 fixed order, so results are bit-reproducible run to run.
//   k_dense_wgrad one workgroup per 16-column tile of [W_l | b_l]: dW = Delta . A^T with K = the
//                 whole batch (the sum over samples of rcn.rs:190-205 is the MFMA contraction), the
//                 bias gradient is the extra column whose A-value is the constant 1, and the SGD
//                 update W <- W - (eta/B) dW (rcn.rs:210-222) is the epilogue.  With APPLY=false the
//                 summed gradient is written out instead (data-parallel all-reduce path).
#pragma once

#include "common.hpp"

namespace rcn {

constexpr int kDenseThreads = 512;
constexpr int kDenseWaves = kDenseThreads / kWave;     // 8: K-split factor
constexpr int kMtp = 2;                                // M-tiles (16 rows each) accumulated per pass
constexpr int kRedTile = kTileS * kLd;                 // 272: one padded 16x16 partial tile
constexpr int kChunkK = 32;                            // features staged per wave per chunk

constexpr int kStageCap = 12288;                       // tail-layer parameters staged in LDS when they fit (elements)

// parameters of layers >= 1 (everything after [W_0|b_0]) -- staged into LDS by k_dense_fwd when small enough
inline int dense_tail_params(const NetDesc& nd) { return nd.L > 1 ? nd.P - nd.w_off[1] : 0; }
inline bool dense_tail_staged(const NetDesc& nd) { return dense_tail_params(nd) <= kStageCap; }
// feature rows can be loaded as 16-byte vectors straight into MFMA fragments
inline bool dense_vec_rows(const NetDesc& nd, size_t esz) { return nd.dims[0] % 4 == 0 && (nd.dims[0] * esz) % 16 == 0; }

// LDS elements (of T) k_dense_fwd needs
inline size_t dense_fwd_lds_elems(const NetDesc& nd) {
    int sumd = 0, maxd = 0;
    for (int j = 1; j <= nd.L; ++j) { sumd += nd.dims[j]; if (nd.dims[j] > maxd) maxd = nd.dims[j]; }
    const size_t stage = dense_tail_staged(nd) ? (size_t)((dense_tail_params(nd) + 3) & ~3) : 0;
    return (size_t)kLd * (size_t)(sumd + 2 * maxd + kDenseWaves * kChunkK) + (size_t)kDenseWaves * kMtp * kRedTile + 64 + stage;
}
inline size_t dense_wgrad_lds_elems() { return (size_t)kDenseWaves * kMtp * kRedTile + 64; }

// wave w's slice [kb, ke) of a K-long contraction, 4-aligned so MFMA k-steps never straddle slices
__device__ inline void wave_k_range(int K, int wave, int& kb, int& ke) {
    const int kw = (((K + kDenseWaves - 1) / kDenseWaves) + 3) & ~3;
    kb = wave * kw;
    ke = kb + kw < K ? kb + kw : K;
    if (kb > K) kb = K;
}

// Partial accumulators -> LDS as [col][row] padded tiles (col = MFMA N index = lane&15).
template <typename T>
__device__ inline void store_partials(T* red, int wave, int lane, const typename Mfma16<T>::acc_t (&acc)[kMtp]) {
#pragma unroll
    for (int mt = 0; mt < kMtp; ++mt) {
        T* t = red + (wave * kMtp + mt) * kRedTile + (lane & 15) * kLd;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[Mfma16<T>::row(lane, i)] = acc[mt][i];
    }
}

// Sum of the 8 waves' partials for element (col c, row r) of M-tile mt, fixed order w = 0..7.
template <typename T>
__device__ inline T sum_partials(const T* red, int mt, int c, int r) {
    T v = red[(0 * kMtp + mt) * kRedTile + c * kLd + r];
#pragma unroll
    for (int w = 1; w < kDenseWaves; ++w) v += red[(w * kMtp + mt) * kRedTile + c * kLd + r];
    return v;
}

template <typename T> struct Vec4;
template <> struct Vec4<float>  { using type = __attribute__((ext_vector_type(4))) float; };
template <> struct Vec4<double> { using type = __attribute__((ext_vector_type(4))) double; };

template <typename T> struct VecGroups { static constexpr int value = sizeof(T) == 8 ? 4 : 8; };   // 16-feature groups in flight (VGPR budget)

// VECX:   feature rows are 16-byte aligned -> layer 0 loads them as vectors directly into MFMA B fragments (no LDS
//         staging, every load of a wave's K-slice in flight at once); otherwise rows are staged through LDS chunks.
// STAGED: the parameters of layers >= 1 are copied to LDS once at kernel start, so the small tail GEMMs and the
//         delta back-propagation never wait on global memory.
template <typename T, bool TRAIN, bool VECX, bool STAGED>
__global__ __launch_bounds__(kDenseThreads) void k_dense_fwd(
    NetDesc nd, const T* __restrict__ params, const T* __restrict__ X, const T* __restrict__ Y,
    const int* __restrict__ idx, int B, T* __restrict__ acts, T* __restrict__ deltas,
    T* __restrict__ loss_part, T* __restrict__ out) {
    using acc_t = typename Mfma16<T>::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int L = nd.L;
    const int s0 = blockIdx.x * kTileS;
    int sumd = 0, maxd = 0;
    for (int j = 1; j <= L; ++j) { sumd += nd.dims[j]; maxd = nd.dims[j] > maxd ? nd.dims[j] : maxd; }

    // LDS carve (all in units of T)
    T* actT = smem;                                   // a_j tiles, j = 1..L : [d_j][kLd] at kLd*act_off[j]
    T* dT0 = actT + kLd * sumd;                       // delta ping
    T* dT1 = dT0 + kLd * maxd;                        // delta pong
    T* xs = dT1 + kLd * maxd;                         // per-wave staging [kChunkK][kLd]
    T* red = xs + kDenseWaves * kChunkK * kLd;        // [wave][mt][16][kLd]
    long long* rowoff = reinterpret_cast<long long*>(red + kDenseWaves * kMtp * kRedTile);  // 16 row bases (elements)
    T* lossred = reinterpret_cast<T*>(rowoff + kTileS);                                        // kDenseWaves scalars
    T* wsm = red + kDenseWaves * kMtp * kRedTile + 64;                                         // staged tail parameters
    const int tail0 = L > 1 ? nd.w_off[1] : nd.P;

    if (STAGED) {
        const int cnt = nd.P - tail0;
        for (int e = tid; e < cnt; e += kDenseThreads) wsm[e] = params[tail0 + e];
    }
    if (tid < kTileS) {
        // rows past the end of the batch alias the batch's last row (always a legal address) and are masked by value,
        // so that no load in the kernel is conditional (a select after an unconditional load keeps them all in flight)
        const int gs = s0 + tid < B ? s0 + tid : B - 1;
        rowoff[tid] = idx ? (long long)idx[gs] : (long long)gs;
    }
    __syncthreads();

    // ------------------------------------------------------------------ forward (rcn.rs:281-291 / 105-116)
    for (int j = 0; j < L; ++j) {
        const int K = nd.dims[j], M = nd.dims[j + 1];
        const T* __restrict__ Wj = params + nd.w_off[j];
        const T* Ws = (STAGED && j >= 1) ? wsm + (nd.w_off[j] - tail0) : Wj;     // LDS copy for the tail layers
        const T* bj = Ws + (size_t)M * K;
        const T* aPrev = actT + kLd * nd.act_off[j];          // valid for j >= 1
        T* aNext = actT + kLd * nd.act_off[j + 1];
        int kb, ke;
        wave_k_range(K, wave, kb, ke);
        for (int mbase = 0; mbase < M; mbase += 16 * kMtp) {
            acc_t acc[kMtp];
#pragma unroll
            for (int mt = 0; mt < kMtp; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
            if (j == 0 && VECX) {
                // lane (n, g) owns features k0+4g .. k0+4g+3 of row n as ONE 16-byte load and feeds them to four
                // MFMAs; the A fragment of MFMA i takes W[:, k0+4g+i], so both operands agree on k.
                using vec4 = typename Vec4<T>::type;
                const long long r = rowoff[n];
                const bool rv = s0 + n < B;
                constexpr int kVecGroups = VecGroups<T>::value;
                for (int kc = kb; kc < ke; kc += 16 * kVecGroups) {
                    vec4 xv[kVecGroups];
                    T wv[kVecGroups][4][kMtp];
#pragma unroll
                    for (int gi = 0; gi < kVecGroups; ++gi) {
                        const int k0 = kc + 16 * gi + 4 * g;
                        const int kk = k0 < ke ? k0 : kb;                 // K % 4 == 0 and kb % 4 == 0: whole vectors only
                        xv[gi] = *reinterpret_cast<const vec4*>(X + r * (long long)K + kk);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int mt = 0; mt < kMtp; ++mt) {
                                const int row = mbase + mt * 16 + n;
                                wv[gi][i][mt] = Wj[(size_t)(kk + i) * M + (row < M ? row : M - 1)];
                            }
                    }
#pragma unroll
                    for (int gi = 0; gi < kVecGroups; ++gi) {
                        const bool kin = kc + 16 * gi + 4 * g < ke;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int mt = 0; mt < kMtp; ++mt) {
                                const bool rowin = mbase + mt * 16 + n < M;
                                acc[mt] = Mfma16<T>::mfma((kin && rowin) ? wv[gi][i][mt] : (T)0, (kin && rv) ? xv[gi][i] : (T)0, acc[mt]);
                            }
                    }
                }
            } else if (j == 0) {
                // stream this wave's K-slice of the 16 feature rows through a private LDS chunk
                T* xw = xs + wave * kChunkK * kLd;
                const int kl = lane & 31, sh = lane >> 5;
                for (int kc = kb; kc < ke; kc += kChunkK) {
                    T v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const long long r = rowoff[2 * q + sh];
                        const int k = kc + kl;
                        const T xr = X[r * (long long)K + (k < ke ? k : kb)];
                        v[q] = (s0 + 2 * q + sh < B && k < ke) ? xr : (T)0;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) xw[kl * kLd + 2 * q + sh] = v[q];
                    const int rem = ke - kc;
                    const int nks = ((rem < kChunkK ? rem : kChunkK) + 3) >> 2;
                    for (int ks = 0; ks < nks; ++ks) {
                        const int k = kc + 4 * ks + g;
                        const T bv = xw[(4 * ks + g) * kLd + n];
#pragma unroll
                        for (int mt = 0; mt < kMtp; ++mt) {
                            const int row = mbase + mt * 16 + n;
                            const T av = (row < M && k < ke) ? Wj[(size_t)k * M + row] : (T)0;
                            acc[mt] = Mfma16<T>::mfma(av, bv, acc[mt]);
                        }
                    }
                }
            } else {
                for (int k0 = kb; k0 < ke; k0 += 4) {
                    const int k = k0 + g;
                    const T bv = (k < ke) ? aPrev[k * kLd + n] : (T)0;
#pragma unroll
                    for (int mt = 0; mt < kMtp; ++mt) {
                        const int row = mbase + mt * 16 + n;
                        const T av = (row < M && k < ke) ? Ws[(size_t)k * M + row] : (T)0;
                        acc[mt] = Mfma16<T>::mfma(av, bv, acc[mt]);
                    }
                }
            }
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            {
                const int mt = tid >> 8, o = tid & 255, s = o >> 4, ml = o & 15;
                const int m = mbase + mt * 16 + ml;
                if (m < M) {
                    const T z = sum_partials<T>(red, mt, s, ml) + bj[m];     // z = W.a + b   rcn.rs:287
                    const T a = sigmoid_ref(z);                               //               rcn.rs:289
                    aNext[m * kLd + s] = a;
                    const int gs = s0 + s;
                    if (gs < B) {
                        if (TRAIN && j + 1 < L) acts[(size_t)B * nd.act_off[j + 1] + (size_t)gs * M + m] = a;
                        if (!TRAIN && j + 1 == L) out[(size_t)gs * M + m] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!TRAIN) return;

    // ------------------------------------------------------------------ output delta (rcn.rs:299)
    {
        const int M = nd.dims[L];
        const T* aL = actT + kLd * nd.act_off[L];
        T lsum = 0;
        for (int e = tid; e < kTileS * M; e += kDenseThreads) {
            const int s = e / M, m = e - s * M;
            const long long r = rowoff[s];
            T d = 0;
            if (s0 + s < B) {
                const T a = aL[m * kLd + s];
                const T diff = a - Y[r * (long long)M + m];
                d = diff * (a * ((T)1 - a));                 // (a_L - y) (*) sigmoid'(z_L), sigmoid' = s(1-s)  rcn.rs:491
                deltas[(size_t)B * nd.act_off[L] + (size_t)(s0 + s) * M + m] = d;
                lsum += diff * diff;
            }
            dT0[m * kLd + s] = d;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
        if (lane == 0) lossred[wave] = lsum;
    }
    __syncthreads();
    if (tid == 0 && loss_part) {
        T t = lossred[0];
        for (int w = 1; w < kDenseWaves; ++w) t += lossred[w];
        loss_part[blockIdx.x] = t;
    }

    // ------------------------------------------------------------------ hidden deltas (rcn.rs:305-309)
    T* dCur = dT0;
    T* dNxt = dT1;
    for (int j = L - 1; j >= 1; --j) {
        const int M = nd.dims[j], K = nd.dims[j + 1];          // delta_j = (W_j^T delta_{j+1}) (*) s'(z_j)
        const T* Wj = STAGED ? wsm + (nd.w_off[j] - tail0) : params + nd.w_off[j];   // K x M column-major: (k, m) at m*K + k
        const T* aJ = actT + kLd * nd.act_off[j];
        int kb, ke;
        wave_k_range(K, wave, kb, ke);
        for (int mbase = 0; mbase < M; mbase += 16 * kMtp) {
            acc_t acc[kMtp];
#pragma unroll
            for (int mt = 0; mt < kMtp; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
            for (int k0 = kb; k0 < ke; k0 += 4) {
                const int k = k0 + g;
                const T bv = (k < ke) ? dCur[k * kLd + n] : (T)0;
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt) {
                    const int row = mbase + mt * 16 + n;
                    const T av = (row < M && k < ke) ? Wj[(size_t)row * K + k] : (T)0;
                    acc[mt] = Mfma16<T>::mfma(av, bv, acc[mt]);
                }
            }
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            {
                const int mt = tid >> 8, o = tid & 255, s = o >> 4, ml = o & 15;
                const int m = mbase + mt * 16 + ml;
                if (m < M) {
                    const T a = aJ[m * kLd + s];
                    const int gs = s0 + s;
                    T d = 0;
                    if (gs < B) {
                        d = sum_partials<T>(red, mt, s, ml) * (a * ((T)1 - a));
                        deltas[(size_t)B * nd.act_off[j] + (size_t)gs * M + m] = d;
                    }
                    dNxt[m * kLd + s] = d;
                }
            }
            __syncthreads();
        }
        T* t = dCur; dCur = dNxt; dNxt = t;
    }
}

// The tail parameters of the one-hidden-layer shape class as k_p2_b's finishing wave wants them (dense_p2.hpp): an image
// [28 words][64 lanes] of ready-made MFMA operand fragments -- words 0..7 W_1 as A[m = c][k = h], 8..15 W_1^T as
// A[m = h][k = c], 16..19 b_1 and 24..27 b_0 per accumulator / slab element (20..23, the targets, are per batch and not
// part of the image).  One updated parameter lands in every place it occupies; pads stay the zeros the image starts with.
// f32 accumulator map (row = 4 (lane >> 4) + i) only.
__device__ inline void p2_frag_scatter(int jl, int cc, int m, int H, float v, float* __restrict__ img) {
    if (jl == 1 && cc < H) {
        const int h = cc, c = m;
        img[(h >> 2) * 64 + (h & 3) * 16 + c] = v;
        img[(8 + (h >> 4) * 4 + (c >> 2)) * 64 + (c & 3) * 16 + (h & 15)] = v;
    } else if (jl == 1) {
        const int c = m;                                                          // b_1[c]
#pragma unroll
        for (int nn = 0; nn < 16; ++nn) img[(16 + (c & 3)) * 64 + (c >> 2) * 16 + nn] = v;
    } else {
        const int h = m;                                                          // b_0[h] (bias column of W_0)
#pragma unroll
        for (int s = 0; s < 8; ++s) img[(24 + (h & 3)) * 64 + (h >> 2) + 8 * s] = v;
    }
}
__device__ inline void p2_frag_scatter(int, int, int, int, double, double*) {}    // the f64 context builds its fragments in the kernel

// Two hidden layers (the reference's own test net 784-10-10-10, rcn.rs:558,577; resident kernel only): the image grows by the
// third dense layer's parameters -- words 28..31 W_2 as A[m = c][k = h2], 32..35 W_2^T as A[m = h2][k = c], 36..39 b_2 per
// accumulator element -- and layer 1 of p2_frag_scatter is then the 32 -> 16 layer in the middle (its "classes" are the h2 units).
constexpr int kP3BFrag = 40;
__device__ inline void p3_frag_scatter(int cc, int m, int H2, float v, float* __restrict__ img) {
    if (cc < H2) {
        const int h2 = cc, c = m;
        img[(28 + (h2 >> 2)) * 64 + (h2 & 3) * 16 + c] = v;
        img[(32 + (c >> 2)) * 64 + (c & 3) * 16 + h2] = v;
    } else {
        const int c = m;                                                          // b_2[c]
#pragma unroll
        for (int nn = 0; nn < 16; ++nn) img[(36 + (c & 3)) * 64 + (c >> 2) * 16 + nn] = v;
    }
}

// One 16-column tile (columns n0..n0+15) of [W_j | b_j]: dW = Delta_{j+1} . [A_j | 1]^T summed over the whole batch
// (the MFMA contraction index is the sample), then either the SGD update or the raw gradient.
template <typename T, bool APPLY>
__device__ inline void wgrad_tile_ld(const NetDesc& nd, int j, int n0, T* __restrict__ params, T* __restrict__ grad_out,
                                     const T* __restrict__ Aprev, long long ldAin, const int* __restrict__ idx,
                                     const T* __restrict__ D, int ldD, int B, T scale, T* red, T* __restrict__ fragimg = nullptr) {
    using acc_t = typename Mfma16<T>::acc_t;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int Kin = nd.dims[j], M = nd.dims[j + 1];
    const int c = n0 + n;                                                        // this lane's column of [W|b]
    // a bias-only tile may be called without the activation rows (Aprev == nullptr): its loads (never used, every
    // column is >= Kin) are pointed at the delta buffer instead, so that they stay unconditional and legal
    const T* __restrict__ Ald = Aprev ? Aprev : D;
    const long long ldA = Aprev ? ldAin : 0;
    int kb, ke;
    wave_k_range(B, wave, kb, ke);
    for (int mbase = 0; mbase < M; mbase += 16 * kMtp) {
        acc_t acc[kMtp];
#pragma unroll
        for (int mt = 0; mt < kMtp; ++mt) acc[mt] = acc_t{0, 0, 0, 0};
        // 8 k-steps (32 samples) per chunk: every index / activation / delta load of the chunk is issued before the
        // first MFMA, so a wave pays the memory latency once per chunk instead of once per k-step
        for (int kc = kb; kc < ke; kc += 32) {
            long long r[8];
            T bv[8], av[8][kMtp];
            const int cc_ld = Aprev ? (c < Kin ? c : Kin - 1) : 0;   // clamped column: loads are unconditional, masked by value
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = kc + 4 * q + g;                  // contraction index = sample
                const int sc = s < ke ? s : kb;
                r[q] = idx ? (long long)idx[sc] : (long long)sc;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = kc + 4 * q + g;
                const int sc = s < ke ? s : kb;
                bv[q] = Ald[r[q] * ldA + cc_ld];
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt) {
                    const int row = mbase + mt * 16 + n;
                    av[q][mt] = D[(size_t)sc * ldD + (row < M ? row : M - 1)];
                }
            }
            __builtin_amdgcn_sched_barrier(0);                 // all loads of the chunk in flight before the first MFMA waits on one
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool sv = kc + 4 * q + g < ke;
                // bias column: the activation is the constant 1, so db = sum_s delta  (rcn.rs:302,309)
                const T b = sv ? (c < Kin ? bv[q] : (c == Kin ? (T)1 : (T)0)) : (T)0;
#pragma unroll
                for (int mt = 0; mt < kMtp; ++mt)
                    acc[mt] = Mfma16<T>::mfma((sv && mbase + mt * 16 + n < M) ? av[q][mt] : (T)0, b, acc[mt]);
            }
        }
        store_partials<T>(red, wave, lane, acc);
        __syncthreads();
        {
            const int mt = tid >> 8, o = tid & 255, cl = o >> 4, ml = o & 15;
            const int m = mbase + mt * 16 + ml, cc = n0 + cl;
            if (m < M && cc <= Kin) {
                const T gsum = sum_partials<T>(red, mt, cl, ml);
                const size_t p = (size_t)nd.w_off[j] + (size_t)cc * M + m;
                if (APPLY) {
                    const T nv = params[p] - scale * gsum;                      // rcn.rs:214,221
                    params[p] = nv;
                    if (fragimg) p2_frag_scatter(j, cc, m, nd.dims[1], nv, fragimg);
                } else grad_out[p] = gsum;
            }
        }
        __syncthreads();
    }
}

// the same tile on the unpadded [B][d] activation / delta images the sample-tile and generic pipeline kernels use
template <typename T, bool APPLY>
__device__ inline void wgrad_tile(const NetDesc& nd, int j, int n0, T* __restrict__ params, T* __restrict__ grad_out,
                                  const T* __restrict__ X, const int* __restrict__ idx, const T* __restrict__ acts,
                                  const T* __restrict__ deltas, int B, T scale, T* red) {
    const T* Aprev = (j == 0) ? X : acts + (size_t)B * nd.act_off[j];           // a_j: [B][Kin]
    wgrad_tile_ld<T, APPLY>(nd, j, n0, params, grad_out, Aprev, nd.dims[j], (j == 0) ? idx : nullptr,
                            deltas + (size_t)B * nd.act_off[j + 1], nd.dims[j + 1], B, scale, red);
}

// quadratic cost 1/(2B) sum ||a_L - y||^2 (what rcn.rs:299 is the gradient of); fixed-order sum by one thread
template <typename T>
__device__ inline void finish_loss(const T* __restrict__ loss_part, int n_loss, T loss_scale, T* __restrict__ loss_out) {
    T t = 0;
    for (int i = 0; i < n_loss; ++i) t += loss_part[i];
    *loss_out = t * loss_scale;
}

// grid = nd.tile_start[L] workgroups; workgroup -> (layer j, 16 columns of [W_j | b_j])
template <typename T, bool APPLY>
__global__ __launch_bounds__(kDenseThreads) void k_dense_wgrad(
    NetDesc nd, T* __restrict__ params, T* __restrict__ grad_out, const T* __restrict__ X,
    const int* __restrict__ idx, const T* __restrict__ acts, const T* __restrict__ deltas, int B, T scale,
    const T* __restrict__ loss_part, int n_loss, T loss_scale, T* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    int j = 0;
    while (j + 1 < nd.L && (int)blockIdx.x >= nd.tile_start[j + 1]) ++j;
    const int n0 = ((int)blockIdx.x - nd.tile_start[j]) * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) finish_loss<T>(loss_part, n_loss, loss_scale, loss_out);
    wgrad_tile<T, APPLY>(nd, j, n0, params, grad_out, X, idx, acts, deltas, B, scale, red);
}

// p <- p - scale * g  (the update half of train_batch when gradients were all-reduced first)
// get_expected_vec (rcn.rs:466-471) for a whole resident set: y[i][c] = (c == labels[i])
template <typename T>
__global__ void k_one_hot(const int32_t* __restrict__ labels, size_t n, int C, T* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * (size_t)C; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (int)(i % (size_t)C) == labels[i / (size_t)C] ? (T)1 : (T)0;
}

template <typename T>
__global__ void k_apply_gradient(T* __restrict__ p, const T* __restrict__ gsrc, T scale, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] - scale * gsrc[i];
}

// fixed-order sum of per-tile loss partials
template <typename T>
__global__ void k_sum_loss(const T* __restrict__ part, int n, T scale, T* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        T t = 0;
        for (int i = 0; i < n; ++i) t += part[i];
        *out = t * scale;
    }
}

// rcn.rs:152-157: accept iff one-hot(v == max) equals the expectation vector
template <typename T>
__global__ void k_eval_accept(const T* __restrict__ outv, const T* __restrict__ y, int n, int C, unsigned long long* count) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    int ok = 0;
    if (s < n) {
        const T* o = outv + (size_t)s * C;
        const T* e = y + (size_t)s * C;
        T mx = o[0];
        for (int i = 1; i < C; ++i) mx = o[i] > mx ? o[i] : mx;
        ok = 1;
        for (int i = 0; i < C; ++i) {
            const T oh = (o[i] == mx) ? (T)1 : (T)0;
            if (oh != e[i]) ok = 0;
        }
    }
    const unsigned long long b = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

// rcn.rs:92-97: max_by(total_cmp) -> index of the LAST maximal element
template <typename T>
__global__ void k_argmax_last(const T* __restrict__ outv, int n, int C, int* __restrict__ cls) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const T* o = outv + (size_t)s * C;
    int best = 0;
    for (int i = 1; i < C; ++i)
        if (!(o[i] < o[best])) best = i;
    cls[s] = best;
}

}  // namespace rcn
