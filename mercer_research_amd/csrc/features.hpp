// features.hpp -- fused conv/pool feature pipeline of RCN::flatten_feature_set (rcn.rs:317-356) on gfx950.
//
// One workgroup walks whole images through the layer stack with every intermediate map resident in LDS
// (two ping-pong buffers); nothing but the u8 image is read from HBM and nothing but the final feature
// vector is written.  All four separable Sobel operators (kernel.rs:38-53) of a pixel are evaluated from
// one 3x3 neighbourhood; the column pass / row pass / ReLU of convolve_2d_separated (kernel.rs:196-207)
// collapse into that stencil, with the Padding::Same pad-copy index quirk (kernel.rs:154-158)
// folded into the neighbourhood masks.  Inputs are u8 and every coefficient a small integer, so the
// arithmetic is exact in f32 up to 5 conv layers (|v| <= 255*8^n < 2^24) and in f64 beyond; the result
// is therefore bit-identical to the reference's f64 loop regardless of summation order.
#pragma once

#include <cmath>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace rcn {

constexpr int kFeatThreads = 256;

// Output slot of (input map i, operator o) when a conv layer expands `cnt` maps (rcn.rs:323-340).
// SEP_OPS = [Top, Left, Right, Bottom] (rcn.rs:41-46); o indexes that array.
__device__ inline int conv_out_slot(int cnt_in, int i, int o) {
    if (cnt_in == 0) return o;                    // first conv: [T, L, R, B]          rcn.rs:339
    if (o == 3) return i;                         // last op (Bottom) replaces slot i   rcn.rs:332
    return cnt_in + 3 * i + o;                    // the others are pushed in order     rcn.rs:334
}

// TC: compute type (float exact up to 5 conv layers, else double); TO: output type (ctx dtype)
// `spill` (nullable): when the maps of one image do not fit LDS (large inputs, several un-pooled conv layers) the two ping-pong
// buffers live in global memory instead, 2 * max_elems values per workgroup -- same code, same results, the reads go through
// L1/L2 (a workgroup's own stores are visible to it after the barrier); the reference has no such size limit.
template <typename TC, typename TO>
__global__ __launch_bounds__(kFeatThreads) void k_features(
    FeatDesc fd, const uint8_t* __restrict__ imgs, int n_img, TO* __restrict__ out, int standardize, TO mean, TO sd, TC* __restrict__ spill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* bufA = spill ? spill + (size_t)blockIdx.x * 2 * fd.max_elems : reinterpret_cast<TC*>(smem_raw);
    TC* bufB = bufA + fd.max_elems;
    const int tid = threadIdx.x;
    const int HW = fd.H * fd.W;

    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
        // get_pixel_matrix (lib.rs:27-41): m[(y,x)] = pixel(x,y); kept row-major [y*W+x] in LDS
        const uint8_t* src = imgs + (size_t)img * HW;
        for (int e = tid; e < HW; e += kFeatThreads) bufA[e] = (TC)src[e];
        __syncthreads();

        TC* cur = bufA;
        TC* nxt = bufB;
        int cnt = 0, R = fd.H, C = fd.W;       // cnt == 0: `cur` holds the image, feature_set is empty
        for (int li = 0; li < fd.n; ++li) {
            if (fd.kind[li] == 0) {
                const bool same = fd.arg[li] == 1;
                const int oR = same ? R : R - 2, oC = same ? C : C - 2;
                const int n_in = cnt == 0 ? 1 : cnt, isz = R * C, osz = oR * oC;
                for (int e = tid; e < n_in * osz; e += kFeatThreads) {
                    const int i = e / osz, p = e - i * osz, y = p / oC, x = p - y * oC;
                    const TC* M = cur + i * isz;
                    TC v[3][3];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            int yy, xx;
                            bool ok;
                            if (same) {
                                // out[y][x] = sum_kx krow[kx] * tmp'[y-1][x+kx-1];  tmp[r][c] = sum_ky kcol[ky] * M'[r+ky-1][c-1]
                                // with the quirk masks: row y-1 of tmp only for y>=1, columns of M only 0..C-2
                                yy = y - 2 + ky; xx = x + kx - 2;
                                const int tc = x + kx - 1;
                                ok = (y >= 1) && (tc >= 0) && (tc < C) && (yy >= 0) && (yy < R) && (xx >= 0) && (xx <= C - 2);
                            } else {
                                yy = y + ky; xx = x + kx; ok = true;
                            }
                            v[ky][kx] = ok ? M[yy * C + xx] : (TC)0;
                        }
                    // column kernels [1,0,-1] (Top) / [1,2,1] (Left,Right); row kernels [1,2,1] / [1,0,-1]  kernel.rs:47-52
                    const TC top = (v[0][0] - v[2][0]) + (TC)2 * (v[0][1] - v[2][1]) + (v[0][2] - v[2][2]);
                    const TC left = (v[0][0] + (TC)2 * v[1][0] + v[2][0]) - (v[0][2] + (TC)2 * v[1][2] + v[2][2]);
                    const TC zero = (TC)0;
                    nxt[conv_out_slot(cnt, i, 0) * osz + p] = top > zero ? top : zero;        // Top
                    nxt[conv_out_slot(cnt, i, 1) * osz + p] = left > zero ? left : zero;      // Left
                    nxt[conv_out_slot(cnt, i, 2) * osz + p] = -left > zero ? -left : zero;    // Right  = -Left
                    nxt[conv_out_slot(cnt, i, 3) * osz + p] = -top > zero ? -top : zero;      // Bottom = -Top
                }
                cnt = n_in * 4; R = oR; C = oC;
            } else {
                if (cnt == 0) continue;          // pooling an empty feature_set is a no-op (rcn.rs:343)
                // Pool2D::pool_2d(Padding::Same, Max): zero-pad odd dims bottom/right (kernel.rs:253-260, 310-319)
                const int oR = (R + 1) / 2, oC = (C + 1) / 2, isz = R * C, osz = oR * oC;
                for (int e = tid; e < cnt * osz; e += kFeatThreads) {
                    const int i = e / osz, p = e - i * osz, y = p / oC, x = p - y * oC;
                    const TC* M = cur + i * isz;
                    const int y0 = 2 * y, x0 = 2 * x;
                    const bool yb = y0 + 1 < R, xb = x0 + 1 < C;
                    const TC a = M[y0 * C + x0];
                    const TC b = xb ? M[y0 * C + x0 + 1] : (TC)0;
                    const TC c = yb ? M[(y0 + 1) * C + x0] : (TC)0;
                    const TC d = (yb && xb) ? M[(y0 + 1) * C + x0 + 1] : (TC)0;
                    const TC m1 = a > b ? a : b, m2 = c > d ? c : d;
                    nxt[i * osz + p] = m1 > m2 ? m1 : m2;
                }
                R = oR; C = oC;
            }
            __syncthreads();
            TC* t = cur; cur = nxt; nxt = t;
        }

        // flatten: maps in slot order, each in column-major order (rcn.rs:350-355), then optionally
        // x <- max((x - mean)/sd, 0)  (rcn.rs:407-412)
        if (cnt > 0) {
            const int sz = R * C;
            TO* dst = out + (size_t)img * fd.F;
            for (int e = tid; e < cnt * sz; e += kFeatThreads) {
                const int i = e / sz, q = e - i * sz, x = q / R, y = q - x * R;      // q = x*R + y
                TO val = (TO)cur[i * sz + y * C + x];
                if (standardize) {
                    const TO dd = (val - mean) / sd;
                    val = dd >= (TO)0 ? dd : (TO)0;
                }
                dst[e] = val;
            }
        }
        __syncthreads();
    }
}

// ---- (x - mean) / sd, clamped at zero (rcn.rs:407-412 / 86-89).
// The f32 quotient costs ~11 VALU instructions as an IEEE division.  When FAST, it is computed as Markstein's
// correction of a reciprocal product instead -- q0 = n*y, r = fma(-q0, sd, n), q = fma(r, y, q0) with y = RN(1/sd) --
// which is the correctly rounded quotient except for rare (n, sd) pairs.  "Rare" is not good enough for a bit-exact
// path, so the host only selects FAST after standardise_fast_is_exact() has compared the two forms over EVERY value a
// feature can take (integers 0..vmax) for the (mean, sd) in force; otherwise the kernels divide.  (In practice the
// check passes: 2000 of 2000 random scales.)
template <bool FAST, typename T>
__device__ inline T standardise_clamp(T v, T mean, T sd, T rcp) {
    const T n = v - mean;
    if constexpr (FAST) {
        const T q0 = n * rcp;
        const T r = __builtin_elementwise_fma(-q0, sd, n);
        const T q = __builtin_elementwise_fma(r, rcp, q0);
        return __builtin_elementwise_max(q, (T)0);       // == the select below: the host check also rules out q = -0 and NaN
    } else {
        const T q = n / sd;
        return q >= (T)0 ? q : (T)0;
    }
}

// two features at once: the FAST f32 form runs as packed fp32 (4 instructions + 2 max for the pair instead of 10)
template <bool FAST, typename T>
__device__ inline void standardise_clamp_pair(T& a, T& b, T mean, T sd, T rcp) {
    if constexpr (FAST && std::is_same<T, float>::value) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {a, b}, m = {mean, mean}, s = {sd, sd}, y = {rcp, rcp};
        const f2 n = v - m;
        const f2 q0 = n * y;
        const f2 r = __builtin_elementwise_fma(-q0, s, n);
        const f2 q = __builtin_elementwise_fma(r, y, q0);
        a = __builtin_elementwise_max(q.x, 0.f);
        b = __builtin_elementwise_max(q.y, 0.f);
    } else {
        a = standardise_clamp<FAST>(a, mean, sd, rcp);
        b = standardise_clamp<FAST>(b, mean, sd, rcp);
    }
}

inline bool standardise_fast_is_exact(float mean, float sd, int vmax, float* rcp_out) {
    if (!(sd > 0.f) || !std::isfinite(sd) || !std::isfinite(mean)) return false;
    const volatile float y = 1.0f / sd;                              // correctly rounded reciprocal
    for (int v = 0; v <= vmax; ++v) {
        const volatile float n = (float)v - mean;
        const volatile float q0 = n * y;                             // volatile: no host-side contraction into an fma
        const float r = std::fmaf(-q0, sd, n);
        const float q = std::fmaf(r, y, q0), want = n / sd;
        if (std::memcmp(&q, &want, sizeof q) != 0) return false;
        if (q != q || (q == 0.f && std::signbit(q))) return false;  // max(q, +0) and (q >= 0 ? q : 0) differ only for NaN and -0
    }
    *rcp_out = y;
    return true;
}

// ---- specialisation: the default stack conv(Same), pool(Max), conv(Same), pool(Max) (rcn/src/main.rs:53-59) on an
// H x W image with H, W multiples of 4 known at compile time (MNIST: 28 x 28).
//
// Same arithmetic as k_features (integer-valued f32, exact), restructured for throughput:
//  * one WAVE owns one image at a time (workgroup = 1 wave: barriers are free, no cross-wave coupling), ~20 waves per
//    CU resident, 8 KB of LDS each;
//  * each conv+pool pair is fused: one work item produces one POOLED pixel of all four operator maps from a 4x4 input
//    window (the four 3x3 neighbourhoods of its 2x2 pooling cell share column sums / differences);
//  * the Padding::Same quirk masks (kernel.rs:154-158: a row/column of zeros ends up top/left and the last input
//    column is never read; output row 0 is zero) become LAYOUT: maps sit in LDS inside a frame of zeros (two rows
//    above, two columns left) with their last column stored as zero, so the stencil runs with no per-element
//    predicates and every window row is two aligned ds_read_b64;
//  * ReLU followed by max-pool is max(+0, max4(.)); Right = -Left and Bottom = -Top come from the negated operands;
//  * all index arithmetic divides by compile-time constants.
// HBM traffic is the algorithmic minimum: H*W bytes in, F*sizeof(TO) out.
template <int H, int W>
struct Cpcp {
    static_assert(H % 4 == 0 && W % 4 == 0, "two exact 2x2 poolings and 4-pixel word loads");
    static constexpr int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4;
    static constexpr int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
    static constexpr int pad_to(int x, int mod, int res) { while (x % mod != res % mod) ++x; return x; }
    // Work items are laid over the lanes in power-of-two runs -- stage 1: RW1 lanes per pooled row (W1 of them live),
    // stage 2: RH2 lanes per pooled column (H2 live) -- and the padded strides are chosen so that each 32-lane group of
    // a ds_read_b64 (bank = dword address mod 64, MI355X_MICROARCH.md LDS table) touches 32 distinct bank pairs:
    //   stage 1: lane (py, px) reads dword 2*PW0*py + 2*px (+ row const): a group is 32/RW1 rows of 2*RW1 banks, so the
    //            rows must sit 2*RW1 banks apart:                PW0 == RW1 (mod 32)           28x28: PW0 = 48
    //   stage 2: lane (i, px, py) reads dword N1*i + 2*PW1*py + 2*px: a group is 32/RH2 columns, 2 banks each, so py must
    //            step by an odd multiple of 64/RH2 banks:        PW1 == 32/RH2 (mod 64/RH2)    28x28: PW1 = 20
    //            and a group that straddles two maps keeps the column sequence going:
    //                                                            N1 == 2*W2 (mod 64/RH2)       28x28: N1 = 326
    // (lanes past the live ones re-read a live lane's address -- a broadcast, not a conflict).  Measured on 28x28:
    // SQ_LDS_BANK_CONFLICT was 70 % of all LDS cycles with the natural strides 32 / 16 / 256.
    static constexpr int RW1 = pow2ceil(W1), RH2 = pow2ceil(H2);
    static_assert(RW1 <= 32 && RH2 <= 32, "one run of lanes per 32-lane group at most");
    static constexpr int PW0 = pad_to(W + 4 > 2 * RW1 + 2 ? W + 4 : 2 * RW1 + 2, 32, RW1);      // frame: 2 left; right slack keeps every lane's window in range
    static constexpr int PW1 = pad_to(W1 + 2 > 2 * W2 + 2 ? W1 + 2 : 2 * W2 + 2, 64 / RH2, 32 / RH2);
    static constexpr int N0 = (H + 2) * PW0, N1 = pad_to((H1 + 2) * PW1, 64 / RH2 > 2 ? 64 / RH2 : 2, 2 * W2);
    static_assert(N1 % 2 == 0 && PW0 % 2 == 0 && PW1 % 2 == 0, "8-byte aligned window rows");
    static constexpr int WORDS = H * W / 4, ROWW = W / 4;
    static constexpr int S1 = H1 * RW1, S2 = 4 * W2 * RH2;      // lane slots of the two stages
    static constexpr int SZ2 = H2 * W2, F = 16 * SZ2;
    static constexpr int LDS_FLOATS = N0 + 4 * N1;
    static constexpr int VMAX = 255 * 16;                        // two [1,2,1]x[1,0,-1] passes: |v| <= 255 * 4 * 4

    // the frame of zeros; once per workgroup (image writes never touch it)
    template <int NT>
    __device__ static inline void init(float* P0, float* P1, int tid) {
        for (int e = tid; e < N0; e += NT) P0[e] = 0.f;
        for (int e = tid; e < 4 * N1; e += NT) P1[e] = 0.f;
    }

    // Stage boundary.  With one wave per workgroup the LDS serves that wave's instructions in issue order, so a write is
    // visible to every later read without a barrier; only the compiler must not reorder.  (__syncthreads() here would
    // also carry a workgroup-scope release -- an s_waitcnt vmcnt(0) that drains the previous picture's 16 global stores
    // and the prefetched loads once per picture: measured, that wait was the largest stall in the kernel.)
    // More than one wave per picture: the stage boundary is a real barrier, but an LDS-only one -- every LDS operation of this wave
    // retired (lgkmcnt), then s_barrier; the global stores of the previous picture and the prefetched loads stay in flight.
    template <int NT>
    __device__ static inline void sync() {
        if constexpr (NT <= 64) asm volatile("" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // one 4x4 window -> the four pooled operator responses (Top, Left, Right, Bottom) of its 2x2 cell.
    // Every value is an integer below 2^24, so f32 arithmetic is exact in any association and a fused multiply-add
    // equals the reference's multiply then add; that licence is used to run the column pass on the window's aligned
    // (even, odd) column pairs as packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two lanes of work per VALU issue slot).
    template <int STRIDE>
    __device__ static inline void cell(const float* win, bool row0, float& tmax, float& lmax, float& rmax, float& bmax) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 a[4], b[4];                                                   // a[r] = columns 0,1 of window row r;  b[r] = columns 2,3
        // Eight single ds_read_b64 (2 LDS cycles each, 64 banks).  Written as asm because the compiler fuses each
        // (a[r], b[r]) pair into one ds_read2_b64, which the LDS serves at half that rate and with mod-32 banking.
        const unsigned addr = (unsigned)(uintptr_t)win;                  // LDS byte address: low half of the generic pointer
#define RCN_LDS_RD(dst, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
        RCN_LDS_RD(a[0], 0 * STRIDE * 4);     RCN_LDS_RD(b[0], 0 * STRIDE * 4 + 8);
        RCN_LDS_RD(a[1], 1 * STRIDE * 4);     RCN_LDS_RD(b[1], 1 * STRIDE * 4 + 8);
        RCN_LDS_RD(a[2], 2 * STRIDE * 4);     RCN_LDS_RD(b[2], 2 * STRIDE * 4 + 8);
        RCN_LDS_RD(a[3], 3 * STRIDE * 4);     RCN_LDS_RD(b[3], 3 * STRIDE * 4 + 8);
#undef RCN_LDS_RD
        // the loads are asynchronous: tie every destination to the wait so no use is scheduled above it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]) : : "memory");
        const f2 two = {2.f, 2.f};
        tmax = 0.f; bmax = 0.f; lmax = 0.f; rmax = 0.f;          // the +0 is the ReLU floor; Bottom = -Top, Right = -Left
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const f2 d01 = a[dy] - a[dy + 2], d23 = b[dy] - b[dy + 2];                                   // column kernel [1,0,-1]   kernel.rs:47
            const f2 s01 = __builtin_elementwise_fma(two, a[dy + 1], a[dy] + a[dy + 2]);                 // column kernel [1,2,1]    kernel.rs:48
            const f2 s23 = __builtin_elementwise_fma(two, b[dy + 1], b[dy] + b[dy + 2]);
            const f2 mid = __builtin_shufflevector(d01, d23, 1, 2);                                      // (d1, d2)
            f2 top = __builtin_elementwise_fma(two, mid, d01 + d23);                                     // row kernel [1,2,1] at dx = 0, 1
            f2 left = s01 - s23;                                                                         // row kernel [1,0,-1] at dx = 0, 1
            if (dy == 0) {                                               // output row 0 is zero (quirk): x*m + (+0) with m = 0 or 1 -- one packed
                const float m1 = row0 ? 0.f : 1.f;                       // fma per pair, and (-x)*0 + (+0) = +0, so no -0 reaches the max chain
                const f2 m = {m1, m1}, z = {0.f, 0.f};
                top = __builtin_elementwise_fma(top, m, z);
                left = __builtin_elementwise_fma(left, m, z);
            }
            tmax = fmaxf(fmaxf(tmax, top.x), top.y); bmax = fmaxf(fmaxf(bmax, -top.x), -top.y);
            lmax = fmaxf(fmaxf(lmax, left.x), left.y); rmax = fmaxf(fmaxf(rmax, -left.x), -left.y);
        }
    }

    // One image through conv,pool,conv,pool by NT threads (a whole workgroup: contains barriers).  emit(ea, eb, va, vb): features ea, eb
    // of the flattened vector (rcn.rs:350-355 order) has the integer value v.
    // get_pixel_matrix (lib.rs:27-41), first half: this thread's 32-bit words of the picture (4 pixels of one row each).
    // Separate from image() so a caller can issue the NEXT picture's loads before working on the current one.
    template <int NT>
    struct Words { uint32_t v[(WORDS + NT - 1) / NT]; };
    template <int NT>
    __device__ static inline Words<NT> load_words(const uint8_t* img, int tid) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(img);
        Words<NT> w;
#pragma unroll
        for (int k = 0; k < (WORDS + NT - 1) / NT; ++k) {
            const int wd = tid + NT * k;
            w.v[k] = src[wd < WORDS ? wd : WORDS - 1];
        }
        return w;
    }

    template <int NT, typename Emit>
    __device__ static inline void image(float* P0, float* P1, const uint8_t* img, int tid, Emit emit) {
        image<NT>(P0, P1, load_words<NT>(img, tid), tid, emit);
    }

    template <int NT, typename Emit>
    __device__ static inline void image(float* P0, float* P1, const Words<NT>& words, int tid, Emit emit) {
        // image pixel (r,c) -> P0[(r+2)*PW0 + c+2]
#pragma unroll
        for (int k = 0; k < (WORDS + NT - 1) / NT; ++k) {
            const int wd = tid + NT * k;
            const int wc = wd < WORDS ? wd : WORDS - 1;
            const uint32_t v = words.v[k];
            const int r = wc / ROWW, c0 = (wc - r * ROWW) * 4;
            float2 lo, hi;
            lo.x = (float)(v & 255u); lo.y = (float)((v >> 8) & 255u);
            hi.x = (float)((v >> 16) & 255u); hi.y = c0 + 3 == W - 1 ? 0.f : (float)(v >> 24);     // last column: never read (quirk)
            float* dst = &P0[(r + 2) * PW0 + c0 + 2];
            if (wd < WORDS) {
                *reinterpret_cast<float2*>(dst) = lo;
                *reinterpret_cast<float2*>(dst + 2) = hi;
            }
        }
        sync<NT>();
        // conv1 + pool1: item = pooled pixel (py,px) of the H1 x W1 maps, RW1 lanes per row
#pragma unroll
        for (int k = 0; k < (S1 + NT - 1) / NT; ++k) {
            const int t = tid + NT * k;
            const int pyr = t / RW1, px = t % RW1;
            const int py = pyr < H1 ? pyr : H1 - 1;
            float tmax, lmax, rmax, bmax;
            cell<PW0>(&P0[2 * py * PW0 + 2 * px], NT * k < RW1 && py == 0, tmax, lmax, rmax, bmax);   // later rounds: py > 0 for every lane
            if (pyr < H1 && px < W1 - 1) {                                    // conv2 never reads a map's last column: it keeps init()'s zeros
                float* q = &P1[(py + 2) * PW1 + px + 2];
                q[0 * N1] = tmax;                                             // Top      (SEP_OPS order rcn.rs:41-46)
                q[1 * N1] = lmax;                                             // Left
                q[2 * N1] = rmax;                                             // Right
                q[3 * N1] = bmax;                                             // Bottom
            }
        }
        sync<NT>();
        // conv2 + pool2 + flatten: item = (input map i, pooled pixel), RH2 lanes per pooled column so that live lanes run
        // down the columns in the flattened vector's order (column-major within a map, rcn.rs:350-355)
#pragma unroll
        for (int k = 0; k < (S2 + NT - 1) / NT; ++k) {
            const int t = tid + NT * k;
            const int cgr = t / RH2, pyr = t % RH2;
            const int cg = cgr < 4 * W2 ? cgr : 4 * W2 - 1, py = pyr < H2 ? pyr : H2 - 1;
            const int i = cg / W2, px = cg - i * W2, q = px * H2 + py;
            float tmax, lmax, rmax, bmax;
            cell<PW1>(&P1[i * N1 + 2 * py * PW1 + 2 * px], py == 0, tmax, lmax, rmax, bmax);
            if (cgr < 4 * W2 && pyr < H2) {
                // slots after the second conv layer (rcn.rs:323-340): Bottom stays in slot i, T/L/R are pushed to 4+3i+o
                emit((4 + 3 * i + 0) * SZ2 + q, (4 + 3 * i + 1) * SZ2 + q, tmax, lmax);
                emit((4 + 3 * i + 2) * SZ2 + q, i * SZ2 + q, rmax, bmax);
            }
        }
    }
};

// NT = 64: one wave per picture (no barriers at all).  NT = 128: two waves share a picture's LDS image and split every stage's work
// items -- the same 11 KB of LDS then carries twice the waves (28 per CU instead of 14), which is what hides the LDS and VALU
// latencies of the other waves; stage boundaries are LDS-only barriers (Cpcp::sync).
template <int H, int W, typename TO, bool STD, bool FAST, int NT = 64>
__global__ __launch_bounds__(NT) void k_features_cpcp(const uint8_t* __restrict__ imgs, int n_img, TO* __restrict__ out, TO mean, TO sd, TO rcp) {
    using K = Cpcp<H, W>;
    __shared__ __attribute__((aligned(16))) float P0[K::N0];
    __shared__ __attribute__((aligned(16))) float P1[4 * K::N1];
    const int lane = threadIdx.x;
    K::template init<NT>(P0, P1, lane);
    __syncthreads();
    // the next picture's pixels are in flight while this one is computed
    auto nxt = K::template load_words<NT>(imgs + (size_t)(blockIdx.x < n_img ? blockIdx.x : 0) * (H * W), lane);
    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
        TO* dst = out + (size_t)img * K::F;
        const auto cur = nxt;
        const int ni = img + (int)gridDim.x;
        nxt = K::template load_words<NT>(imgs + (size_t)(ni < n_img ? ni : img) * (H * W), lane);
        K::template image<NT>(P0, P1, cur, lane, [&](int ea, int eb, float fa, float fb) {
            TO va = (TO)fa, vb = (TO)fb;
            if constexpr (STD) standardise_clamp_pair<FAST>(va, vb, mean, sd, rcp);
            dst[ea] = va;
            dst[eb] = vb;
        });
        K::template sync<NT>();
    }
}

// flatten_feature_set + standardise of a whole epoch segment, written STRAIGHT into the slice-major image the pipelined
// training kernels read (dense_pipe.hpp: Xs[batch][slice][sample][16 features], Ys[batch][sample][classes]) -- the
// end-to-end form: u8 images in, no [N][F] feature matrix in HBM, no separate gather pass.  One wave per (batch, sample);
// the sample's image is perm[batch * B + sample] (or that index itself).  Same arithmetic as k_features_cpcp with the
// fused standardisation, so the packed values are bit-identical to features -> standardise -> k_pack_epoch.
template <int H, int W, typename TO, bool FAST>
__global__ __launch_bounds__(64) void k_features_cpcp_packed(const uint8_t* __restrict__ imgs, const TO* __restrict__ Y, const int* __restrict__ perm,
                                                             int B, int n_batches, int G, int C, TO mean, TO sd, TO rcp, TO* __restrict__ xs,
                                                             TO* __restrict__ ys) {
    using K = Cpcp<H, W>;
    __shared__ __attribute__((aligned(16))) float P0[K::N0];
    __shared__ __attribute__((aligned(16))) float P1[4 * K::N1];
    const int lane = threadIdx.x;
    K::template init<64>(P0, P1, lane);
    __syncthreads();
    const int total = n_batches * B;
    // two-deep software pipeline: the picture index of step L+2 and the pixels of step L+1 are in flight during step L
    const int g = (int)gridDim.x, L0 = blockIdx.x;
    auto pick = [&](int L) -> long long { const int Lc = L < total ? L : L0; return perm ? perm[Lc] : Lc; };
    long long img_n = L0 < total ? pick(L0) : 0;
    auto nxt = K::template load_words<64>(imgs + (size_t)img_n * (H * W), lane);
    long long img_nn = pick(L0 + g);
    for (int L = L0; L < total; L += g) {
        const int jb = L / B, smp = L - jb * B;
        const long long img = img_n;
        const auto cur = nxt;
        img_n = img_nn;
        nxt = K::template load_words<64>(imgs + (size_t)img_n * (H * W), lane);
        img_nn = pick(L + 2 * g);
        TO* xb = xs + ((size_t)jb * G * B + smp) * 16;                 // + slice * B * 16 + feature % 16
        K::template image<64>(P0, P1, cur, lane, [&](int ea, int eb, float fa, float fb) {
            TO va = (TO)fa, vb = (TO)fb;
            standardise_clamp_pair<FAST>(va, vb, mean, sd, rcp);
            xb[(size_t)(ea >> 4) * B * 16 + (ea & 15)] = va;
            xb[(size_t)(eb >> 4) * B * 16 + (eb & 15)] = vb;
        });
        if (lane < C) ys[(size_t)L * C + lane] = Y[(size_t)img * C + lane];
        K::template sync<64>();
    }
}

// ---- gen_scales (rcn.rs:230-251): two-pass population mean / sd, f64 accumulation, per-block partials that
// the host sums in block order (deterministic).
template <typename T, bool SQDEV>
__global__ __launch_bounds__(256) void k_reduce(const T* __restrict__ x, size_t n, double mean, double* __restrict__ partial) {
    __shared__ double wsum[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        if (SQDEV) { const double d = v - mean; acc += d * d; } else acc += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// x <- max((x-mean)/sd, 0)   rcn.rs:407-412 / 86-89
template <typename T>
__global__ void k_standardize(T* __restrict__ x, size_t n, T mean, T sd) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T d = (x[i] - mean) / sd;
        x[i] = d >= (T)0 ? d : (T)0;
    }
}

}  // namespace rcn
