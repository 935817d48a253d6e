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
    static constexpr int PW0 = W + 4, PW1 = W1 + 2;          // padded row strides (frame: 2 left; right slack keeps windows in range)
    static constexpr int N0 = (H + 2) * PW0, N1 = (H1 + 2) * PW1;
    static constexpr int WORDS = H * W / 4, ROWW = W / 4;
    static constexpr int I1 = H1 * W1, I2 = 4 * H2 * W2, SZ2 = H2 * W2, F = 16 * SZ2;
    static constexpr int LDS_FLOATS = N0 + 4 * N1;

    // the frame of zeros; once per workgroup (image writes never touch it)
    template <int NT>
    __device__ static inline void init(float* P0, float* P1, int tid) {
        for (int e = tid; e < N0; e += NT) P0[e] = 0.f;
        for (int e = tid; e < 4 * N1; e += NT) P1[e] = 0.f;
    }

    // one 4x4 window -> the four pooled operator responses (Top, Left, Right, Bottom) of its 2x2 cell
    __device__ static inline void cell(const float* win, int stride, bool row0, float& tmax, float& lmax, float& rmax, float& bmax) {
        float w[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 a = *reinterpret_cast<const float2*>(win + r * stride), b = *reinterpret_cast<const float2*>(win + r * stride + 2);
            w[r][0] = a.x; w[r][1] = a.y; w[r][2] = b.x; w[r][3] = b.y;
        }
        tmax = 0.f; bmax = 0.f; lmax = 0.f; rmax = 0.f;          // the +0 is the ReLU floor; Bottom = -Top, Right = -Left
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float d[4], sm[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d[c] = w[dy][c] - w[dy + 2][c];                          // column kernel [1,0,-1]   kernel.rs:47
                sm[c] = (w[dy][c] + w[dy + 2][c]) + 2.f * w[dy + 1][c];   // column kernel [1,2,1]    kernel.rs:48
            }
            const bool live = dy == 1 || !row0;                          // output row 0 is zero (quirk)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float top = (d[dx] + d[dx + 2]) + 2.f * d[dx + 1];        // row kernel [1,2,1]
                float left = sm[dx] - sm[dx + 2];                         // row kernel [1,0,-1]
                top = live ? top : 0.f; left = live ? left : 0.f;
                tmax = fmaxf(tmax, top); bmax = fmaxf(bmax, -top);
                lmax = fmaxf(lmax, left); rmax = fmaxf(rmax, -left);
            }
        }
    }

    // One image through conv,pool,conv,pool by NT threads (a whole workgroup: contains barriers).  emit(e, v): feature e
    // of the flattened vector (rcn.rs:350-355 order) has the integer value v.
    template <int NT, typename Emit>
    __device__ static inline void image(float* P0, float* P1, const uint8_t* img, int tid, Emit emit) {
        // get_pixel_matrix (lib.rs:27-41): 4 pixels of one row per 32-bit word; image pixel (r,c) -> P0[(r+2)*PW0 + c+2]
        const uint32_t* src = reinterpret_cast<const uint32_t*>(img);
#pragma unroll
        for (int k = 0; k < (WORDS + NT - 1) / NT; ++k) {
            const int wd = tid + NT * k;
            const int wc = wd < WORDS ? wd : WORDS - 1;
            const uint32_t v = src[wc];
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
        __syncthreads();
        // conv1 + pool1: item = pooled pixel (py,px) of the H1 x W1 maps
#pragma unroll
        for (int k = 0; k < (I1 + NT - 1) / NT; ++k) {
            const int t = tid + NT * k;
            const int tc = t < I1 ? t : I1 - 1;
            const int py = tc / W1, px = tc - py * W1;
            float tmax, lmax, rmax, bmax;
            cell(&P0[2 * py * PW0 + 2 * px], PW0, py == 0, tmax, lmax, rmax, bmax);
            if (t < I1) {
                const bool lastc = px == W1 - 1;                              // conv2 never reads a map's last column
                float* q = &P1[(py + 2) * PW1 + px + 2];
                q[0 * N1] = lastc ? 0.f : tmax;                               // Top      (SEP_OPS order rcn.rs:41-46)
                q[1 * N1] = lastc ? 0.f : lmax;                               // Left
                q[2 * N1] = lastc ? 0.f : rmax;                               // Right
                q[3 * N1] = lastc ? 0.f : bmax;                               // Bottom
            }
        }
        __syncthreads();
        // conv2 + pool2 + flatten: item = (input map i, pooled pixel) ordered column-major within the map (rcn.rs:350-355)
#pragma unroll
        for (int k = 0; k < (I2 + NT - 1) / NT; ++k) {
            const int t = tid + NT * k;
            const int tc = t < I2 ? t : I2 - 1;
            const int i = tc / SZ2, q = tc - i * SZ2, px = q / H2, py = q - px * H2;
            float tmax, lmax, rmax, bmax;
            cell(&P1[i * N1 + 2 * py * PW1 + 2 * px], PW1, py == 0, tmax, lmax, rmax, bmax);
            if (t < I2) {
                // slots after the second conv layer (rcn.rs:323-340): Bottom stays in slot i, T/L/R are pushed to 4+3i+o
                emit((4 + 3 * i + 0) * SZ2 + q, tmax);
                emit((4 + 3 * i + 1) * SZ2 + q, lmax);
                emit((4 + 3 * i + 2) * SZ2 + q, rmax);
                emit(i * SZ2 + q, bmax);
            }
        }
    }
};

template <int H, int W, typename TO>
__global__ __launch_bounds__(64) void k_features_cpcp(const uint8_t* __restrict__ imgs, int n_img, TO* __restrict__ out, int standardize,
                                                      TO mean, TO sd) {
    using K = Cpcp<H, W>;
    __shared__ __attribute__((aligned(16))) float P0[K::N0];
    __shared__ __attribute__((aligned(16))) float P1[4 * K::N1];
    const int lane = threadIdx.x;
    K::template init<64>(P0, P1, lane);
    __syncthreads();
    for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
        TO* dst = out + (size_t)img * K::F;
        K::template image<64>(P0, P1, imgs + (size_t)img * (H * W), lane, [&](int e, float fv) {
            TO v = (TO)fv;
            if (standardize) {
                const TO dd = (v - mean) / sd;                               // rcn.rs:407-412
                v = dd >= (TO)0 ? dd : (TO)0;
            }
            dst[e] = v;
        });
        __syncthreads();
    }
}

// flatten_feature_set + standardise of a whole epoch segment, written STRAIGHT into the slice-major image the pipelined
// training kernels read (dense_pipe.hpp: Xs[batch][slice][sample][16 features], Ys[batch][sample][classes]) -- the
// end-to-end form: u8 images in, no [N][F] feature matrix in HBM, no separate gather pass.  One wave per (batch, sample);
// the sample's image is perm[batch * B + sample] (or that index itself).  Same arithmetic as k_features_cpcp with the
// fused standardisation, so the packed values are bit-identical to features -> standardise -> k_pack_epoch.
template <int H, int W, typename TO>
__global__ __launch_bounds__(64) void k_features_cpcp_packed(const uint8_t* __restrict__ imgs, const TO* __restrict__ Y, const int* __restrict__ perm,
                                                             int B, int n_batches, int G, int C, TO mean, TO sd, TO* __restrict__ xs,
                                                             TO* __restrict__ ys) {
    using K = Cpcp<H, W>;
    __shared__ __attribute__((aligned(16))) float P0[K::N0];
    __shared__ __attribute__((aligned(16))) float P1[4 * K::N1];
    const int lane = threadIdx.x;
    K::template init<64>(P0, P1, lane);
    __syncthreads();
    const int total = n_batches * B;
    for (int L = blockIdx.x; L < total; L += gridDim.x) {
        const int jb = L / B, smp = L - jb * B;
        const long long img = perm ? perm[L] : L;
        TO* xb = xs + ((size_t)jb * G * B + smp) * 16;                 // + slice * B * 16 + feature % 16
        K::template image<64>(P0, P1, imgs + (size_t)img * (H * W), lane, [&](int e, float fv) {
            const TO dd = ((TO)fv - mean) / sd;                           // rcn.rs:407-412
            xb[(size_t)(e >> 4) * B * 16 + (e & 15)] = dd >= (TO)0 ? dd : (TO)0;
        });
        if (lane < C) ys[(size_t)L * C + lane] = Y[(size_t)img * C + lane];
        __syncthreads();
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
