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
template <typename TC, typename TO>
__global__ __launch_bounds__(kFeatThreads) void k_features(
    FeatDesc fd, const uint8_t* __restrict__ imgs, int n_img, TO* __restrict__ out, int standardize, TO mean, TO sd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* bufA = reinterpret_cast<TC*>(smem_raw);
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
