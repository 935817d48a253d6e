// ops.hpp -- device versions of the public operator traits Convolve2D / Pool2D (utils/kernel.rs:61-100,
// 219-236) for arbitrary f64 matrices.  One thread per output element, f64 arithmetic in the reference's
// summation order (ky outer, kx inner, accumulate from zero; the library is built with -ffp-contract=off),
// column-major matrices exactly as nalgebra stores them.
#pragma once

#include "common.hpp"

namespace rcn {

// Convolve2D::convolve_2d, kernel.rs:110-194.  n matrices R x C back to back; one kernel kr x kc.
__global__ void k_convolve_2d_f64(const double* __restrict__ m, int n, int R, int C, const double* __restrict__ k,
                                  int kr, int kc, int same, int oR, int oC, double* __restrict__ out) {
    const size_t total = (size_t)n * oR * oC;
    const int pr = kr / 2, pc = kc / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(e / ((size_t)oR * oC));
        const int p = (int)(e - (size_t)img * oR * oC);
        const int cx = p / oR, cy = p - cx * oR;                   // column-major output index
        const double* M = m + (size_t)img * R * C;
        double conv = 0.0;
        for (int ky = 0; ky < kr; ++ky)
            for (int kx = 0; kx < kc; ++kx) {
                double v;
                if (same) {
                    // padded[(a,b)] = self[(a-1,b-1)] for a in 1..R+pr, b in 1..C+pc, else 0   kernel.rs:154-158
                    const int a = cy + ky, b = cx + kx;
                    v = (a >= 1 && a < R + pr && b >= 1 && b < C + pc) ? M[(size_t)(b - 1) * R + (a - 1)] : 0.0;
                } else {
                    v = M[(size_t)(cx + kx) * R + (cy + ky)];      // kernel.rs:186
                }
                conv += v * k[(size_t)kx * kr + ky];
            }
        out[e] = conv;
    }
}

// Convolve2D::relu, kernel.rs:209-216
__global__ void k_relu_f64(const double* __restrict__ m, size_t n, double* __restrict__ out) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        out[e] = m[e] >= 0.0 ? m[e] : 0.0;
}

// Pool2D::pool_2d (Max), kernel.rs:245-349.  `same`: zero-pad odd dims bottom/right; else truncate.
__global__ void k_pool_2d_f64(const double* __restrict__ m, int n, int R, int C, int oR, int oC, double* __restrict__ out) {
    const size_t total = (size_t)n * oR * oC;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(e / ((size_t)oR * oC));
        const int p = (int)(e - (size_t)img * oR * oC);
        const int rx = p / oR, ry = p - rx * oR;
        const double* M = m + (size_t)img * R * C;
        double best = 0.0;
        bool first = true;
        for (int px = 0; px < 2; ++px)
            for (int py = 0; py < 2; ++py) {
                const int y = ry * 2 + px, x = rx * 2 + py;
                const double v = (y < R && x < C) ? M[(size_t)x * R + y] : 0.0;   // padded cells are zero (kernel.rs:310-319)
                if (first || !(v < best)) best = v;                               // max_by(partial_cmp): last maximum
                first = false;
            }
        out[e] = best;
    }
}

}  // namespace rcn
