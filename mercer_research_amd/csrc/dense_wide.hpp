// dense_wide.hpp -- the dense forward / delta pass for layer stacks too wide for k_dense_fwd, which keeps every layer's
// activations of a 16-sample tile in LDS (about 512 hidden units in f32, 128 in f64).  The reference has no width limit
// (RCN::new never fails, rcn.rs:58-75), so beyond it the same step runs layer by layer on activations held in global memory:
// plain FMA loops, one thread per (sample, unit), each dot product accumulated in ascending k exactly as the reference's
// gemv does (rcn.rs:113, 287, 308).  Correctness path, not a tuned one; the weight gradient is unchanged (k_dense_wgrad's
// LDS use does not depend on the layer widths).  Buffers and layouts are k_dense_fwd's: acts / deltas [layer][B][d].
#pragma once

#include "dense.hpp"

namespace rcn {

inline bool dense_is_wide(const NetDesc& nd, size_t esz) { return dense_fwd_lds_elems(nd) * esz > 160 * 1024; }

// a_{j+1}[s][m] = sigmoid(sum_k W_j[m][k] a_j[s][k] + b_j[m])     (rcn.rs:111-114, 285-290)
template <typename T>
__global__ void k_wide_forward(NetDesc nd, const T* __restrict__ params, int j, const T* __restrict__ Ain, long long ldA,
                               const int* __restrict__ idx, int B, T* __restrict__ Aout) {
    const int K = nd.dims[j], M = nd.dims[j + 1];
    const T* W = params + nd.w_off[j];
    const T* b = W + (size_t)K * M;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)B * M; e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e / M), m = (int)(e - (long long)s * M);
        const T* a = Ain + (idx ? (long long)idx[s] : (long long)s) * ldA;
        T z = 0;
        for (int k = 0; k < K; ++k) z += W[(size_t)k * M + m] * a[k];
        Aout[(size_t)s * M + m] = sigmoid_ref(z + b[m]);
    }
}

// output layer: delta_L = (a_L - y) (*) a_L (1 - a_L), and the 16-sample tile's share of sum ||a_L - y||^2   (rcn.rs:299)
template <typename T>
__global__ void k_wide_output_delta(NetDesc nd, const T* __restrict__ aL, const T* __restrict__ Y, const int* __restrict__ idx, int B,
                                    T* __restrict__ dL, T* __restrict__ loss_part) {
    const int M = nd.dims[nd.L], tile = blockIdx.x, s0 = tile * kTileS;
    __shared__ T part[kTileS];
    if (threadIdx.x < kTileS) {
        const int s = s0 + threadIdx.x;
        T t = 0;
        if (s < B) {
            const T* y = Y + (idx ? (long long)idx[s] : (long long)s) * M;
            for (int m = 0; m < M; ++m) {
                const T a = aL[(size_t)s * M + m], diff = a - y[m];
                dL[(size_t)s * M + m] = diff * (a * ((T)1 - a));
                t += diff * diff;
            }
        }
        part[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int i = 0; i < kTileS; ++i) t += part[i];
        loss_part[tile] = t;
    }
}

// delta_j[s][k] = (sum_m W_j[m][k] delta_{j+1}[s][m]) * a_j[s][k] (1 - a_j[s][k])     (rcn.rs:305-309), j >= 1
template <typename T>
__global__ void k_wide_delta(NetDesc nd, const T* __restrict__ params, int j, const T* __restrict__ dNext, const T* __restrict__ aj, int B,
                             T* __restrict__ dj) {
    const int K = nd.dims[j], M = nd.dims[j + 1];
    const T* W = params + nd.w_off[j];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)B * K; e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e / K), k = (int)(e - (long long)s * K);
        const T* w = W + (size_t)k * M;
        const T* d = dNext + (size_t)s * M;
        T t = 0;
        for (int m = 0; m < M; ++m) t += w[m] * d[m];
        const T a = aj[(size_t)s * K + k];
        dj[(size_t)s * K + k] = t * (a * ((T)1 - a));
    }
}

}  // namespace rcn
