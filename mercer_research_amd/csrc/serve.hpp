// serve.hpp -- RCN::classify minus the PNG decode (rcn.rs:82-98) as ONE launch per request on gfx950:
// u8 image -> conv,pool,conv,pool features -> standardise+clamp -> every dense layer -> arg-max, one workgroup per image.
// This is the latency path behind rcn_hip_classify_images (one request of backend/src/main.rs:22-42); throughput work
// goes through the batch kernels.
//
// What shapes the kernel is dependent latency, not bandwidth (a request is 784 B in, 94 KB of W_0, 4 B out):
//  * the first layer's weights do not depend on the image, so every thread issues its (up to 32) W_0 loads BEFORE the
//    feature stage and they land while the stencils run -- one HBM/L2 round trip for the whole forward pass instead of
//    one per stage;
//  * 1024 threads: thread (j, kk) owns output j and the input columns kk, kk+kkn, ... (kkn = 1024 / d_1), so a
//    wave's loads of one step are consecutive 4-byte words of the column-major W_0; partial sums meet in LDS and are
//    added in fixed kk order (deterministic);
//  * all later layers ([b_0 | W_1 | b_1 | ...], a few hundred values) are staged in LDS up front, also under the
//    feature stage;
//  * the class index is written straight into host-mapped memory; the host waits on that word.
#pragma once

#include "common.hpp"
#include "features.hpp"

namespace rcn {

constexpr int kServeThreads = 1024;
constexpr int kServeWReg = 32;        // first-layer weights prefetched per thread
constexpr int kServeTail = 4096;      // values of [b_0 | W_1 | b_1 | ...] staged in LDS
constexpr int kServeMaxDim = 256;     // widest layer after the input

inline bool serve_supported(const NetDesc& nd) {
    if (nd.L < 1 || nd.dims[0] != Cpcp<28, 28>::F) return false;
    for (int l = 1; l <= nd.L; ++l)
        if (nd.dims[l] < 1 || nd.dims[l] > kServeMaxDim) return false;
    const int kkn = kServeThreads / nd.dims[1];
    if ((nd.dims[0] + kkn - 1) / kkn > kServeWReg) return false;
    return nd.P - (nd.w_off[0] + nd.dims[0] * nd.dims[1]) <= kServeTail;
}

template <typename T>
__global__ __launch_bounds__(kServeThreads) void k_serve(NetDesc nd, const T* __restrict__ params, const uint8_t* __restrict__ imgs, T mean, T sd,
                                                         int* __restrict__ cls, T* __restrict__ out_opt) {
    using K = Cpcp<28, 28>;
    __shared__ __attribute__((aligned(16))) float P0[K::N0];
    __shared__ __attribute__((aligned(16))) float P1[4 * K::N1];
    __shared__ T xf[K::F];
    __shared__ T tail[kServeTail];
    __shared__ T part[kServeThreads];
    __shared__ T act[2][kServeMaxDim];
    const int tid = threadIdx.x, img = blockIdx.x;
    const int d0 = nd.dims[0], d1 = nd.dims[1];
    const int kkn = kServeThreads / d1;
    const int kk = tid / d1, j = tid - kk * d1;
    const bool lane_on = kk < kkn;

    // (1) first-layer weights: unconditional loads from clamped addresses, masked by value later
    T wreg[kServeWReg];
    const T* W0 = params + nd.w_off[0];
#pragma unroll
    for (int i = 0; i < kServeWReg; ++i) {
        const int k = kk + kkn * i;
        const int kc = (lane_on && k < d0) ? k : 0;
        wreg[i] = W0[(size_t)kc * d1 + j];
    }
    const int tail_base = nd.w_off[0] + d0 * d1, ntail = nd.P - tail_base;
    for (int e = tid; e < ntail; e += kServeThreads) tail[e] = params[tail_base + e];

    // (2) flatten_feature_set + standardise (rcn.rs:84-89)
    K::template init<kServeThreads>(P0, P1, tid);
    __syncthreads();
    K::template image<kServeThreads>(P0, P1, imgs + (size_t)img * (28 * 28), tid, [&](int ea, int eb, float fa, float fb) {
        xf[ea] = standardise_clamp<false>((T)fa, mean, sd, (T)0);
        xf[eb] = standardise_clamp<false>((T)fb, mean, sd, (T)0);
    });
    __syncthreads();

    // (3) a_1 = sigmoid(W_0 x + b_0)   (rcn.rs:111-114)
    T acc = 0;
#pragma unroll
    for (int i = 0; i < kServeWReg; ++i) {
        const int k = kk + kkn * i;
        const bool on = lane_on && k < d0;
        const T xv = xf[on ? k : 0];
        acc += wreg[i] * (on ? xv : (T)0);
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < d1) {
        T s = 0;
        for (int q = 0; q < kkn; ++q) s += part[q * d1 + tid];
        act[0][tid] = sigmoid_ref(s + tail[tid]);
    }
    __syncthreads();

    // (4) the remaining layers out of LDS
    int cur = 0;
    for (int l = 1; l < nd.L; ++l) {
        const int in = nd.dims[l], on = nd.dims[l + 1];
        const T* W = tail + (nd.w_off[l] - tail_base);
        const T* b = W + in * on;
        if (tid < on) {
            T s = 0;
            for (int k = 0; k < in; ++k) s += W[k * on + tid] * act[cur][k];
            act[cur ^ 1][tid] = sigmoid_ref(s + b[tid]);
        }
        __syncthreads();
        cur ^= 1;
    }

    // (5) rcn.rs:92-97: max_by(total_cmp) keeps the LAST maximal index
    const int C = nd.dims[nd.L];
    if (out_opt && tid < C) out_opt[(size_t)img * C + tid] = act[cur][tid];
    if (tid == 0) {
        int best = 0;
        for (int i = 1; i < C; ++i)
            if (!(act[cur][i] < act[cur][best])) best = i;
        cls[img] = best;
        __threadfence_system();
    }
}

}  // namespace rcn
