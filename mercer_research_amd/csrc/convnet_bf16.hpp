// convnet_bf16.hpp -- Track X, the CDNA4 bf16 MFMA path (BASELINE.json configs[4] asks for one).
//
// Mixed precision in the usual sense: activations, gradients and parameters stay fp32 in HBM (master copies; the SGD update
// and every reduction are fp32), the two GEMM operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS,
// and v_mfma_f32_32x32x16_bf16 accumulates in fp32.  That is 8x the MFMA rate of the fp32 path (32x32x2) at the same
// HBM traffic, with ~2^-9 relative rounding per operand -- results agree with the f64 oracle to ~1e-2, which the tests
// state, not to the fp32 path's 1e-4.
//
// Operand layout (cdna_hip_programming.md, "A/B operand lane maps, bf16"): lane l (r = l & 31, h = l >> 5) holds
// A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7 -- eight CONSECUTIVE k per lane for both operands.  So both LDS
// tiles are stored k-contiguous and every fragment is one ds_read_b128:
//   A tile [128 rows][32 k]   rows of the implicit im2col matrix; thread loads 4 fp32 of a row (16 B), stores 4 bf16 (8 B)
//   B tile [BN cols][32 k]    from a bf16 copy of the weights stored TRANSPOSED, [Cout][K] (k_prep_weights_bf16, once per
//                             step and layer -- weights are small), so a tile row is a plain 64-byte run of HBM
// Row stride 40 halves (80 B): 16-byte aligned, and consecutive rows start 20 banks apart.
#pragma once

#include "convnet.hpp"

namespace rcnx {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int kLdH = kBK + 8;      // halves per LDS tile row

// dst[c][r] = bf16(src[r][c]) for r < R, 0 for R <= r < Rp   (src: [R][C] fp32 row-major, dst: [C][Rp] bf16)
__global__ void k_prep_weights_bf16(const float* __restrict__ src, int R, int C, __bf16* __restrict__ dst, int Rp) {
    const long long total = (long long)C * Rp;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e / Rp), r = (int)(e - (long long)c * Rp);
        dst[e] = r < R ? (__bf16)src[(long long)r * C + c] : (__bf16)0.f;
    }
}

__device__ inline bf16x4 to_bf16x4(const f32x4& v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
}

// Same tiling, split-K and epilogues as k_conv_fwd; WB = weights as bf16 [Cout][Kp], Kp = K rounded up to 32
template <int KS, bool SMALLC, int BN, int EPI>
__global__ __launch_bounds__(kThreads) void k_conv_fwd_bf16(const float* __restrict__ X, const __bf16* __restrict__ WB,
                                                            const float* __restrict__ bias, float* __restrict__ Y, ConvShape s) {
    constexpr int NT = BN / 32;
    constexpr int BCH = (BN * 4 + kThreads - 1) / kThreads;          // 16-byte chunks of the B tile per thread
    __shared__ __attribute__((aligned(16))) __bf16 As[2][kBM * kLdH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN * kLdH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long M = (long long)s.N * s.H * s.W;
    const long long m0 = (long long)blockIdx.x * kBM;
    const int n0 = blockIdx.y * BN;
    const int K = KS * KS * s.Cin;
    const int Kp = (K + kBK - 1) / kBK * kBK;
    const int nkt_all = Kp / kBK;
    const int kt0 = (int)((long long)nkt_all * blockIdx.z / gridDim.z), kt1 = (int)((long long)nkt_all * (blockIdx.z + 1) / gridDim.z);
    const int nkt = kt1 - kt0;
    Y += (long long)blockIdx.z * M * s.Cout;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x4 av[4];
    bf16x8 bv[BCH];
    auto load_b = [&](int kt) {
#pragma unroll
        for (int q = 0; q < BCH; ++q) {
            const int e = tid + kThreads * q;
            const int ec = e < BN * 4 ? e : 0;                        // unconditional load from a clamped address
            const int col = ec >> 2, kq = ec & 3;
            bv[q] = *reinterpret_cast<const bf16x8*>(WB + (long long)(n0 + col) * Kp + kt * kBK + kq * 8);
        }
    };
    auto store_tiles = [&](int buf) {
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<bf16x4*>(&As[buf][((tid >> 3) + 32 * q) * kLdH + c4]) = to_bf16x4(av[q]);
#pragma unroll
        for (int q = 0; q < BCH; ++q) {
            const int e = tid + kThreads * q;
            if (e < BN * 4) *reinterpret_cast<bf16x8*>(&Bs[buf][(e >> 2) * kLdH + (e & 3) * 8]) = bv[q];
        }
    };

    const ARows rows = decode_rows(s, M, m0, tid);
    load_a_regs<KS, SMALLC>(X, s, rows, kt0, tid, av);
    load_b(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            load_a_regs<KS, SMALLC>(X, s, rows, kt0 + kt + 1, tid, av);
            load_b(kt0 + kt + 1);
        }
        const __bf16* a = &As[cur][(wave * 32 + (lane & 31)) * kLdH + 8 * (lane >> 5)];
        const __bf16* b = &Bs[cur][(lane & 31) * kLdH + 8 * (lane >> 5)];
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + 16 * ks);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b + 32 * t * kLdH + 16 * ks);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) {
            store_tiles(cur ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = n0 + 32 * t + (lane & 31);
        const float bb = (EPI >= 1) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + mfma32_row(lane, r);
            if (m < M) {
                float v = acc[t][r] + bb;
                if (EPI == 2) v = v > 0.f ? v : 0.f;
                Y[m * s.Cout + co] = v;
            }
        }
    }
}

}  // namespace rcnx
