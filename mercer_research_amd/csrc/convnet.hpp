// convnet.hpp -- "Track X": the trainable convolution network BASELINE.json's north_star asks for and the reference
// does NOT have (its conv layers are four fixed Sobel filters without a backward pass).  No reference
// counterpart => parity is against this repository's own f64 oracle (oracle/convnet_oracle.py) + finite differences.
//
// Layout: activations NHWC fp32 ([N][H][W][C], channels contiguous), so a convolution is an implicit GEMM
//     Y[m][co] = sum_k A[m][k] * Wk[k][co],   m = (n*H + oh)*W + ow,   k = (kh*KS + kw)*Cin + ci
// whose A rows are, per filter tap, CONTIGUOUS runs of Cin floats of the input -- the im2col matrix is never written to
// HBM; the A-tile loader gathers it straight into LDS (zero rows for the padding halo).  Weights are stored K-major
// (Wk[K][Cout]), which is also the dense layer's [in][out] matrix: a dense layer is the KS = 1 case on a 1x1 image.
//
//   k_conv_fwd     128 x BN output tile per workgroup, BK = 32, fp32 MFMA 32x32x2 (exact f32 FMA chains),
//                  register-staged double buffering, bias + ReLU epilogue.
//   dgrad          the same kernel on dZ with the tap-flipped, transposed weights Wt[(kh',kw',co)][ci] (k_flip_weights).
//   k_conv_wgrad   dW[k][co] = sum_m A[m][k] dZ[m][co]: contraction over the output pixels, split over workgroups in
//                  chunks whose partial tiles go to a slab; k_reduce_all sums every layer's slab in chunk order (bit-
//                  reproducible) and applies the SGD step.
//   k_pool_fwd/bwd 2x2/2 max-pool with a 2-bit arg-max image; backward also applies the ReLU mask.
//   k_softmax_ce   fused softmax + cross-entropy forward and (p - onehot)/B backward.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rcnx {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Storage type of an activation / gradient tensor of the convolutional stage: float, or __bf16 (rcn_hipx_set_precision
// RCN_HIPX_BF16_STORED: every consumer of these tensors rounds them to bf16 on the way into LDS anyway, so storing them rounded halves
// the stage's HBM traffic and changes the arithmetic only where the first layer's fp32 weight-gradient kernel reads dZ).
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
template <typename TS> struct Chunk4;
template <> struct Chunk4<float> { using type = f32x4; };
template <> struct Chunk4<__bf16> { using type = bf16x4; };
template <typename TS> using chunk4_t = typename Chunk4<TS>::type;      // four consecutive channels as they lie in memory
__device__ inline f32x4 widen4(const f32x4& v) { return v; }
__device__ inline f32x4 widen4(const bf16x4& v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
__device__ inline float widen(float v) { return v; }
__device__ inline float widen(__bf16 v) { return (float)v; }
template <typename TS> __device__ inline TS narrow(float v) { return (TS)v; }                      // (float -> __bf16: round to nearest even)
template <typename TS> __device__ inline chunk4_t<TS> narrow4(const f32x4& v);
template <> __device__ inline f32x4 narrow4<float>(const f32x4& v) { return v; }
template <> __device__ inline bf16x4 narrow4<__bf16>(const f32x4& v) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }


constexpr int kBM = 128, kBK = 32, kThreads = 256;
constexpr int kLdA = kBK + 1;

// D layout of v_mfma_f32_32x32x2_f32: lane l holds col = l & 31, rows (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r in [0,16)
__device__ inline int mfma32_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

struct ConvShape {
    int N, H, W, Cin, Cout;     // stride 1, pad = KS/2, output H x W
};

// Per-thread description of the 4 rows of the 128 x 32 A tile it stages: decoded ONCE per workgroup (the pixel of a row does
// not change along K), so that per K-tile only the tap offset -- uniform over the workgroup -- is applied.
struct ARows {
    long long base[4];      // element offset of the row's own pixel, channel 0 (or -1: row past M)
    int oh[4], ow[4];
};

__device__ inline ARows decode_rows(const ConvShape& s, long long M, long long m0, int tid) {
    ARows r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long m = m0 + (tid >> 3) + 32 * q;
        if (m < M) {
            const int ow = (int)(m % s.W);
            const long long t2 = m / s.W;
            const int oh = (int)(t2 % s.H);
            r.base[q] = m * (long long)s.Cin;          // ((n*H + oh)*W + ow) * Cin == m * Cin
            r.oh[q] = oh; r.ow[q] = ow;
        } else { r.base[q] = -1; r.oh[q] = 0; r.ow[q] = 0; }
    }
    return r;
}

// A-tile element source: row m of the implicit im2col matrix, K-tile kt (32 consecutive k)
template <int KS, bool SMALLC, typename TX = float>
__device__ inline void load_a_regs(const TX* __restrict__ X, const ConvShape& s, const ARows& rows, int kt, int tid, f32x4 (&v)[4]) {
    // thread t loads rows (t>>3) + 32 q, q = 0..3, columns 4*(t&7) .. +3 of the 128 x 32 tile
    const int c4 = (tid & 7) * 4;
    if (!SMALLC) {
        const int k0 = kt * kBK;                                    // Cin % 32 == 0: the whole K-tile lies in one tap
        const int tap = k0 / s.Cin, ci = k0 - tap * s.Cin + c4;
        const int dh = tap / KS - KS / 2, dw = tap % KS - KS / 2;
        const long long toff = ((long long)dh * s.W + dw) * s.Cin + ci;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = rows.base[q] >= 0 && (unsigned)(rows.oh[q] + dh) < (unsigned)s.H && (unsigned)(rows.ow[q] + dw) < (unsigned)s.W;
            const f32x4 val = widen4(*reinterpret_cast<const chunk4_t<TX>*>(X + (ok ? rows.base[q] + toff : 0)));     // unconditional load, masked by value
            v[q] = ok ? val : f32x4{0, 0, 0, 0};
        }
    } else {
        // whole K (= KS*KS*Cin <= 32) in one tile.  The tap of column k does not depend on the row: decoded ONCE per thread (four
        // columns), with the division by a compile-time 3 for RGB input -- as a per-element k / s.Cin this loader issued ~100 VALU
        // instructions per MFMA of the first layer (PMC, CIFAR shape).
        int dh[4], dw[4], off[4];
        bool kv[4];
        auto decode = [&](auto cin_c) {
            constexpr int CC = decltype(cin_c)::value;
            const int Cin = CC ? CC : s.Cin;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = c4 + i;
                const int tap = k / Cin, ci = k - tap * Cin;
                kv[i] = k < KS * KS * Cin;
                dh[i] = tap / KS - KS / 2;
                dw[i] = tap % KS - KS / 2;
                off[i] = (dh[i] * s.W + dw[i]) * Cin + ci;
            }
        };
        if (s.Cin == 3) decode(std::integral_constant<int, 3>{});
        else if (s.Cin == 1) decode(std::integral_constant<int, 1>{});
        else decode(std::integral_constant<int, 0>{});
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 val = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = rows.base[q] >= 0 && kv[i] && (unsigned)(rows.oh[q] + dh[i]) < (unsigned)s.H && (unsigned)(rows.ow[q] + dw[i]) < (unsigned)s.W;
                const float x = widen(X[ok ? rows.base[q] + off[i] : 0]);           // unconditional load, masked by value
                val[i] = ok ? x : 0.f;
            }
            v[q] = val;
        }
    }
}

// EPI: 0 = raw, 1 = + bias, 2 = + bias, ReLU, 3 = raw gated by (G > 0) where `bias` points at a tensor G shaped like Y (the
// input-gradient pass writes dZ of the layer below directly: G = that layer's post-ReLU output)
template <int KS, bool SMALLC, int BN, int EPI>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3))) void k_conv_fwd(const float* __restrict__ X, const float* __restrict__ Wk,
                                                       const float* __restrict__ bias, float* __restrict__ Y, ConvShape s) {
    constexpr int kLdB = BN + 1, NT = BN / 32;
    __shared__ float As[2][kBM * kLdA];
    __shared__ float Bs[2][kBK * kLdB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long M = (long long)s.N * s.H * s.W;
    const long long m0 = (long long)blockIdx.x * kBM;
    const int n0 = blockIdx.y * BN;
    const int K = KS * KS * s.Cin;
    const int nkt_all = SMALLC ? 1 : K / kBK;
    // split-K: workgroup z of gridDim.z contracts K-tiles [kt0, kt1) and writes a raw partial tile to Y + z * M * Cout
    const int kt0 = (int)((long long)nkt_all * blockIdx.z / gridDim.z), kt1 = (int)((long long)nkt_all * (blockIdx.z + 1) / gridDim.z);
    const int nkt = kt1 - kt0;
    Y += (long long)blockIdx.z * M * s.Cout;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Two register stages: the global loads of K-tile kt + 2 are issued before tile kt is contracted and waited for only when tile
    // kt + 1 has been contracted too -- two tiles of MFMA work (4096 cycles per wave at BN = 64) under every load instead of one.
    // (PMC, CIFAR shape, one stage: SQ_WAIT_INST_ANY 48 % of the wave cycles, SQ_WAIT_INST_LDS 2 %: the waves sat on vmcnt.)
    f32x4 av[2][4];
    f32x4 bv[2][BN / 32];                                // B tile: 32 x BN floats = 8*BN float4 / 256 threads
    auto load_b = [&](int kt, f32x4 (&b)[BN / 32]) {
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            const int e = tid + kThreads * q;            // float4 index in the 32 x (BN/4) tile
            const int kr = e / (BN / 4), c4 = (e - kr * (BN / 4)) * 4;
            const int k = kt * kBK + kr;
            b[q] = (k < K) ? *reinterpret_cast<const f32x4*>(Wk + (long long)k * s.Cout + n0 + c4) : f32x4{0, 0, 0, 0};
        }
    };
    auto store_tiles = [&](int buf, const f32x4 (&a)[4], const f32x4 (&b)[BN / 32]) {
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* d = &As[buf][((tid >> 3) + 32 * q) * kLdA + c4];
            d[0] = a[q][0]; d[1] = a[q][1]; d[2] = a[q][2]; d[3] = a[q][3];
        }
#pragma unroll
        for (int q = 0; q < BN / 32; ++q) {
            const int e = tid + kThreads * q;
            const int kr = e / (BN / 4), c4b = (e - kr * (BN / 4)) * 4;
            float* d = &Bs[buf][kr * kLdB + c4b];
            d[0] = b[q][0]; d[1] = b[q][1]; d[2] = b[q][2]; d[3] = b[q][3];
        }
    };
    auto contract = [&](int cur) {
        const float* a = &As[cur][(wave * 32 + (lane & 31)) * kLdA + (lane >> 5)];
        const float* b = &Bs[cur][(lane >> 5) * kLdB + (lane & 31)];
#pragma unroll
        for (int ks = 0; ks < kBK / 2; ++ks) {
            const float af = a[2 * ks];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b[2 * ks * kLdB + 32 * t], acc[t], 0, 0, 0);
        }
    };

    const ARows rows = decode_rows(s, M, m0, tid);
    load_a_regs<KS, SMALLC>(X, s, rows, kt0, tid, av[0]);
    load_b(kt0, bv[0]);
    if (nkt > 1) {
        load_a_regs<KS, SMALLC>(X, s, rows, kt0 + 1, tid, av[1]);
        load_b(kt0 + 1, bv[1]);
    }
    store_tiles(0, av[0], bv[0]);
    __syncthreads();
    // tile kt sits in LDS buffer kt & 1 and came through register stage kt & 1: that stage is free for tile kt + 2
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 2 < nkt) {
            load_a_regs<KS, SMALLC>(X, s, rows, kt0 + kt + 2, tid, av[0]);
            load_b(kt0 + kt + 2, bv[0]);
        }
        contract(0);
        if (kt + 1 < nkt) {
            store_tiles(1, av[1], bv[1]);
            __syncthreads();
            if (kt + 3 < nkt) {
                load_a_regs<KS, SMALLC>(X, s, rows, kt0 + kt + 3, tid, av[1]);
                load_b(kt0 + kt + 3, bv[1]);
            }
            contract(1);
            if (kt + 2 < nkt) {
                store_tiles(0, av[0], bv[0]);
                __syncthreads();
            }
        }
    }
    // epilogue: lane holds column co = n0 + 32 t + (lane & 31), 16 rows
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = n0 + 32 * t + (lane & 31);
        const float bb = (EPI == 1 || EPI == 2) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + mfma32_row(lane, r);
            if (m < M) {
                float v = acc[t][r] + bb;
                if (EPI == 2) v = v > 0.f ? v : 0.f;
                if (EPI == 3) v = bias[m * s.Cout + co] > 0.f ? v : 0.f;      // dgrad: ReLU mask of the layer below, `bias` = its output
                Y[m * s.Cout + co] = v;
            }
        }
    }
}

// Wt[(kh', kw', co)][ci] = Wk[((2-kh')*KS + (2-kw'))*Cin + ci][co]  (tap-flipped transpose for dgrad); KS = 1: plain transpose
__global__ void k_flip_weights(const float* __restrict__ Wk, float* __restrict__ Wt, int KS, int Cin, int Cout) {
    const long long total = (long long)KS * KS * Cin * Cout;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % Cin);
        const long long t = e / Cin;
        const int co = (int)(t % Cout);
        const int tap = (int)(t / Cout);
        const int kh = tap / KS, kw = tap - kh * KS;
        const int ftap = (KS - 1 - kh) * KS + (KS - 1 - kw);
        Wt[e] = Wk[((long long)ftap * Cin + ci) * Cout + co];
    }
}

// XCD-aware workgroup -> (k-tile, n-tile, pixel chunk) map for the weight-gradient kernels.  All tiles of one pixel chunk
// read the same rows of X (nine shifted views, one per filter tap) and of dZ; hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs, each with its own L2, so a plain 3-D grid scatters a chunk's tiles over all eight L2s and
// every one of them fetches the chunk again.  Here workgroup id L runs on XCD L % 8 and works on chunk 8 (L / (8 T)) + L % 8,
// tile (L / 8) % T: the T tiles of a chunk are consecutive ON ONE XCD and share its L2.
struct WgradGrid {
    int tiles_x, tiles_y, nchunks, xcd_aware;
    __host__ __device__ int tiles() const { return tiles_x * tiles_y; }
    __host__ unsigned launch_blocks() const { return (unsigned)(tiles() * (xcd_aware ? (nchunks + 7) / 8 * 8 : nchunks)); }
    __device__ bool decode(int L, int& tx, int& ty, int& chunk) const {
        const int T = tiles();
        int inner;
        if (xcd_aware) { const int j = L >> 3; inner = j % T; chunk = (j / T) * 8 + (L & 7); }
        else { inner = L % T; chunk = L / T; }
        tx = inner % tiles_x; ty = inner / tiles_x;
        return chunk < nchunks;
    }
};

// dW partial tiles: workgroup (kb, nb, chunk) computes rows [32 kb, +32) x cols [BN nb, +BN) of dW over the pixels of
// its chunk and writes them to slab[chunk][K + 1][Cout]; row K is the chunk's partial bias gradient (column sums of dZ,
// taken by the kb == 0 workgroups from the dZ tiles they stage anyway).  [W | b] is contiguous in the parameter buffer,
// so ONE k_reduce_all job over (K + 1) * Cout elements finishes both.
template <int KS, bool SMALLC, int BN>
__global__ __launch_bounds__(kThreads) void k_conv_wgrad(const float* __restrict__ X, const float* __restrict__ dZ,
                                                         float* __restrict__ slab, ConvShape s, int pix_per_chunk, WgradGrid gd) {
    // Both MFMA operands are read along a staged row (A[m = k][kk = pixel] = Xs[pixel][k], B[kk = pixel][n] = Ds[pixel][n]:
    // a half-wave reads 32 consecutive floats of one row), so the LDS images need no padding and are filled with 16-byte
    // stores.  128 pixels per iteration in ONE LDS buffer (48 KB at BN = 64 -> three workgroups per CU); the next 128 are
    // prefetched into registers while the current ones are contracted (32 MFMAs per wave between barriers).
    constexpr int NT = BN / 32, kPT = 128;
    __shared__ __attribute__((aligned(16))) float smem[kPT * 32 + kPT * BN];
    float* Xs = smem;                                    // [pixel][k]
    float* Ds = smem + kPT * 32;                         // [pixel][co]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long M = (long long)s.N * s.H * s.W;
    const int K = KS * KS * s.Cin;
    int kb, nby, chunk;
    if (!gd.decode((int)blockIdx.x, kb, nby, chunk)) return;             // padding of the XCD-aware grid (uniform per workgroup)
    const int n0 = nby * BN;
    const long long p0 = (long long)chunk * pix_per_chunk;
    const long long p1 = p0 + pix_per_chunk < M ? p0 + pix_per_chunk : M;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float colsum = 0.f;                                  // bias partial: thread c < BN of the kb == 0 workgroups

    // tap of this k-block: uniform over the workgroup
    const int c4 = (tid & 7) * 4;
    int dh = 0, dw = 0; long long toff = 0;
    if (!SMALLC) {
        const int k0 = kb * 32, tap = k0 / s.Cin;
        dh = tap / KS - KS / 2; dw = tap % KS - KS / 2;
        toff = ((long long)dh * s.W + dw) * s.Cin + (k0 - tap * s.Cin) + c4;
    }
    // SMALLC: the taps of this thread's four columns (row-independent), decoded once; division by a compile-time 3 for RGB input
    int sdh[4] = {0, 0, 0, 0}, sdw[4] = {0, 0, 0, 0}, soff[4] = {0, 0, 0, 0};
    bool skv[4] = {false, false, false, false};
    if (SMALLC) {
        auto decode = [&](auto cin_c) {
            constexpr int CC = decltype(cin_c)::value;
            const int Cin = CC ? CC : s.Cin;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = c4 + i;
                const int tap = k / Cin, ci = k - tap * Cin;
                skv[i] = k < KS * KS * Cin;
                sdh[i] = tap / KS - KS / 2;
                sdw[i] = tap % KS - KS / 2;
                soff[i] = (sdh[i] * s.W + sdw[i]) * Cin + ci;
            }
        };
        if (s.Cin == 3) decode(std::integral_constant<int, 3>{});
        else if (s.Cin == 1) decode(std::integral_constant<int, 1>{});
        else decode(std::integral_constant<int, 0>{});
    }
    f32x4 xv[4], dv[BN / 8];                             // 128 x 32 floats / 256 thr = 4 float4; 128 x BN / 256 = BN/8 float4
    // The thread's X rows are pixels m_first, m_first + 32, ... across q AND across stages (128 = 4 * 32): (oh, ow) of the next
    // row is running state advanced by 32 pixels per load -- one 32-bit division per kernel instead of two 64-bit ones per load.
    int row_m = (int)p0 + (tid >> 3);
    int row_ow = row_m % s.W, row_oh = (row_m / s.W) % s.H;
    const int adv_h = 32 / s.W, adv_w = 32 % s.W;
    auto gload = [&](long long pb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = row_m < (int)p1;
            const long long mc = in ? row_m : p0;
            const int ow = row_ow, oh = row_oh;
            row_m += 32; row_ow += adv_w; row_oh += adv_h;
            if (row_ow >= s.W) { row_ow -= s.W; ++row_oh; }
            if (row_oh >= s.H) row_oh %= s.H;
            if (!SMALLC) {
                const bool ok = in && (unsigned)(oh + dh) < (unsigned)s.H && (unsigned)(ow + dw) < (unsigned)s.W;
                const f32x4 val = *reinterpret_cast<const f32x4*>(X + (ok ? mc * (long long)s.Cin + toff : 0));
                xv[q] = ok ? val : f32x4{0, 0, 0, 0};
            } else {
                f32x4 val = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = in && skv[i] && (unsigned)(oh + sdh[i]) < (unsigned)s.H && (unsigned)(ow + sdw[i]) < (unsigned)s.W;
                    const float x = X[ok ? mc * (long long)s.Cin + soff[i] : 0];
                    val[i] = ok ? x : 0.f;
                }
                xv[q] = val;
            }
        }
#pragma unroll
        for (int q = 0; q < BN / 8; ++q) {
            const int e = tid + kThreads * q;
            const int pr = e / (BN / 4), c4b = (e - pr * (BN / 4)) * 4;
            const long long m = pb + pr;
            const f32x4 val = *reinterpret_cast<const f32x4*>(dZ + (m < p1 ? m : p0) * s.Cout + n0 + c4b);
            dv[q] = (m < p1) ? val : f32x4{0, 0, 0, 0};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&Xs[((tid >> 3) + 32 * q) * 32 + c4]) = xv[q];
#pragma unroll
        for (int q = 0; q < BN / 8; ++q) *reinterpret_cast<f32x4*>(&Ds[(tid + kThreads * q) * 4]) = dv[q];
    };

    gload(p0);
    for (long long pb = p0; pb < p1; pb += kPT) {
        lstore();
        __syncthreads();
        if (pb + kPT < p1) gload(pb + kPT);              // next 128 pixels fly while these are contracted
        if (kb == 0) {                                   // bias partial: thread (column tid % BN, pixel part tid / BN)
            constexpr int kParts = kThreads / BN;
            const int col = tid % BN, part = tid / BN;
#pragma unroll 8
            for (int px = part * (kPT / kParts); px < (part + 1) * (kPT / kParts); ++px) colsum += Ds[px * BN + col];
        }
        // wave w contracts pixels 32w .. 32w+31 of the staged 128
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int px = wave * 32 + 2 * ks + (lane >> 5);
            const float af = Xs[px * 32 + (lane & 31)];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, Ds[px * BN + 32 * t + (lane & 31)], acc[t], 0, 0, 0);
        }
        __syncthreads();                                 // everyone is done reading before the next store
    }
    // combine the four waves' partial tiles in wave order (through the now idle staging memory), write the chunk's tile
    constexpr int kLdR = BN + 1;
    static_assert(4 * 32 * kLdR <= kPT * 32 + kPT * BN, "partial tiles must fit the staging memory");
    float* Red = smem;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Red[(wave * 32 + mfma32_row(lane, r)) * kLdR + 32 * t + (lane & 31)] = acc[t][r];
    __syncthreads();
    float* out = slab + (long long)chunk * (K + 1) * s.Cout;
    for (int e = tid; e < 32 * BN; e += kThreads) {
        const int kr = e / BN, c = e - kr * BN;
        const int k = kb * 32 + kr;
        if (k < K) {
            const float v = (Red[(0 * 32 + kr) * kLdR + c] + Red[(1 * 32 + kr) * kLdR + c]) + (Red[(2 * 32 + kr) * kLdR + c] + Red[(3 * 32 + kr) * kLdR + c]);
            out[(long long)k * s.Cout + n0 + c] = v;
        }
    }
    if (kb == 0) {
        __syncthreads();
        smem[tid] = colsum;
        __syncthreads();
        if (tid < BN) {
            float t = 0.f;
            for (int part = 0; part < kThreads / BN; ++part) t += smem[part * BN + tid];
            out[(long long)K * s.Cout + n0 + tid] = t;
        }
    }
}

// split-K epilogue: Y[m][co] = act(bias[co] + sum_z part[z][m][co]), z in order
// TY: storage type of Y and of epi 3's gate tensor (which arrives through `bias`)
template <typename TY = float>
__global__ void k_splitk_epilogue(const float* __restrict__ part, const float* __restrict__ bias, TY* __restrict__ Y, long long MN, int Cout, int Z, int epi) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < MN; e += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < Z; ++z) v += part[(long long)z * MN + e];
        if (epi == 1 || epi == 2) v += bias[(int)(e % Cout)];
        if (epi == 2) v = v > 0.f ? v : 0.f;
        if (epi == 3) v = widen(reinterpret_cast<const TY*>(bias)[e]) > 0.f ? v : 0.f;
        Y[e] = narrow<TY>(v);
    }
}

// The input-gradient pass wants the weights tap-flipped and transposed (k_flip_weights).  Instead of re-making that copy every
// step, the kernels that UPDATE a weight also store it at its flipped position: the copy is always current (the host refreshes it
// with k_flip_weights after every other kind of parameter write).
struct FlipSpec {
    float* wt;                // nullptr: no copy (first layer: no input gradient)
    int taps, Cin, Cout;      // element i < taps * Cin * Cout of [W | b] is W[(tap, ci)][co]; the rest is the bias row
};
__device__ inline void store_flipped(const FlipSpec& f, long long i, float v) {
    if (!f.wt || i >= (long long)f.taps * f.Cin * f.Cout) return;
    const int co = (int)(i % f.Cout), k = (int)(i / f.Cout);
    const int tap = k / f.Cin, ci = k - tap * f.Cin;
    f.wt[((long long)(f.taps - 1 - tap) * f.Cout + co) * f.Cin + ci] = v;
}

// All layers' slabs in ONE launch at the end of the backward pass (a launch per layer -- two above 64 chunks -- was 5-9 us each of
// mostly latency, seven of them per CIFAR step).  Nothing updates a weight before every input-gradient kernel has read it, so the
// jobs are independent.  Job q owns workgroups [first_block, next job's first_block); a workgroup of 1024 threads sums 1024 / g elements
// in g chunk groups (g = the power of two at or above the chunk count, at most 32); groups are combined as a fixed tree: bit-reproducible.
constexpr int kMaxReduceJobs = 16;
struct ReduceJob {
    float* p;                 // [W | b] of the layer
    float* grad;              // or nullptr
    const float* slab;        // [chunks][n]
    FlipSpec flip;
    long long n;
    int chunks, first_block;
};
struct ReduceJobs { ReduceJob j[kMaxReduceJobs]; int njobs; float lr; int apply; };

constexpr int kReduceThreads = 1024;
// chunk groups of a job: the power of two at or above its chunk count, at most 32; a workgroup then covers 4 * 1024 / groups elements
// (a thread owns FOUR consecutive elements, 16-byte loads: with one element per thread a workgroup of the 64-chunk CIFAR layer had 8 KB
// in flight per residency and the kernel ran at 1.8 TB/s of slab, latency times rounds)
__host__ __device__ inline int reduce_job_groups(int chunks) { int g = 1; while (g < chunks && g < 32) g <<= 1; return g; }
__host__ __device__ inline int reduce_job_elems(int chunks) { return 4 * kReduceThreads / reduce_job_groups(chunks); }

__global__ __launch_bounds__(kReduceThreads) void k_reduce_all(ReduceJobs J) {
    __shared__ f32x4 red[kReduceThreads];
    int q = 0;
    while (q + 1 < J.njobs && (int)blockIdx.x >= J.j[q + 1].first_block) ++q;
    const ReduceJob jb = J.j[q];
    const int lb = (int)blockIdx.x - jb.first_block;
    // whole 128-byte rows of the slab per chunk group and wave instruction: 32 threads x 16 bytes x 32 groups (many chunks) ... 1024 threads x 1 group (one chunk)
    const int GR = reduce_job_groups(jb.chunks), EL = kReduceThreads / GR;
    const int el = threadIdx.x % EL, grp = threadIdx.x / EL;
    const long long i = ((long long)lb * EL + el) * 4;               // jb.n % 4 == 0 (host)
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (i < jb.n) {
        // eight loads in flight per thread (a load per loop trip, each waited for, left this kernel at 1.6 TB/s)
        for (int c = grp; c < jb.chunks; c += 8 * GR) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = c + k * GR < jb.chunks ? *reinterpret_cast<const f32x4*>(jb.slab + (long long)(c + k * GR) * jb.n + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            g += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    red[grp * EL + el] = g;
    // the chunk groups combined as a fixed binary tree by all threads (instead of one thread per element walking up to 32 partial sums
    // in LDS; measured: no difference)
    for (int st = GR >> 1; st >= 1; st >>= 1) {
        __syncthreads();
        if (grp < st) red[grp * EL + el] += red[(grp + st) * EL + el];
    }
    __syncthreads();
    if ((int)threadIdx.x < EL && i < jb.n) {
        const f32x4 t = red[threadIdx.x];
        if (jb.grad) *reinterpret_cast<f32x4*>(jb.grad + i) = t;
        if (J.apply) {
            f32x4 v = *reinterpret_cast<const f32x4*>(jb.p + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] - J.lr * t[k];
            *reinterpret_cast<f32x4*>(jb.p + i) = v;
#pragma unroll
            for (int k = 0; k < 4; ++k) store_flipped(jb.flip, i + k, v[k]);
        }
    }
}

// db[co] = sum_m dZ[m][co]: one workgroup per 32-column block, rows strided over threads, fixed-order tree
__global__ __launch_bounds__(256) void k_bias_grad(const float* __restrict__ dZ, long long M, int Cout, float* __restrict__ b, float* __restrict__ grad_out,
                                                   float lr, int apply) {
    __shared__ float red[8][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rgrp = threadIdx.x >> 5;
    float acc = 0.f;
    if (c < Cout)
        for (long long m = rgrp; m < M; m += 8) acc += dZ[m * Cout + c];
    red[rgrp][threadIdx.x & 31] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && c < Cout) {
        float g = 0.f;
        for (int r = 0; r < 8; ++r) g += red[r][threadIdx.x];
        if (grad_out) grad_out[c] = g;
        if (apply) b[c] = b[c] - lr * g;
    }
}

// 2x2 stride-2 max-pool, NHWC, H and W even, C % 4 == 0; idx = 2-bit position (dy*2+dx) of the first maximum.
// One thread per 4 channels (16-byte accesses; the index arithmetic is paid once per 4 elements).
__global__ void k_pool_fwd(const float* __restrict__ Y, float* __restrict__ P, uint8_t* __restrict__ idx, int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, C4 = C / 4;
    const long long total = (long long)N * OH * OW * C4;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        long long t = e / C4;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        const float* src = Y + ((n * H + 2 * oh) * W + 2 * ow) * (long long)C + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + C);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (long long)W * C), v3 = *reinterpret_cast<const f32x4*>(src + (long long)W * C + C);
        f32x4 best; uint8_t bi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float b = v0[i]; int k = 0;
            if (v1[i] > b) { b = v1[i]; k = 1; }
            if (v2[i] > b) { b = v2[i]; k = 2; }
            if (v3[i] > b) { b = v3[i]; k = 3; }
            best[i] = b; bi[i] = (uint8_t)k;
        }
        const long long o = e * 4;
        *reinterpret_cast<f32x4*>(P + o) = best;
        *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

// dZ[n,h,w,c] = (position is the arg-max of its window && pooled value > 0) ? dP : 0   (max-pool backward + ReLU mask).
// One thread per pooled element group of 4 channels writes all four window positions.
template <typename TS = float>       // storage type of all three tensors
__global__ void k_pool_bwd(const TS* __restrict__ dP, const TS* __restrict__ P, const uint8_t* __restrict__ idx, TS* __restrict__ dZ,
                           int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, C4 = C / 4;
    const long long total = (long long)N * OH * OW * C4;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        long long t = e / C4;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        const long long o = e * 4;
        const chunk4_t<TS> g = *reinterpret_cast<const chunk4_t<TS>*>(dP + o);
        const f32x4 pv = widen4(*reinterpret_cast<const chunk4_t<TS>*>(P + o));
        const uint32_t ii = *reinterpret_cast<const uint32_t*>(idx + o);
        TS* dst = dZ + ((n * H + 2 * oh) * W + 2 * ow) * (long long)C + c;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            chunk4_t<TS> v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (((ii >> (8 * i)) & 3u) == (uint32_t)pos && pv[i] > 0.f) ? g[i] : (TS)0.f;
            *reinterpret_cast<chunk4_t<TS>*>(dst + (long long)(pos >> 1) * W * C + (pos & 1) * C) = v;
        }
    }
}

// dZ = dY where the ReLU output is positive (layers without pooling)
__global__ void k_relu_bwd(const float* __restrict__ dY, const float* __restrict__ Y, float* __restrict__ dZ, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) dZ[e] = Y[e] > 0.f ? dY[e] : 0.f;
}

// fused softmax + cross-entropy: 32 lanes per sample (lane c takes classes c, c + 32, ...; the padded logits row is ldl wide), eight
// samples per 256-thread workgroup.  loss_part[block] = sum of -log p[label] over the block's samples in order; the workgroup that
// finishes LAST (a counter behind the partials, which it leaves at zero for the next launch) adds the partials in block order and
// writes the mean loss -- no second launch, and the sum does not depend on which workgroup that was.
// (One thread per sample -- three serial passes over the classes on two CUs -- took 11 us for 512 samples, plus 5 us for the
// one-thread k_sum_small behind it.)
__global__ __launch_bounds__(256) void k_softmax_ce(const float* __restrict__ logits, const int* __restrict__ labels, int B, int C, int ldl,
                                                    float* __restrict__ dlogits, float* loss_part, unsigned* counter, float inv_b, float* __restrict__ loss_out) {
    __shared__ float red[256];
    __shared__ int last;
    const int grp = threadIdx.x >> 5, ln = threadIdx.x & 31;
    const int s = blockIdx.x * 8 + grp;
    float loss = 0.f;
    if (s < B) {
        const float* z = logits + (long long)s * ldl;
        const int y = labels[s];
        float mx = -3.0e38f;
        for (int c = ln; c < C; c += 32) mx = z[c] > mx ? z[c] : mx;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { const float o = __shfl_xor(mx, off, 32); mx = o > mx ? o : mx; }
        float sum = 0.f;
        for (int c = ln; c < C; c += 32) sum += expf(z[c] - mx);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 32);
        loss = -(z[y] - mx - logf(sum));
        if (dlogits) {
            float* d = dlogits + (long long)s * ldl;
            for (int c = ln; c < ldl; c += 32) d[c] = c < C ? (expf(z[c] - mx) / sum - (c == y ? 1.f : 0.f)) * inv_b : 0.f;
        }
    }
    if (ln == 0) red[grp] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int g = 0; g < 8; ++g) t += red[g];
        __hip_atomic_store(&loss_part[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || !loss_out) {
        if (last && threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    float t = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) t += __hip_atomic_load(&loss_part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += red[i];
        *loss_out = tot * inv_b;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace rcnx

namespace rcnx {

// A gradient tensor that exists only at pooled resolution.  dZ of a conv layer followed by a 2x2 max-pool is
//     dZ[n, y, x, c] = (arg-max of its window == (y & 1, x & 1) && pooled value > 0) ? dP[n, y/2, x/2, c] : 0
// (k_pool_bwd); the LDS-tiled kernels can rebuild it while staging instead of reading a full-resolution copy that another
// kernel wrote: three reads at quarter resolution (4 + 4 + 1 bytes per channel) replace one write and one read of 4 x 4.
template <typename TS> struct PooledGradT {
    const TS* dP;             // [N][H/2][W/2][C] gradient wrt the pooled map; nullptr: the tensor is materialised, read it directly
    const TS* P;              // pooled activations (the ReLU gate)
    const uint8_t* idx;       // arg-max position 0..3 = dy * 2 + dx
};
using PooledGrad = PooledGradT<float>;


// The full-resolution values of one pooled gradient chunk (four channels): position pos = dy * 2 + dx of the window gets the pooled
// gradient where it was the arg-max and the pooled activation is positive (k_pool_bwd's rule), zero elsewhere.  All four positions at
// once: the ReLU gate and the arg-max field are taken once per channel, a compare + select per (position, channel) is what is left.
__device__ inline void unpool4x4(const f32x4& d, const f32x4& p, unsigned idx4, f32x4 (&v)[4]) {
    float m[4];
    unsigned k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] = p[i] > 0.f ? d[i] : 0.f; k[i] = (idx4 >> (8 * i)) & 3u; }
#pragma unroll
    for (int pos = 0; pos < 4; ++pos)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[pos][i] = k[i] == (unsigned)pos ? m[i] : 0.f;
}

// Arg-max bytes of four adjacent channels (the four lanes of a quad) as ONE dword store by the quad's first lane.  As byte stores --
// 64 one-byte requests per instruction, sixteen instructions per wave -- the fused-pool epilogue took 7 us per workgroup (timeline,
// tools/halo_stamps.hip) against 1-2 us for the float stores beside it.
__device__ inline void store_idx_quad(uint8_t* __restrict__ pool_idx, long long o, int bk, bool ok, int lane) {
    const int b0 = __builtin_amdgcn_mov_dpp(bk, 0x00, 0xf, 0xf, true), b1 = __builtin_amdgcn_mov_dpp(bk, 0x55, 0xf, 0xf, true);
    const int b2 = __builtin_amdgcn_mov_dpp(bk, 0xaa, 0xf, 0xf, true), b3 = __builtin_amdgcn_mov_dpp(bk, 0xff, 0xf, 0xf, true);
    if (ok && (lane & 3) == 0) *reinterpret_cast<uint32_t*>(pool_idx + o) = (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
}

}  // namespace rcnx
