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

// The same for every layer and both orientations in ONE launch at the start of a step (a launch per layer and pass was 5 us each, six of
// them on the CIFAR net: 9 % of the bf16 step).  The weights only change in the step's last kernel, so copies made at its start
// serve the forward and the backward pass.  Job q owns workgroups [first_block, next job's first_block), one 32 x 32 tile each.
constexpr int kMaxPrepJobs = 32;
struct PrepJob { const float* src; __bf16* dst; int R, C, Rp, first_block; };
struct PrepJobs { PrepJob j[kMaxPrepJobs]; int njobs; };
__host__ __device__ inline int prep_job_blocks(int C, int Rp) { return ((C + 31) / 32) * (Rp / 32); }      // 32 x 32 tiles (Rp is a multiple of 32)
__global__ __launch_bounds__(256) void k_prep_all_bf16(PrepJobs J) {
    // a workgroup transposes one 32 (r) x 32 (c) tile through LDS: rows of src read along c, rows of dst written along r -- both sides in
    // whole 64 / 128-byte pieces (element by element one side is strided: the 16 per-layer launches of the 224 x 224 net took 88 us, one
    // launch of the same loop 100)
    __shared__ float tile[32][33];
    int q = 0;
    while (q + 1 < J.njobs && (int)blockIdx.x >= J.j[q + 1].first_block) ++q;
    const PrepJob jb = J.j[q];
    const int lb = (int)blockIdx.x - jb.first_block, tiles_r = jb.Rp / 32;
    const int r0 = (lb % tiles_r) * 32, c0 = (lb / tiles_r) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + ty + 8 * u, c = c0 + tx;
        tile[ty + 8 * u][tx] = (r < jb.R && c < jb.C) ? jb.src[(long long)r * jb.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + ty + 8 * u, r = r0 + tx;
        if (c < jb.C) jb.dst[(long long)c * jb.Rp + r] = (__bf16)tile[tx][ty + 8 * u];
    }
}

__device__ inline bf16x4 to_bf16x4(const f32x4& v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
}

// (PooledGrad / unpool4 -- a gradient tensor that exists only at pooled resolution -- live in convnet.hpp: the fp32 LDS-tiled
// kernels use them too)

// Same tiling, split-K and epilogues as k_conv_fwd; WB = weights as bf16 [Cout][Kp], Kp = K rounded up to 32
// TX / TY: storage types of X and of Y (and of EPI 3's gate tensor, which arrives through `bias`); a split-K launch writes float partials
template <int KS, bool SMALLC, int BN, int EPI, typename TX = float, typename TY = float>
__global__ __launch_bounds__(kThreads) void k_conv_fwd_bf16(const TX* __restrict__ X, const __bf16* __restrict__ WB,
                                                            const float* __restrict__ bias, TY* __restrict__ Y, ConvShape s) {
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
    load_a_regs<KS, SMALLC, TX>(X, s, rows, kt0, tid, av);
    load_b(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            load_a_regs<KS, SMALLC, TX>(X, s, rows, kt0 + kt + 1, tid, av);
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
        const float bb = (EPI == 1 || EPI == 2) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wave * 32 + mfma32_row(lane, r);
            if (m < M) {
                float v = acc[t][r] + bb;
                if (EPI == 2) v = v > 0.f ? v : 0.f;
                if (EPI == 3) v = widen(reinterpret_cast<const TY*>(bias)[m * s.Cout + co]) > 0.f ? v : 0.f;      // dgrad: ReLU mask of the layer below, `bias` = its output
                Y[m * s.Cout + co] = narrow<TY>(v);
            }
        }
    }
}

}  // namespace rcnx

namespace rcnx {

using s16x4 = __attribute__((ext_vector_type(4))) short;

// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group a 4-row x 16-column block of 16-bit elements comes
// back column-major -- lane i of the group receives column i of the 4 rows.  Lane 4q+p supplies the address of row q,
// columns 4p..4p+3.  EXEC must be all ones; addresses 8-byte aligned.
__device__ inline s16x4 lds_read_tr16(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// One 32x32x16 operand fragment whose CONTRACTION index runs along the rows of a row-major LDS image: rows = pixels
// px0 .. px0+15, columns = col0 .. col0+31 (the operand's M or N index).  MFMA lane l (r = l & 31, h = l >> 5) needs rows
// px0 + 8h + j, j = 0..7, of column col0 + r: two transposed 4-row reads.
__device__ inline bf16x8 tr_fragment(const __bf16* img, int ld, int px0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a = img + (px0 + 8 * (g >> 1) + q) * ld + col0 + 16 * (g & 1) + 4 * p;
    const s16x4 lo = lds_read_tr16(a), hi = lds_read_tr16(a + 4 * ld);
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.v;
}

// dW partial tiles with bf16 operands: workgroup (kg, nb, chunk) of NKB waves computes rows [32 NKB kg, +32 NKB) x cols
// [BN nb, +BN) of dW over its pixel chunk -- wave w owns k-block NKB kg + w (a 32 x BN tile, no cross-wave reduction) --
// and writes slab[chunk][K + 1][Cout] exactly like k_conv_wgrad (row K = the chunk's bias gradient, summed in fp32 from
// the unrounded dZ values as they pass through registers).
//
// The contraction runs over PIXELS, which are the rows of both staged images (Xs[pixel][k], Ds[pixel][co], filled with
// plain 8-byte stores of converted fp32 rows); the MFMA wants 8 consecutive contraction indices per lane, i.e. a column of
// those images -- the hardware transposed read delivers exactly that, so no operand is ever transposed by software.
// A k-block lies inside one filter tap (Cin % 32 == 0), different waves' blocks may be different taps.
// TX: storage type of X (the first dense layer's input is the convolutional stage's last map)
template <int KS, int BN, int NKB, typename TX = float>
__global__ __launch_bounds__(64 * NKB) void k_conv_wgrad_bf16(const TX* __restrict__ X, const float* __restrict__ dZ,
                                                              float* __restrict__ slab, ConvShape s, int pix_per_chunk, WgradGrid gd) {
    constexpr int NT = BN / 32, kPT = 128, NTHR = 64 * NKB;
    constexpr int LDX = 32 * NKB + 8, LDD = BN + 8;
    constexpr int XCH = kPT * 8 * NKB / NTHR;                              // f32x4 chunks of the X tile per thread (= 16)
    constexpr int DCH = (kPT * (BN / 4) + NTHR - 1) / NTHR;               // ... of the dZ tile
    __shared__ __attribute__((aligned(16))) __bf16 Xs[kPT * LDX];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[kPT * LDD];
    __shared__ __attribute__((aligned(16))) float red[NTHR * 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long M = (long long)s.N * s.H * s.W;
    const int K = KS * KS * s.Cin, nkb = K / 32;
    int kgx, nby, chunk;
    if (!gd.decode((int)blockIdx.x, kgx, nby, chunk)) return;            // padding of the XCD-aware grid (uniform per workgroup)
    const int kb0 = kgx * NKB, n0 = nby * BN;
    const long long p0 = (long long)chunk * pix_per_chunk;
    const long long p1 = p0 + pix_per_chunk < M ? p0 + pix_per_chunk : M;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};

    // X loader: chunk e = tid + NTHR q -> pixel row e / (8 NKB), 4 floats at tile column 4 (e % (8 NKB)).  NTHR is a multiple
    // of 8 NKB, so a thread's tile column -- hence its k-block and filter tap -- is the same for all its chunks.
    const int xcol = (tid % (8 * NKB)) * 4;
    const int xkb = kb0 + xcol / 32;
    const bool xlive = xkb < nkb;
    const int k0 = (xlive ? xkb : 0) * 32, tap = k0 / s.Cin;
    const int dh = tap / KS - KS / 2, dw = tap % KS - KS / 2;
    const long long toff = ((long long)dh * s.W + dw) * s.Cin + (k0 - tap * s.Cin) + (xcol & 31);
    f32x4 xv[XCH], dv[DCH];
    // The thread's X rows are pixels m_first, m_first + 8, m_first + 16, ... across q AND across stages (128 = 16 * 8), so
    // the (oh, ow) of its next row is kept as running state and advanced by 8 pixels per load -- one 32-bit division per
    // kernel instead of two 64-bit ones per load (which used to cost more than the MFMAs).
    int row_m = (int)p0 + tid / (8 * NKB);
    int row_ow = row_m % s.W, row_oh = (row_m / s.W) % s.H;
    const int adv_h = 8 / s.W, adv_w = 8 % s.W;
    auto gload = [&](long long pb) {
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const bool in = xlive && row_m < (int)p1;
            const bool ok = in && (unsigned)(row_oh + dh) < (unsigned)s.H && (unsigned)(row_ow + dw) < (unsigned)s.W;
            const f32x4 val = widen4(*reinterpret_cast<const chunk4_t<TX>*>(X + (ok ? (long long)row_m * s.Cin + toff : 0)));   // unconditional, masked by value
            xv[q] = ok ? val : f32x4{0, 0, 0, 0};
            row_m += 8; row_ow += adv_w; row_oh += adv_h;
            if (row_ow >= s.W) { row_ow -= s.W; ++row_oh; }
            if (row_oh >= s.H) row_oh %= s.H;
        }
#pragma unroll
        for (int q = 0; q < DCH; ++q) {
            const int e = tid + NTHR * q;
            const int ec = e < kPT * (BN / 4) ? e : 0;
            const int pr = ec / (BN / 4), c4b = (ec - pr * (BN / 4)) * 4;
            const long long m = pb + pr;
            const f32x4 val = *reinterpret_cast<const f32x4*>(dZ + (m < p1 ? m : p0) * s.Cout + n0 + c4b);
            dv[q] = (m < p1 && e < kPT * (BN / 4)) ? val : f32x4{0, 0, 0, 0};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < XCH; ++q)
            *reinterpret_cast<bf16x4*>(&Xs[((tid + NTHR * q) / (8 * NKB)) * LDX + xcol]) = to_bf16x4(xv[q]);
#pragma unroll
        for (int q = 0; q < DCH; ++q) {
            const int e = tid + NTHR * q;
            if (e < kPT * (BN / 4)) {
                const int pr = e / (BN / 4), c4b = (e - pr * (BN / 4)) * 4;
                *reinterpret_cast<bf16x4*>(&Ds[pr * LDD + c4b]) = to_bf16x4(dv[q]);
                colsum += dv[q];                                             // fp32, unrounded: the bias gradient
            }
        }
    };

    const bool wlive = kb0 + wave < nkb;                                    // wave-uniform
    gload(p0);
    for (long long pb = p0; pb < p1; pb += kPT) {
        lstore();
        __syncthreads();
        if (pb + kPT < p1) gload(pb + kPT);                                 // next 128 pixels fly while these are contracted
        if (wlive) {
#pragma unroll
            for (int st = 0; st < kPT / 16; ++st) {
                const bf16x8 af = tr_fragment(Xs, LDX, 16 * st, 32 * wave, lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 bf = tr_fragment(Ds, LDD, 16 * st, 32 * t, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* out = slab + (long long)chunk * (K + 1) * s.Cout;
    if (wlive) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(long long)((kb0 + wave) * 32 + mfma32_row(lane, r)) * s.Cout + n0 + 32 * t + (lane & 31)] = acc[t][r];
    }
    if (kgx == 0) {
        // bias row: thread's four columns are c4b = 4 (tid % (BN/4)) (NTHR is a multiple of BN/4); add the threads in fixed order
        *reinterpret_cast<f32x4*>(&red[tid * 4]) = colsum;
        __syncthreads();
        if (tid < BN) {
            const int grp = tid >> 2, comp = tid & 3;
            float t = 0.f;
            for (int u = grp; u < NTHR; u += BN / 4) t += red[u * 4 + comp];
            out[(long long)K * s.Cout + n0 + tid] = t;
        }
    }
}

}  // namespace rcnx

namespace rcnx {

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3x3_halo_bf16 -- 3x3 convolution, LDS-tiled (channel blocks of 32 or 64).
//
// The implicit-GEMM kernel above re-reads every input pixel once per filter tap (nine shifted A tiles out of L2); for
// thin layers that traffic, not the MFMA, is the limit (5-7 % MFMA-busy by PMC at Cin = 32).  Here a workgroup owns an
// 8 x 16 block of output pixels of one image and stages the 10 x 18 input HALO of one channel block, as bf16, in LDS; all nine
// taps then read their A fragments out of that one image (a lane's eight consecutive channels of pixel (y + kh, x + kw):
// one ds_read_b128, no im2col anywhere), so global A traffic drops from 9 x 128 to 180 pixel rows per tile.  The bf16
// weights of one filter row (3 taps x BN x Cin) sit beside it and are restaged per filter row.  Wave w computes output
// rows 2w, 2w+1 of the block (32 pixels) x BN channels.  Same operands, rounding and epilogues as k_conv_fwd_bf16, so
// the two are interchangeable; used for forward and (on dZ with the flipped weights) for the input gradient.
constexpr int kHaloTH = 8, kHaloTW = 16;

template <int CB, int BN, int EPI, bool PIN = false>
__global__ __launch_bounds__(kThreads) void k_conv3x3_halo_bf16(const float* __restrict__ X, const __bf16* __restrict__ WB,
                                                                const float* __restrict__ bias, float* __restrict__ Y, ConvShape s, int tiles_w,
                                                                int tiles_h, uint8_t* __restrict__ pool_idx, PooledGrad pin) {
    static_assert(CB % 16 == 0 && BN % 32 == 0, "channel blocks of the 32x32x16 MFMA");
    constexpr int NT = BN / 32, LDC = CB + 8;                        // halves per pixel / per weight row in LDS (16-byte aligned, bank-skewed)
    constexpr int HH = kHaloTH + 2, HW = kHaloTW + 2;
    constexpr int HCH = HH * HW * (CB / 4);                           // f32x4 chunks of one channel block of the halo
    constexpr int BCH = 3 * BN * (CB / 8);                            // 16-byte chunks of one filter row's weights for that block
    __shared__ __attribute__((aligned(16))) __bf16 Hs[HH * HW * LDC];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3 * BN * LDC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int tile = blockIdx.x;
    const int tw = tile % tiles_w; tile /= tiles_w;
    const int th = tile % tiles_h;
    const int img = tile / tiles_h;
    const int oh0 = th * kHaloTH, ow0 = tw * kHaloTW, n0 = blockIdx.y * BN;
    const int Cin = s.Cin, K = 9 * Cin;                               // Cin: a multiple of CB; K = row length of WB

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    const int py = 2 * wave + (r >> 4), px = r & 15;                  // this lane's A row = output pixel (py, px) of the block
#pragma unroll 1
    for (int cb = 0; cb < Cin; cb += CB) {
        if (cb) __syncthreads();                                      // the previous channel block's halo has been consumed
        // ---- halo: pixel (oh0 - 1 + hy, ow0 - 1 + hx), channels cb .. cb + CB - 1, zero outside the image; unconditional
        // loads from clamped addresses
        if (PIN) {
            // the input is a pooled-resolution gradient: one thread loads (dP, P, arg-max) of ONE pooled pixel's four channels -- the 6 x 10
            // pooled pixels under the halo -- and expands it into the up to four halo pixels of its window (round 3: a load triple per
            // halo pixel had fetched every pooled value four times; these kernels were 1.6 x their forward twins)
            constexpr int PPW = kHaloTW / 2 + 2, PCH = 6 * PPW * (CB / 4);
            for (int e = tid; e < PCH; e += kThreads) {
                const int pp = e / (CB / 4), c4 = (e - pp * (CB / 4)) * 4;
                const int pr = pp / PPW, pc = pp - pr * PPW;
                const int poh = (oh0 >> 1) - 1 + pr, pow_ = (ow0 >> 1) - 1 + pc;
                const bool ok = (unsigned)poh < (unsigned)(s.H >> 1) && (unsigned)pow_ < (unsigned)(s.W >> 1);
                const long long o = ok ? (((long long)img * (s.H >> 1) + poh) * (s.W >> 1) + pow_) * Cin + cb + c4 : 0;
                const f32x4 d = *reinterpret_cast<const f32x4*>(pin.dP + o), pv = *reinterpret_cast<const f32x4*>(pin.P + o);
                const unsigned ii = *reinterpret_cast<const unsigned*>(pin.idx + o);
                f32x4 v[4];
                unpool4x4(ok ? d : f32x4{0, 0, 0, 0}, pv, ii, v);
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const int hy = 2 * pr - 1 + (pos >> 1), hx = 2 * pc - 1 + (pos & 1);
                    if ((unsigned)hy < (unsigned)HH && (unsigned)hx < (unsigned)HW)
                        *reinterpret_cast<bf16x4*>(&Hs[(hy * HW + hx) * LDC + c4]) = to_bf16x4(v[pos]);
                }
            }
        } else {
            for (int e = tid; e < HCH; e += kThreads) {
                const int pix = e / (CB / 4), c4 = (e - pix * (CB / 4)) * 4;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
                const bool ok = (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(X + (ok ? (((long long)img * s.H + ih) * s.W + iw) * Cin + cb + c4 : 0));
                *reinterpret_cast<bf16x4*>(&Hs[pix * LDC + c4]) = to_bf16x4(ok ? v : f32x4{0, 0, 0, 0});
            }
        }
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            if (kh) __syncthreads();                                  // everyone is done with the previous filter row's weights
            for (int e = tid; e < BCH; e += kThreads) {
                const int row = e / (CB / 8), c8 = (e - row * (CB / 8)) * 8;    // row = kw * BN + co
                const int kw = row / BN, co = row - kw * BN;
                *reinterpret_cast<bf16x8*>(&Bs[row * LDC + c8]) =
                    *reinterpret_cast<const bf16x8*>(WB + (long long)(n0 + co) * K + (kh * 3 + kw) * Cin + cb + c8);
            }
            __syncthreads();
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const __bf16* a = &Hs[((py + kh) * HW + px + kw) * LDC + 8 * h];
                const __bf16* b = &Bs[(kw * BN + r) * LDC + 8 * h];
#pragma unroll
                for (int ks = 0; ks < CB / 16; ++ks) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + 16 * ks);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b + 32 * t * LDC + 16 * ks);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- epilogue: accumulator row i of lane = block pixel mfma32_row(lane, i) of this wave's 32
    if (EPI == 4) {
        // bias + ReLU + the 2x2 max-pool that follows, fused: a lane's sixteen rows are columns 4h..4h+3 and 8+4h..8+4h+3 of BOTH
        // pixel rows of its wave, i.e. four complete pooling windows -- the pool is a max over the lane's own registers.  Writes the
        // pooled map and the arg-max image exactly as k_pool_fwd does (first maximum in the order 00, 01, 10, 11), so k_pool_bwd is
        // unchanged; the un-pooled activation is never written.
        const int OH = s.H / 2, OW = s.W / 2;
        const int poh = oh0 / 2 + wave;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = n0 + 32 * t + (lane & 31);
            const float bb = bias[co];
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int i0 = 4 * gq + 2 * pp;
                    float v[4] = {acc[t][i0] + bb, acc[t][i0 + 1] + bb, acc[t][8 + i0] + bb, acc[t][8 + i0 + 1] + bb};
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
                    float best = v[0];
                    int bk = 0;
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (v[k] > best) { best = v[k]; bk = k; }
                    const int pow_ = ow0 / 2 + 2 * h + 4 * gq + pp;
                    const bool ok = poh < OH && pow_ < OW;
                    const long long o = (((long long)img * OH + poh) * OW + pow_) * s.Cout + co;
                    if (ok) Y[o] = best;
                    store_idx_quad(pool_idx, o, bk, ok, lane);
                }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = n0 + 32 * t + (lane & 31);
        const float bb = (EPI == 1 || EPI == 2) ? bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int pr = mfma32_row(lane, i);
            const int oh = oh0 + 2 * wave + (pr >> 4), ow = ow0 + (pr & 15);
            if (oh < s.H && ow < s.W) {
                const long long m = ((long long)img * s.H + oh) * s.W + ow;
                float v = acc[t][i] + bb;
                if (EPI == 2) v = v > 0.f ? v : 0.f;
                if (EPI == 3) v = bias[m * s.Cout + co] > 0.f ? v : 0.f;
                Y[m * s.Cout + co] = v;
            }
        }
    }
}

}  // namespace rcnx

namespace rcnx {

// ---------------------------------------------------------------------------------------------------------------------
// k_wgrad3x3_halo_bf16 -- weight gradient of a 3x3 layer, LDS-tiled: the companion of k_conv3x3_halo_bf16.
//
// k_conv_wgrad_bf16 gives every 32-row k-block (= one filter tap of 32 channels) its own staged X tile, i.e. reads the
// input nine times and dZ once per k-group.  Here a workgroup stages, per 8 x 16 block of output pixels, the 10 x 18 input
// halo of one channel block and the block's dZ ONCE, and its NINE waves each contract one filter tap over the block:
// wave (kh, kw) reads its A fragments -- 16 consecutive pixels of halo row py + kh starting at column kw, 32 channels -- and
// the shared dZ fragments with the hardware transposed LDS read (the contraction index, the pixel, is the row of both LDS
// images), accumulating dW[tap][ci][co] in registers across all the blocks of its chunk.  Global traffic per block: 180
// input pixels + 128 dZ pixels instead of 9 x 128 + 128 per k-group.  Output: the same slab[chunk][K + 1][Cout] as
// k_conv_wgrad (row K = bias partial, fp32 sums of the unrounded dZ), so k_reduce_all finishes either.
constexpr int kWgHaloThreads = 9 * 64;

__device__ inline bf16x4 stored_bf16x4(const f32x4& v) { return to_bf16x4(v); }
__device__ inline bf16x4 stored_bf16x4(const bf16x4& v) { return v; }

// TS: storage type of X, dZ and the pooled gradient (convnet.hpp, Chunk4)
template <int CB, int BN, bool PDZ = false, typename TS = float>
__global__ __launch_bounds__(kWgHaloThreads) void k_wgrad3x3_halo_bf16(const TS* __restrict__ X, const TS* __restrict__ dZ,
                                                                       float* __restrict__ slab, ConvShape s, int tiles_w, int tiles_h,
                                                                       int blocks_per_chunk, int n_chunks, PooledGradT<TS> pdz) {
    constexpr int NA = CB / 32, NT = BN / 32, LDC = CB + 8, LDD = BN + 8;
    constexpr int HH = kHaloTH + 2, HW = kHaloTW + 2, NPX = kHaloTH * kHaloTW;
    constexpr int HCH = HH * HW * (CB / 4), DCH = NPX * (BN / 4);
    __shared__ __attribute__((aligned(16))) __bf16 Hs[HH * HW * LDC];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[NPX * LDD];
    __shared__ __attribute__((aligned(16))) float red[kWgHaloThreads * 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kh = wave / 3, kw = wave - 3 * kh;                      // this wave's filter tap
    const int cb = blockIdx.x * CB, n0 = blockIdx.y * BN, chunk = blockIdx.z;
    const int Cin = s.Cin, K = 9 * Cin;
    const int total_blocks = tiles_w * tiles_h * s.N;
    const int b0 = chunk * blocks_per_chunk, b1 = b0 + blocks_per_chunk < total_blocks ? b0 + blocks_per_chunk : total_blocks;

    f32x16 acc[NA][NT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};

    // Staging in two steps (round 3): the global loads of block blk + 1 are issued into registers before block blk is contracted and
    // stored to LDS after it -- with one 576-thread workgroup per CU (64 accumulator registers per wave) nothing else covered their
    // latency.  A thread's chunks: halo chunk q = 4 channels of halo pixel (tid + 576 q) / (CB/4); dZ chunk q likewise over the block's
    // 128 pixels, or, PDZ, ONE pooled pixel's (dP, P, arg-max), expanded into its window when stored.
    constexpr int NHQ = (HCH + kWgHaloThreads - 1) / kWgHaloThreads, NDQ = PDZ ? 1 : (DCH + kWgHaloThreads - 1) / kWgHaloThreads;
    static_assert(!PDZ || 32 * (BN / 4) <= kWgHaloThreads, "one pooled chunk per thread");
    int hrel[NHQ];                                                    // (hy * W + hx) * Cin + c4, or -1: no such chunk; hy | hx << 8 in hpk
    int hpk[NHQ];
#pragma unroll
    for (int q = 0; q < NHQ; ++q) {
        const int e = tid + kWgHaloThreads * q;
        const int pix = e / (CB / 4), c4 = (e - pix * (CB / 4)) * 4;
        const int hy = pix / HW, hx = pix - hy * HW;
        hpk[q] = e < HCH ? (hy | (hx << 8)) : -1;
        hrel[q] = (hy * s.W + hx) * Cin + c4;
    }
    // S blocks' operands are in flight at once (round 4; one was: a 32-channel layer's block is 8 MFMAs per wave against a load latency
    // of more than a microsecond, with ONE workgroup per CU -- the kernel ran at exactly blocks x latency).  A stage is a handful of
    // registers, refilled with the block S further on as soon as its own block has gone to LDS; every gload issues the same number of
    // loads whatever the block (past the chunk's last block: that block again), so the wait in front of a stage's LDS stores is an exact
    // count that leaves the younger stages in flight.  (64 x 64 tiles: 64 accumulator registers, and a block is 32 MFMAs per wave; fp32
    // tensors: twice the registers per stage.)
    constexpr int S = sizeof(TS) == 4 ? (NA * NT == 4 ? 1 : 2) : (NA * NT == 1 ? 4 : NA * NT == 2 ? 3 : PDZ ? 2 : 1);
    struct Stage { chunk4_t<TS> hv[NHQ], dv[NDQ], dpv; unsigned dii, okm; };
    Stage stg[S];
    auto gload = [&](int blk, Stage& g) {
        chunk4_t<TS> (&hv)[NHQ] = g.hv; chunk4_t<TS> (&dv)[NDQ] = g.dv; chunk4_t<TS>& dpv = g.dpv; unsigned& dii = g.dii; unsigned& okm = g.okm;
        int q0 = blk;
        const int tw = q0 % tiles_w; q0 /= tiles_w;
        const int th = q0 % tiles_h;
        const int img = q0 / tiles_h;
        const int oh0 = th * kHaloTH, ow0 = tw * kHaloTW;
        const long long hbase = (((long long)img * s.H + oh0 - 1) * s.W + ow0 - 1) * Cin + cb;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NHQ; ++q) {
            const int pk = hpk[q];
            const bool ok = pk >= 0 && (unsigned)(oh0 - 1 + (pk & 255)) < (unsigned)s.H && (unsigned)(ow0 - 1 + (pk >> 8)) < (unsigned)s.W;
            okm |= (ok ? 1u : 0u) << q;
            hv[q] = *reinterpret_cast<const chunk4_t<TS>*>(X + (ok ? hbase + hrel[q] : 0));
        }
        if (PDZ) {
            const int pp = tid / (BN / 4), c4 = (tid - pp * (BN / 4)) * 4;
            const int poh = (oh0 >> 1) + (pp >> 3), pow_ = (ow0 >> 1) + (pp & 7);
            const bool ok = tid < 32 * (BN / 4) && poh < (s.H >> 1) && pow_ < (s.W >> 1);
            okm |= (ok ? 1u : 0u) << 16;
            const long long o = ok ? (((long long)img * (s.H >> 1) + poh) * (s.W >> 1) + pow_) * s.Cout + n0 + c4 : 0;
            dv[0] = *reinterpret_cast<const chunk4_t<TS>*>(pdz.dP + o);
            dpv = *reinterpret_cast<const chunk4_t<TS>*>(pdz.P + o);
            dii = *reinterpret_cast<const unsigned*>(pdz.idx + o);
        } else {
#pragma unroll
            for (int q = 0; q < NDQ; ++q) {
                const int e = tid + kWgHaloThreads * q;
                const int pix = e / (BN / 4), c4 = (e - pix * (BN / 4)) * 4;
                const int oh = oh0 + pix / kHaloTW, ow = ow0 + pix % kHaloTW;
                const bool ok = e < DCH && oh < s.H && ow < s.W;
                okm |= (ok ? 1u : 0u) << (16 + q);
                dv[q] = *reinterpret_cast<const chunk4_t<TS>*>(dZ + (ok ? (((long long)img * s.H + oh) * s.W + ow) * s.Cout + n0 + c4 : 0));
            }
        }
    };
    auto lstore = [&](const Stage& g) {
        const chunk4_t<TS> (&hv)[NHQ] = g.hv; const chunk4_t<TS> (&dv)[NDQ] = g.dv; const chunk4_t<TS>& dpv = g.dpv; const unsigned dii = g.dii, okm = g.okm;
#pragma unroll
        for (int q = 0; q < NHQ; ++q) {
            const int e = tid + kWgHaloThreads * q;
            const int pix = e / (CB / 4), c4 = (e - pix * (CB / 4)) * 4;
            if (NHQ * kWgHaloThreads == HCH || e < HCH) *reinterpret_cast<bf16x4*>(&Hs[pix * LDC + c4]) = ((okm >> q) & 1u) ? stored_bf16x4(hv[q]) : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
        if (PDZ) {
            if (tid < 32 * (BN / 4)) {
                const int pp = tid / (BN / 4), c4 = (tid - pp * (BN / 4)) * 4;
                const int ppy = pp >> 3, ppx = pp & 7;
                f32x4 v[4];
                unpool4x4(((okm >> 16) & 1u) ? widen4(dv[0]) : f32x4{0, 0, 0, 0}, widen4(dpv), dii, v);
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    *reinterpret_cast<bf16x4*>(&Ds[((2 * ppy + (pos >> 1)) * kHaloTW + 2 * ppx + (pos & 1)) * LDD + c4]) = to_bf16x4(v[pos]);
                    colsum += v[pos];                                 // fp32, unrounded: the bias gradient (a thread's columns are the same for every block)
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NDQ; ++q) {
                const int e = tid + kWgHaloThreads * q;
                const int pix = e / (BN / 4), c4 = (e - pix * (BN / 4)) * 4;
                if (NDQ * kWgHaloThreads == DCH || e < DCH) {
                    const f32x4 v = ((okm >> (16 + q)) & 1u) ? widen4(dv[q]) : f32x4{0, 0, 0, 0};
                    *reinterpret_cast<bf16x4*>(&Ds[pix * LDD + c4]) = to_bf16x4(v);
                    colsum += v;                                      // fp32, unrounded: the bias gradient (same columns every time: 576 % (BN/4) == 0)
                }
            }
        }
    };

    if (b0 < b1) {
#pragma unroll
        for (int u = 0; u < S; ++u) gload(b0 + u < b1 ? b0 + u : b1 - 1, stg[u]);
    }
#pragma unroll 1
    for (int blk0 = b0; blk0 < b1; blk0 += S)
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int blk = blk0 + u;
        if (blk >= b1) break;
        if (blk != b0) __syncthreads();                               // the previous block's images have been consumed
        lstore(stg[u]);
        __syncthreads();
        gload(blk + S < b1 ? blk + S : b1 - 1, stg[u]);              // S blocks ahead, under this and the next blocks' MFMAs
#pragma unroll
        for (int py = 0; py < kHaloTH; ++py) {                        // one 16-pixel contraction step per block row
            bf16x8 af[NA], bf[NT];
#pragma unroll
            for (int a = 0; a < NA; ++a) af[a] = tr_fragment(Hs + ((py + kh) * HW + kw) * LDC, LDC, 0, 32 * a, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = tr_fragment(Ds + py * kHaloTW * LDD, LDD, 0, 32 * t, lane);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[t], acc[a][t], 0, 0, 0);
        }
    }
    float* out = slab + (long long)chunk * (K + 1) * s.Cout;
    const int krow0 = (kh * 3 + kw) * Cin + cb;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(long long)(krow0 + 32 * a + mfma32_row(lane, r)) * s.Cout + n0 + 32 * t + (lane & 31)] = acc[a][t][r];
    if (blockIdx.x == 0) {                                            // bias row: thread's four columns are 4 (tid % (BN/4)); fixed-order sum
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[tid * 4]) = colsum;
        __syncthreads();
        if (tid < BN) {
            const int grp = tid >> 2, comp = tid & 3;
            float t = 0.f;
            for (int u = grp; u < kWgHaloThreads; u += BN / 4) t += red[u * 4 + comp];
            out[(long long)K * s.Cout + n0 + tid] = t;
        }
    }
}

}  // namespace rcnx
