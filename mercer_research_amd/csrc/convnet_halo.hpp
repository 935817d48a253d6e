// convnet_halo.hpp -- Track X, fp32 MFMA path: LDS-tiled 3x3 convolution kernels (round 3; north_star: "coalesced HBM loads of
// image / filter tiles into LDS", configs[2] "LDS-tiled filters").  No reference counterpart; parity is against
// oracle/convnet_oracle.py like the rest of Track X.
//
// The implicit-GEMM kernels of convnet.hpp re-gather every input pixel once per filter tap: nine shifted 128 x 32 A tiles out of L2
// per K-sweep (PMC, round 2: 7.4 VALU instructions per MFMA, 48 % of the wave cycles waiting, ~21 TB/s of L2 traffic over the chip).
// Here a workgroup owns a block of 128 output pixels -- 8 x 16 of one image, or, for narrow maps, 8 x 8 of TWO images side by side
// (HaloGeom<8>) -- and stages the input HALO of one 32-channel block ONCE; all nine taps read their operands out of that image.
//
//   k_conv3x3_halo_f32    forward, and the input gradient (on dZ with the tap-flipped transposed weights)
//   k_wgrad3x3_halo_f32   weight gradient: one staged halo + dZ block serves all nine taps of a 32 x 32 (ci, co) tile
//   k_conv1_fwd_f32 / k_conv1_wgrad_f32   the first layer (9 * Cin <= 32: the whole patch is ONE k-block), organised around its
//                         activations: the weights live in registers, the input halo of a block is a few hundred floats
#pragma once

#include "convnet.hpp"

// Timeline instrumentation for tools/halo_stamps.hip (never defined in the library build): thread 0 of a workgroup records the
// 100 MHz wall clock at a few points of its life.
#ifdef RCNX_STAMPS
__device__ unsigned long long* g_rcnx_stamps = nullptr;      // [workgroup][32]: 30 stamps, [30] = XCC id, [31] = HW_ID
#define RCNX_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (g_rcnx_stamps && threadIdx.x == 0 && (slot) < 30) { unsigned long long t_; \
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_rcnx_stamps[(size_t)blockIdx.x * 32 + (slot)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define RCNX_STAMP_HW() do { if (g_rcnx_stamps && threadIdx.x == 0) { unsigned x_, h_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_)); g_rcnx_stamps[(size_t)blockIdx.x * 32 + 30] = x_ & 0xf; g_rcnx_stamps[(size_t)blockIdx.x * 32 + 31] = h_; } } while (0)
#else
#define RCNX_STAMP(slot) do { } while (0)
#define RCNX_STAMP_HW() do { } while (0)
#endif

namespace rcnx {

template <int TW> struct HaloGeom {
    static constexpr int NIMG = 16 / TW;                  // images per block (TW = 16: one, TW = 8: two side by side)
    static constexpr int TH = 8, HH = TH + 2, HWD = TW + 2, IMG_PIX = HH * HWD;
    // LDS pixel stride between the two images of a TW = 8 block: 104 = 8 (mod 16), so that the 16 lanes of one ds_read_b128 pass
    // (8 pixels of image 0, 8 of image 1, 36-float rows) fall into 16 different 4-bank groups
    static constexpr int IS = TW == 16 ? IMG_PIX : 104;
    static constexpr int NPIX = NIMG * IMG_PIX;            // staged pixels
    static constexpr int LPIX = (NIMG - 1) * IS + IMG_PIX; // LDS pixel slots
    __device__ static int lds_pix(int pix) { return TW == 16 ? pix : pix + (pix >= IMG_PIX ? IS - IMG_PIX : 0); }
};

// Reading the thread index through an empty asm makes it opaque: what is computed from it is redone where it is used instead of
// being hoisted out of the loops into registers that then spill (first-layer kernels: a handful of integer instructions).
__device__ inline int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// One thread's share of a staged pixel grid (ROWS x COLS pixels of each of NIMG images, 32 channels): chunk q is channels
// 4 (tid & 7) .. + 3 of grid pixel (tid >> 3) + 32 q.  Decoded ONCE per kernel into one register per chunk -- the pixel's
// (row, column, image), packed -- so that staging a grid costs a few integer instructions per chunk and no division: every VALU instruction issued between the MFMAs takes an issue slot from them (timeline + PMC, round 3:
// four VALU instructions per MFMA held these kernels at 60-65 % of the MFMA rate).
template <int NQ> struct StageMap {
    int pk[NQ];               // row | column << 8 | image << 16 of the chunk's pixel; negative: no such pixel
};
template <int NQ, int ROWS, int COLS, int NIMG, int CPP = 8>          // CPP = 16-byte chunks per pixel (channels of the block / 4)
__device__ inline void stage_map_init(StageMap<NQ>& m, int tid) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pix = tid / CPP + (kThreads / CPP) * q;
        const int i = pix / (ROWS * COLS), pr = pix - i * (ROWS * COLS);
        const int y = pr / COLS, x = pr - y * COLS;
        m.pk[q] = pix < NIMG * ROWS * COLS ? (y | (x << 8) | (i << 16)) : -1;
    }
}
// Element offset of a packed pixel relative to the grid's origin, and whether it lies inside the image, for a grid whose origin is
// pixel (y0, x0) of image img0.  Callers pass pk through opaque(): otherwise these few instructions are hoisted out of the loops
// too, into registers the kernels do not have.
template <int CPP = 8> __device__ inline int stage_rel(int pk, int tid, int Himg, int Wimg, int C) {
    return (((pk >> 16) * Himg + (pk & 255)) * Wimg + ((pk >> 8) & 255)) * C + (tid % CPP) * 4;
}
__device__ inline bool stage_ok(int pk, int y0, int x0, int img0, int Himg, int Wimg, int N) {
    return pk >= 0 && (unsigned)(y0 + (pk & 255)) < (unsigned)Himg && (unsigned)(x0 + ((pk >> 8) & 255)) < (unsigned)Wimg && img0 + (pk >> 16) < N;
}

// Epilogue of the LDS-tiled forward / input-gradient kernels (fp32 and bf16 operands alike: the accumulators are fp32): accumulator
// row i of a lane is block pixel mfma32_row(lane, i) of its wave's 32 pixels (block rows 2 wave, 2 wave + 1).
//
// Addresses (round 4).  Row i sits at block column 4 h + (i & 3) + 8 ((i >> 2) & 1) and block row 2 wave + (i >> 3), so its element
// offset is the lane's offset of row 0 plus a delta that is the SAME for every lane -- a few scalar products of the shape.  Computed
// per row as ((img * H + oh) * W + ow) * Cout + co the compiler used v_mad_u64_u32 with an undefined upper half for every row, and
// whichever register it picked for that half usually had one of the NEXT item's prefetched loads pending: an s_waitcnt vmcnt(0) in
// front of every store (ISA) -- and four vector instructions per row where one add does.
template <int TW> struct EpiRows {
    int img, oh, ow;          // row 0 of the lane
    unsigned base;            // its element offset at channel `co`
    __device__ EpiRows(int lane, int wave, int img0, int oh0, int ow0, const ConvShape& s, int co)
        : img(img0), oh(oh0 + 2 * wave), ow(ow0 + 4 * (lane >> 5)), base((unsigned)(((img0 * s.H + oh0 + 2 * wave) * s.W + ow0 + 4 * (lane >> 5)) * s.Cout + co)) {}
    // (i is a compile-time constant where this is used: scalar arithmetic)
    __device__ static unsigned delta(int i, const ConvShape& s) {
        const int dc = i & 3, half = (i >> 2) & 1, dy = i >> 3;
        return TW == 16 ? (unsigned)((dy * s.W + dc + 8 * half) * s.Cout) : (unsigned)(((half * s.H + dy) * s.W + dc) * s.Cout);
    }
    __device__ bool ok(int i, const ConvShape& s) const {
        const int dc = i & 3, half = (i >> 2) & 1, dy = i >> 3;
        return (TW == 16 ? img : img + half) < s.N && oh + dy < s.H && (TW == 16 ? ow + dc + 8 * half : ow + dc) < s.W;
    }
};

// EPI 3's gate values (the layer below's output at the positions this lane is about to write) loaded AHEAD of the item's last MFMAs:
// inside the epilogue their latency sat on every item's critical path (a 32-channel layer at 224 x 224: 18 MFMAs per item, then
// sixteen dependent loads, then the stores).
template <int TW, int NT, typename TG = float>
__device__ inline void halo_gate_prefetch(float (&g)[NT][16], int lane, int wave, int img0, int oh0, int ow0, int n0, const ConvShape& s,
                                          const TG* __restrict__ G) {
    if constexpr (sizeof(TG) == 2) {
        // bf16 tensors: the epilogue writes channel PAIRS (halo_epilogue, below) -- lane (l, l ^ 1) = channels (co, co + 1): the even lane
        // rows 2j, the odd lane rows 2j + 1 -- so a lane's gates are the eight dwords at (its row of pair j, co & ~1): g[t][j], as bits
        const int odd = lane & 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            EpiRows<TW> R(lane, wave, img0, oh0, ow0, s, n0 + 32 * t + (lane & 30));
            R.ow += odd; R.base += odd ? (unsigned)s.Cout : 0u;       // row 2j + 1 is one block column to the right of row 2j
#pragma unroll
            for (int j = 0; j < 8; ++j)
                g[t][j] = __uint_as_float(*reinterpret_cast<const unsigned*>(G + (R.ok(2 * j, s) ? R.base + R.delta(2 * j, s) : 0u)));
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const EpiRows<TW> R(lane, wave, img0, oh0, ow0, s, n0 + 32 * t + (lane & 31));
#pragma unroll
        for (int i = 0; i < 16; ++i) g[t][i] = widen(G[R.ok(i, s) ? R.base + R.delta(i, s) : 0u]);
    }
}

// TY: storage type of Y; TG: of EPI 3's gate tensor, which arrives through `bias` (convnet.hpp, Chunk4).
template <int TW, int NT, int EPI, typename TY = float, typename TG = float>
__device__ inline void halo_epilogue(const f32x16 (&acc)[NT], int lane, int wave, int img0, int oh0, int ow0, int n0, const ConvShape& s,
                                     const float* __restrict__ bias, TY* __restrict__ Y, uint8_t* __restrict__ pool_idx,
                                     const float (*gate)[16] = nullptr, const float* bias_ready = nullptr) {      // bias_ready[t]: the lane's bias, already in registers
    const int h = lane >> 5;
    // EPI 4 (bias + ReLU + the 2x2 max-pool that follows): a lane's sixteen rows are block columns 4h..4h+3 and 8+4h..8+4h+3 of BOTH pixel
    // rows of its wave -- four complete pooling windows (gq, pp) whose left pixel is block column 4 h + 8 gq + 2 pp (TW = 8: columns 0..7
    // are image 0, 8..15 image 1: still whole windows).  Pooled offset = the lane's offset of window (0, 0) + a uniform delta.
    const int OH = s.H / 2, OW = s.W / 2;
    const int poh = oh0 / 2 + wave, pow0 = ow0 / 2 + 2 * h;
    auto pool_base = [&](int co) { return (unsigned)(((img0 * OH + poh) * OW + pow0) * s.Cout + co); };
    auto pool_delta = [&](int gq, int pp) { return TW == 16 ? (unsigned)((4 * gq + pp) * s.Cout) : (unsigned)((gq * OH * OW + pp) * s.Cout); };
    auto pool_ok = [&](int gq, int pp) { return (TW == 16 ? img0 : img0 + gq) < s.N && poh < OH && (TW == 16 ? pow0 + 4 * gq + pp : pow0 + pp) < OW; };
    if constexpr (sizeof(TY) == 2) {
        // bf16 output: two-byte stores are half a dword each -- sixteen store instructions per tile for 2 KB.  Lanes l and l ^ 1 hold
        // adjacent channels of the same sixteen rows: they swap half of their values (one DPP move per row pair) so that the even lane
        // holds both channels of rows 2j and the odd lane both channels of rows 2j + 1, and each writes eight DWORDS (the gate tensor
        // of EPI 3 is read the same way).  Same values, same rounding.
        static_assert(EPI != 3 || sizeof(TG) == 2, "a bf16 gradient is gated by a bf16 map");
        using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
        const int odd = lane & 1;
        auto swap = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); };   // quad_perm [1, 0, 3, 2]
        if (EPI == 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = n0 + 32 * t + (lane & 31);
                const float bb = bias_ready ? bias_ready[t] : bias[co];
                const unsigned pb = pool_base(co), pb2 = pool_base(co & ~1);
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    float best[2];
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const int i0 = 4 * gq + 2 * pp;
                        const float v[4] = {acc[t][i0], acc[t][i0 + 1], acc[t][8 + i0], acc[t][8 + i0 + 1]};
                        float b = v[0];
                        int bk = 0;
#pragma unroll
                        for (int k = 1; k < 4; ++k)
                            if (v[k] > b) { b = v[k]; bk = k; }
                        b += bb;
                        best[pp] = b > 0.f ? b : 0.f;
                        store_idx_quad(pool_idx, pb + pool_delta(gq, pp), bk, pool_ok(gq, pp), lane);
                    }
                    // the even lane writes window pp = 0 for both channels, the odd lane window pp = 1
                    const float recv = swap(odd ? best[0] : best[1]);
                    bf16x2 pk = {(__bf16)(odd ? recv : best[0]), (__bf16)(odd ? best[1] : recv)};
                    asm volatile("" : "+v"(pk));
                    const bool ok = odd ? pool_ok(gq, 1) : pool_ok(gq, 0);
                    const unsigned o = pb2 + (odd ? pool_delta(gq, 1) : pool_delta(gq, 0));
                    if (ok) *reinterpret_cast<bf16x2*>(Y + o) = pk;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = n0 + 32 * t + (lane & 31);
                const float bb = (EPI == 1 || EPI == 2) ? (bias_ready ? bias_ready[t] : bias[co]) : 0.f;
                EpiRows<TW> R(lane, wave, img0, oh0, ow0, s, co & ~1);
                R.ow += odd; R.base += odd ? (unsigned)s.Cout : 0u;   // this lane's row of pair j is 2j + odd: one block column to the right
                // every gate word FIRST, in one burst: loaded where it is used, each load sat between the previous row's store and its own
                // s_waitcnt vmcnt(0) (the compiler may not move a load above a store it cannot tell apart) -- eight serial round trips per
                // tile, and, vmcnt being in order, each of them also waited for the next phase's prefetched operands (ISA, round 4)
                unsigned gb[8];
                if (EPI == 3 && !gate) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        gb[j] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const TG*>(bias) + (R.ok(2 * j, s) ? R.base + R.delta(2 * j, s) : 0u));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v0 = acc[t][2 * j] + bb, v1 = acc[t][2 * j + 1] + bb;
                    if (EPI == 2) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
                    const float recv = swap(odd ? v0 : v1);
                    float lo = odd ? recv : v0, hi = odd ? v1 : recv;                 // channels co & ~1, (co & ~1) + 1 of this lane's row
                    if (EPI == 3) {
                        const unsigned g = gate ? __float_as_uint(gate[t][j]) : gb[j];
                        lo = __uint_as_float(g << 16) > 0.f ? lo : 0.f;
                        hi = __uint_as_float(g & 0xffff0000u) > 0.f ? hi : 0.f;
                    }
                    bf16x2 pk = {(__bf16)lo, (__bf16)hi};
                    asm volatile("" : "+v"(pk));                                       // (the value is complete BEFORE the branch around its store: nothing waits inside it)
                    if (R.ok(2 * j, s)) *reinterpret_cast<bf16x2*>(Y + R.base + R.delta(2 * j, s)) = pk;
                }
            }
        }
        return;
    }
    if (EPI == 4) {
        // First maximum in the order 00, 01, 10, 11, as k_pool_fwd.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = n0 + 32 * t + (lane & 31);
            const float bb = bias_ready ? bias_ready[t] : bias[co];
            const unsigned pb = pool_base(co);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    // max-pool of relu(x + b) = relu(max(x) + b): the maximum (first of equals) is taken on the raw sums
                    const int i0 = 4 * gq + 2 * pp;
                    const float v[4] = {acc[t][i0], acc[t][i0 + 1], acc[t][8 + i0], acc[t][8 + i0 + 1]};
                    float best = v[0];
                    int bk = 0;
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (v[k] > best) { best = v[k]; bk = k; }
                    best += bb;
                    best = best > 0.f ? best : 0.f;
                    asm volatile("" : "+v"(best));
                    const bool ok = pool_ok(gq, pp);                            // the same for the 32 lanes of a half-wave
                    const unsigned o = pb + pool_delta(gq, pp);
                    if (ok) Y[o] = narrow<TY>(best);
                    store_idx_quad(pool_idx, o, bk, ok, lane);
                }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = n0 + 32 * t + (lane & 31);
            const float bb = (EPI == 1 || EPI == 2) ? (bias_ready ? bias_ready[t] : bias[co]) : 0.f;
            const EpiRows<TW> R(lane, wave, img0, oh0, ow0, s, co);
            // (all gate values first, and every value complete before the branch around its store: see the bf16 form above -- as written
            // before, each row's gate load waited behind the previous row's store, and the bias add sat inside the branch behind an
            // s_waitcnt vmcnt(0) that, after the first store, waited for that store)
            float gv[16];
            if (EPI == 3 && !gate) {
#pragma unroll
                for (int i = 0; i < 16; ++i) gv[i] = widen(reinterpret_cast<const TG*>(bias)[R.ok(i, s) ? R.base + R.delta(i, s) : 0u]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[t][i] + bb;
                if (EPI == 2) v = v > 0.f ? v : 0.f;
                if (EPI == 3) v = (gate ? gate[t][i] : gv[i]) > 0.f ? v : 0.f;
                asm volatile("" : "+v"(v));
                if (R.ok(i, s)) Y[R.base + R.delta(i, s)] = narrow<TY>(v);       // the host keeps tensors below 2^31 elements
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3x3_halo_f32
//
// LDS images.  Halo: [pixel][32 channels], 36-float rows (16-byte stores, and a lane's FOUR k-steps are ONE ds_read_b128: the
// contraction order inside a channel block is free, so lane (r, h) of the 32x32x2 MFMA takes channels 8j + 4h .. 8j + 4h + 3 of its
// pixel for k-steps 4j .. 4j + 3; the weights are read in the matching order).  Weights of one filter row: [kw * 32 + ci][BN],
// unpadded, 16-byte stores, bank-swizzled so that the two half-waves of a ds_read_b32 (weight rows 4 apart) hit different halves of
// the 64 banks: BN = 64 -> column ^ 32 on rows with bit 2 set; BN = 32 -> bits 0 and 2 of the row index swapped.
// Work items = (pixel block, BN-wide column block); a workgroup takes items blockIdx.x, + gridDim.x, ... (the host launches at most
// as many workgroups as the chip holds at once: a grid of 1024 one-item workgroups on 768 slots ran as two full-length rounds,
// measured), as ONE software pipeline of phases (item, channel block, filter row): the global loads of the next phase's weights --
// and halo, when the channel block or the item changes -- are issued before the 96 / 48 MFMAs of the current phase and stored to LDS
// after them, so an item's epilogue stores and the next item's first loads overlap too.
// Wave w computes output rows 2w, 2w + 1 of the block (32 pixels) x BN channels.  Epilogues as k_conv_fwd, plus EPI 4: bias +
// ReLU + the 2 x 2 max-pool that follows, fused (a lane's sixteen accumulator rows are four complete pooling windows), writing
// the pooled map and the arg-max image as k_pool_fwd would (the arg-max is taken on the raw sums: it can differ from k_pool_fwd's
// only in windows whose pooled value is zero, where the ReLU gate of every consumer ignores it).  PIN: the input is a gradient that exists only at pooled
// resolution (PooledGrad): a thread loads (dP, P, arg-max) of ONE pooled pixel's four channels and expands it into the up to four
// halo pixels of that window -- no k_pool_bwd, no full-size dZ, a quarter of the loads.
template <int TW, int BN, int EPI, bool PIN = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3))) void k_conv3x3_halo_f32(
    const float* __restrict__ X, const float* __restrict__ Wk, const float* __restrict__ bias, float* __restrict__ Y, ConvShape s, int tiles_w,
    int tiles_h, int n_items, uint8_t* __restrict__ pool_idx, PooledGrad pin) {
    using Gm = HaloGeom<TW>;
    constexpr int CB = 32, NT = BN / 32, LDC = 36;
    constexpr int PPW = TW / 2 + 2;                                   // PIN: pooled pixels under one image's halo: 6 rows x (TW/2 + 2)
    constexpr int GR = PIN ? 6 : Gm::HH, GC = PIN ? PPW : Gm::HWD;    // the grid that is loaded: pooled pixels, or the halo itself
    constexpr int NH = (Gm::NIMG * GR * GC * (CB / 4) + kThreads - 1) / kThreads;
    constexpr int NB = 3 * CB * (BN / 4) / kThreads;                  // f32x4 chunks per thread of one filter row's weights
    static_assert(3 * CB * (BN / 4) % kThreads == 0, "weight chunks divide over the threads");
    __shared__ __attribute__((aligned(16))) float Hs[Gm::LPIX * LDC];
    __shared__ __attribute__((aligned(16))) float Bs[3 * CB * BN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Cin = s.Cin, nblk = s.Cout / BN;
    const int GH = PIN ? s.H >> 1 : s.H, GW = PIN ? s.W >> 1 : s.W;   // image size of the loaded grid

    const int r = lane & 31, h = lane >> 5;
    // this lane's A row = output pixel r of the wave's 32: block row 2 wave + (r >> 4), block column r & 15 -> image (column / TW)
    const int py = 2 * wave + (r >> 4), pxb = r & 15;
    const float* arow = Hs + ((pxb / TW) * Gm::IS + py * Gm::HWD + pxb % TW) * LDC + 4 * h;
    const float* bt[NT];
    if (BN == 64) { bt[0] = Bs + 256 * h + r + 32 * h; bt[NT - 1] = Bs + 256 * h + r + 32 * (1 - h); }
    else bt[0] = Bs + 32 * h + r;

    // Work item = (column block nb fastest, block column tw, block row th, image group g): kept as a mixed-radix counter that is
    // advanced by the (decomposed) grid size -- scalar adds and compares per item instead of three integer divisions (which
    // compile to dozens of VALU instructions each, even for uniform operands).
    struct Item { int img0, oh0, ow0, n0; };
    struct Pos { int nb, tw, th, g; };
    auto split = [&](int item) { Pos p; p.nb = item % nblk; item /= nblk; p.tw = item % tiles_w; item /= tiles_w; p.th = item % tiles_h; p.g = item / tiles_h; return p; };
    const Pos stride = split((int)gridDim.x);
    auto advance = [&](Pos p) {
        p.nb += stride.nb; if (p.nb >= nblk) { p.nb -= nblk; ++p.tw; }
        p.tw += stride.tw; if (p.tw >= tiles_w) { p.tw -= tiles_w; ++p.th; }
        p.th += stride.th; if (p.th >= tiles_h) { p.th -= tiles_h; ++p.g; }
        p.g += stride.g;
        return p;
    };
    auto item_of = [&](const Pos& p) { return Item{p.g * Gm::NIMG, p.th * Gm::TH, p.tw * TW, p.nb * BN}; };

    // ---- staging: registers first (loads fly under the MFMAs), LDS after the barrier
    StageMap<NH> hm;
    stage_map_init<NH, GR, GC, Gm::NIMG>(hm, tid);
    f32x4 hv[NH], hp[PIN ? NH : 1];
    unsigned hi[PIN ? NH : 1];
    unsigned okm = 0;
    auto halo_load = [&](const Item& it, int cb) {
        // grid origin: halo pixel (oh0 - 1, ow0 - 1), or the pooled pixel under it
        const int y0 = PIN ? (it.oh0 >> 1) - 1 : it.oh0 - 1, x0 = PIN ? (it.ow0 >> 1) - 1 : it.ow0 - 1;
        const int base = ((it.img0 * GH + y0) * GW + x0) * Cin + cb;          // element offset of the origin (may be negative: masked below)
        okm = 0;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int pk = opaque(hm.pk[q]);
            const bool ok = stage_ok(pk, y0, x0, it.img0, GH, GW, s.N);
            okm |= (ok ? 1u : 0u) << q;
            const unsigned off = ok ? (unsigned)(base + stage_rel(pk, tid, GH, GW, Cin)) : 0u;
            if (PIN) {
                hv[q] = *reinterpret_cast<const f32x4*>(pin.dP + off);
                hp[q] = *reinterpret_cast<const f32x4*>(pin.P + off);
                hi[q] = *reinterpret_cast<const unsigned*>(pin.idx + off);
            } else {
                hv[q] = *reinterpret_cast<const f32x4*>(X + off);
            }
        }
    };
    auto halo_store = [&]() {
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const bool ok = (okm >> q) & 1u;
            const int pk = hm.pk[q];
            if (PIN) {
                if (pk >= 0) {
                    const int pr = pk & 255, pc = (pk >> 8) & 255, i = pk >> 16;
                    f32x4 v[4];
                    unpool4x4(ok ? hv[q] : f32x4{0, 0, 0, 0}, hp[q], hi[q], v);
                    float* w0 = &Hs[(i * Gm::IS + (2 * pr - 1) * Gm::HWD + 2 * pc - 1) * LDC + c4];      // window position 0 (may lie outside the halo)
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) {
                        // halo pixel of window position pos: row 2 pr - 1 + dy in [0, HH), column 2 pc - 1 + dx in [0, HWD)
                        const bool in = ((pos >> 1) ? pr < Gm::HH / 2 : pr > 0) && ((pos & 1) ? pc < Gm::HWD / 2 : pc > 0);
                        if (in) *reinterpret_cast<f32x4*>(w0 + ((pos >> 1) * Gm::HWD + (pos & 1)) * LDC) = v[pos];
                    }
                }
            } else if (pk >= 0) {
                *reinterpret_cast<f32x4*>(&Hs[Gm::lds_pix((tid >> 3) + 32 * q) * LDC + c4]) = ok ? hv[q] : f32x4{0, 0, 0, 0};
            }
        }
    };
    // weights of one filter row: chunk q of the thread is row (tid / (BN/4)) + (1024 / BN) q of the [kw * 32 + ci][BN] tile, i.e. a
    // thread offset plus offsets that are uniform over the workgroup; likewise in LDS (the swizzles only involve the thread's part)
    f32x4 bv[NB];
    constexpr int RQ = kThreads / (BN / 4);                           // rows per q step: 16 (BN = 64) / 32 (BN = 32)
    const int br0 = tid / (BN / 4), bc4 = (tid % (BN / 4)) * 4;
    const unsigned b_goff = (unsigned)(br0 * s.Cout + bc4);
    float* const b_lds = BN == 64 ? &Bs[br0 * 64 + (bc4 ^ (((br0 >> 2) & 1) << 5))] : &Bs[((br0 & ~5) | ((br0 & 1) << 2) | ((br0 >> 2) & 1)) * 32 + bc4];
    auto b_load = [&](const Item& it, int cb, int kh) {
        const float* wp = Wk + ((long long)(kh * 3) * Cin + cb) * s.Cout + it.n0;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int row = RQ * q;                                    // + br0: kw = row >> 5 and ci = row & 31 (+ br0 < RQ <= 32) are compile-time
            bv[q] = *reinterpret_cast<const f32x4*>(wp + ((long long)(row >> 5) * Cin + (row & 31)) * s.Cout + b_goff);
        }
    };
    auto b_store = [&]() {
#pragma unroll
        for (int q = 0; q < NB; ++q) *reinterpret_cast<f32x4*>(b_lds + RQ * q * BN) = bv[q];
    };

    const int nph = (Cin / CB) * 3;                                   // phases of one item: (channel block, filter row)
    int item = blockIdx.x;
    if (item >= n_items) return;
    RCNX_STAMP(0);
    RCNX_STAMP_HW();
    int stamp_slot = 1;
    Pos pos = split(item);
    Item cur = item_of(pos);
    halo_load(cur, 0);
    b_load(cur, 0, 0);
    bool first = true;
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
        const int nitem = item + gridDim.x;
        pos = advance(pos);
        const Item nxt = item_of(pos);
        int cb = 0, kh = 0;
#pragma unroll 1
        for (int ph = 0; ph < nph; ++ph) {
            if (!first) __syncthreads();                              // the previous phase's operands have been consumed
            first = false;
            if (kh == 0) halo_store();
            b_store();
            __syncthreads();
            RCNX_STAMP(stamp_slot); ++stamp_slot;                     // operands staged (slot 1: end of the prologue)
            const int nkh = kh == 2 ? 0 : kh + 1, ncb = kh == 2 ? cb + CB : cb;
            {
                // ONE load site for "this item's next phase" and "the next item's first phase" (as two sites the second one's registers were
                // copied over behind an s_waitcnt vmcnt(0), and phases waited for just-issued loads before their MFMAs: convnet_halo_bf16.hpp)
                const bool same = ph + 1 < nph;
                const Item li = same ? cur : nxt;
                const int lcb = same ? ncb : 0, lkh = same ? nkh : 0;
                if (same || nitem < n_items) {
                    b_load(li, lcb, lkh);
                    if (lkh == 0) halo_load(li, lcb);
                }
            }
            RCNX_STAMP(stamp_slot); ++stamp_slot;                     // the next phase's loads issued
            const float* ak = arow + kh * Gm::HWD * LDC;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ak + kw * LDC + 8 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float b = BN == 64 ? bt[t][(kw * 32 + 8 * j + i) * 64] : bt[0][(kw * 32 + 8 * j + 4 * (i & 1) + (i & 2)) * 32];
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            kh = nkh; cb = ncb;
        }
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // the item's MFMAs issued
        halo_epilogue<TW, NT, EPI>(acc, lane, wave, cur.img0, cur.oh0, cur.ow0, cur.n0, s, bias, Y, pool_idx);
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // epilogue stores issued
        cur = nxt;
    }
    RCNX_STAMP(29);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_wgrad3x3_halo_f32 -- dW of a 3x3 layer, LDS-tiled.
//
// k_conv_wgrad gives every 32-row k-block (one filter tap of 32 channels) its own staged X tile: the input is read nine times and
// dZ once per k-block.  Here workgroup (ci block, co block, chunk) stages, per block of 128 output pixels, the input halo of its 32
// input channels and the block's dZ for its 32 output channels ONCE; each of its four waves takes two of the block's eight pixel
// rows and contracts them into ALL NINE taps' 32 x 32 tiles (144 accumulator registers): per pair of pixels one dZ fragment is
// read and used by nine MFMAs.  The contraction index is the pixel = the row of both LDS images, so lane (r, h) reads element r of
// pixel p + h: 32-float rows put the two half-waves on different halves of the banks with no padding, and every offset except the
// lane's own is an immediate.  The four waves' tiles are summed in wave order through LDS at the end; the partial goes to
// slab[chunk][K + 1][Cout] exactly as k_conv_wgrad's (row K = bias partial from the ci block 0 workgroups), so k_reduce_all
// finishes either.  PDZ: dZ exists only at pooled resolution (PooledGrad): a block has 32 pooled pixels x 8 channel chunks = one
// chunk per thread, expanded into its window's four pixels while staging.
template <int TW, bool PDZ = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_wgrad3x3_halo_f32(
    const float* __restrict__ X, const float* __restrict__ dZ, float* __restrict__ slab, ConvShape s, int tiles_w, int tiles_h, int blocks_per_chunk,
    PooledGrad pdz) {
    using Gm = HaloGeom<TW>;
    constexpr int CB = 32, BN = 32, NPX = 128;
    constexpr int HCH = Gm::NPIX * (CB / 4), NH = (HCH + kThreads - 1) / kThreads;
    constexpr int ND = PDZ ? 1 : NPX * (BN / 4) / kThreads;           // 4 (full resolution) or 1 (pooled)
    constexpr int kHalo = Gm::LPIX * CB, kLds = kHalo + NPX * BN;
    static_assert(kLds >= 4 * 16 * 64, "the wave partials of one tap must fit the staging memory");
    __shared__ __attribute__((aligned(16))) float smem[kLds];
    float* Hs = smem;                                                 // [halo pixel][ci]
    float* Ds = smem + kHalo;                                         // [block pixel][co]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cb = blockIdx.x * CB, n0 = blockIdx.y * BN, chunk = blockIdx.z;
    const int Cin = s.Cin, K = 9 * Cin;
    const int total_blocks = tiles_w * tiles_h * ((s.N + Gm::NIMG - 1) / Gm::NIMG);
    const int b0 = chunk * blocks_per_chunk, b1 = b0 + blocks_per_chunk < total_blocks ? b0 + blocks_per_chunk : total_blocks;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};

    // this thread's chunks of the two staged grids, decoded once (StageMap): halo pixels of X, and the block's dZ pixels -- full
    // resolution [8][16] (column = block column: TW = 8 puts image 1 at columns 8..15), or, PDZ, its 4 x 8 pooled pixels
    // (144 accumulator registers leave no room for the table: it lives in LDS, whose reads do not share a counter with the global
    // loads as a scratch reload would)
    __shared__ int Pk[NH * kThreads];
    {
        StageMap<NH> hm;
        stage_map_init<NH, Gm::HH, Gm::HWD, Gm::NIMG>(hm, tid);
#pragma unroll
        for (int q = 0; q < NH; ++q) Pk[q * kThreads + tid] = hm.pk[q];          // read back by the same thread only: no barrier needed
    }
    int drel0;                                                     // dZ chunk q: block pixel (tid >> 3) + 32 q = (row (tid >> 7) + 2 q, block column (tid >> 3) & 15)
    const int dcolb = PDZ ? 2 * ((tid >> 3) & 7) : (tid >> 3) & 15;   // block column of the chunk's (window's left) pixel
    const int drow0 = PDZ ? tid >> 6 : tid >> 7;                      // its (pooled) row for q = 0
    const int DH = PDZ ? s.H >> 1 : s.H, DW = PDZ ? s.W >> 1 : s.W;
    const int dximg = dcolb / TW, dxcol = PDZ ? (dcolb % TW) >> 1 : dcolb % TW;
    drel0 = ((dximg * DH + drow0) * DW + dxcol) * s.Cout + (tid & 7) * 4;
    f32x4 hv[NH], dv[ND], dp[1];
    unsigned di = 0;
    unsigned okm = 0;
    auto gload = [&](int blk) {
        int q0 = blk;
        const int tw = q0 % tiles_w; q0 /= tiles_w;
        const int th = q0 % tiles_h;
        const int img0 = (q0 / tiles_h) * Gm::NIMG;
        const int oh0 = th * Gm::TH, ow0 = tw * TW;
        const int hbase = ((img0 * s.H + oh0 - 1) * s.W + ow0 - 1) * Cin + cb;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int pk = opaque(Pk[q * kThreads + tid]);            // opaque: or the offsets below are hoisted out of the block loop into registers
            const bool ok = stage_ok(pk, oh0 - 1, ow0 - 1, img0, s.H, s.W, s.N);
            okm |= (ok ? 1u : 0u) << q;
            hv[q] = *reinterpret_cast<const f32x4*>(X + (ok ? (unsigned)(hbase + stage_rel(pk, tid, s.H, s.W, Cin)) : 0u));
        }
        const int dy0 = PDZ ? oh0 >> 1 : oh0, dx0 = PDZ ? ow0 >> 1 : ow0;
        const int dbase = ((img0 * DH + dy0) * DW + dx0) * s.Cout + n0;
#pragma unroll
        for (int q = 0; q < ND; ++q) {
            const bool ok = img0 + dximg < s.N && dy0 + drow0 + 2 * q < DH && dx0 + dxcol < DW;
            okm |= (ok ? 1u : 0u) << (16 + q);
            const unsigned off = ok ? (unsigned)(dbase + drel0 + 2 * q * DW * s.Cout) : 0u;
            if (PDZ) {
                dv[q] = *reinterpret_cast<const f32x4*>(pdz.dP + off);
                dp[q] = *reinterpret_cast<const f32x4*>(pdz.P + off);
                di = *reinterpret_cast<const unsigned*>(pdz.idx + off);
            } else {
                dv[q] = *reinterpret_cast<const f32x4*>(dZ + off);
            }
        }
    };
    auto lstore = [&]() {
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < NH; ++q)
            if ((NH * kThreads == HCH) || (tid >> 3) + 32 * q < Gm::NPIX) *reinterpret_cast<f32x4*>(&Hs[Gm::lds_pix((tid >> 3) + 32 * q) * CB + c4]) = ((okm >> q) & 1u) ? hv[q] : f32x4{0, 0, 0, 0};
        if (PDZ) {
            const bool ok = (okm >> 16) & 1u;
            f32x4 v[4];
            unpool4x4(ok ? dv[0] : f32x4{0, 0, 0, 0}, dp[0], di, v);
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                *reinterpret_cast<f32x4*>(&Ds[((2 * drow0 + (pos >> 1)) * 16 + dcolb + (pos & 1)) * BN + c4]) = v[pos];
                colsum += v[pos];
            }
        } else {
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const f32x4 v = ((okm >> (16 + q)) & 1u) ? dv[q] : f32x4{0, 0, 0, 0};
                *reinterpret_cast<f32x4*>(&Ds[(tid + kThreads * q) * 4]) = v;
                colsum += v;                                          // the thread's four columns are the same for every q (256 % 8 == 0)
            }
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const float* al = Hs + 32 * h + r;
    const float* dl = Ds + 32 * h + r;
    if (b0 < b1) gload(b0);
#pragma unroll 1
    for (int blk = b0; blk < b1; ++blk) {
        if (blk != b0) __syncthreads();                               // the previous block's images have been consumed
        lstore();
        __syncthreads();
        if (blk + 1 < b1) gload(blk + 1);                             // the next block's loads fly under this block's MFMAs
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            // pixels (y, xb) and (y, xb + 1) of the block, y = 2 wave + (ks >> 3); lane half h takes xb + h
            const int xb = 2 * (ks & 7);
            const float b = dl[((ks >> 3) * 16 + xb) * 32 + wave * (2 * 16 * 32)];
            const int hp0 = (xb / TW) * Gm::IS + (xb % TW);           // halo pixel of (row 0, xb), tap (0, 0), before the row term
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float a = al[(hp0 + ((ks >> 3) + kh) * Gm::HWD + kw) * 32 + wave * (2 * Gm::HWD * 32)];
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kh * 3 + kw], 0, 0, 0);
                }
        }
    }
    // ---- the four waves' tiles, summed in wave order, tap by tap
    float* out = slab + (long long)chunk * (K + 1) * s.Cout;
    float* Red = smem;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) Red[(wave * 16 + i) * 64 + lane] = acc[tap][i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + kThreads * u, i = e >> 6, ln = e & 63;
            const float v = (Red[(0 * 16 + i) * 64 + ln] + Red[(1 * 16 + i) * 64 + ln]) + (Red[(2 * 16 + i) * 64 + ln] + Red[(3 * 16 + i) * 64 + ln]);
            out[(long long)(tap * Cin + cb + mfma32_row(ln, i)) * s.Cout + n0 + (ln & 31)] = v;
        }
    }
    if (blockIdx.x == 0) {                                            // bias row: thread's four columns are 4 (tid % 8); fixed-order sum
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Red[tid * 4]) = colsum;
        __syncthreads();
        if (tid < BN) {
            const int grp = tid >> 2, comp = tid & 3;
            float t = 0.f;
            for (int u = grp; u < kThreads; u += BN / 4) t += Red[u * 4 + comp];
            out[(long long)K * s.Cout + n0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// First layer (CIN = 1 or 3 input channels: the whole 3 x 3 x CIN patch is one k-block of 9 / 27 <= 32).
//
// The implicit-GEMM kernels gather this layer's A tile with one guarded scalar load per element (and the layer is pure traffic:
// 1.8 of the CIFAR step's 32 GFLOP, 67 MB of activations).  Here the input halo of a block of 128 output pixels is a few hundred
// floats in LDS ([pixel][CIN], rows of the image stay contiguous, so the loads are coalesced dwords), element k = (kh, kw, ci) of
// pixel p sits at p * CIN + koff(k) with koff a compile-time table, and the weights never leave registers.

template <int CIN, int HWD> __host__ __device__ constexpr int conv1_koff(int k) {
    const int kk = k < 9 * CIN ? k : 9 * CIN - 1;           // padding columns read a real element (their weight / their output row is dropped)
    const int tap = kk / CIN, ci = kk % CIN;
    return ((tap / 3) * HWD + tap % 3) * CIN + ci;
}

// k_conv1_fwd_f32: Y = relu(conv3x3(X) + b), EPI 2, or the same followed by the 2 x 2 max-pool, EPI 4 (as k_conv3x3_halo_f32).
// 14 (CIN = 3) / 5 (CIN = 1) MFMAs per wave and block; work items = (32-channel column block, pixel block), persistent workgroups,
// the next block's halo in flight under the MFMAs and the stores of this one.
// One thread's elements of a staged first-layer halo: element q is float (tid + 256 q) of the [pixel][CIN] image.  Decoded once.
template <int NL> struct Stage1Map {
    int pk[NL];               // halo row | column << 8 | image << 16; negative: past the end
    int rel[NL];              // ((image * H + row) * W + column) * CIN + channel
    int lds[NL];              // float index in the LDS image
};
template <int NL, int CIN, int TW>
__device__ inline void stage1_init(Stage1Map<NL>& m, int tid, int Himg, int Wimg) {
    using Gm = HaloGeom<TW>;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        const int e = tid + kThreads * q;
        const int pix = e / CIN, ci = e - pix * CIN;
        const int i = pix / Gm::IMG_PIX, pr = pix - i * Gm::IMG_PIX;
        const int hy = pr / Gm::HWD, hx = pr - hy * Gm::HWD;
        const bool valid = e < Gm::NPIX * CIN;
        m.pk[q] = valid ? (hy | (hx << 8) | (i << 16)) : -1;
        m.rel[q] = valid ? ((i * Himg + hy) * Wimg + hx) * CIN + ci : 0;
        m.lds[q] = valid ? Gm::lds_pix(pix) * CIN + ci : 0;
    }
}

template <int CIN, int TW, int EPI, typename TY = float>
__global__ __launch_bounds__(kThreads) void k_conv1_fwd_f32(const float* __restrict__ X, const float* __restrict__ Wk, const float* __restrict__ bias,
                                                            TY* __restrict__ Y, ConvShape s, int tiles_w, int tiles_h, int n_items,
                                                            uint8_t* __restrict__ pool_idx) {
    using Gm = HaloGeom<TW>;
    constexpr int K = 9 * CIN, KS2 = (K + 1) / 2;
    constexpr int HF = Gm::NPIX * CIN, NL = (HF + kThreads - 1) / kThreads;
    __shared__ float Hs[Gm::LPIX * CIN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int py = 2 * wave + (r >> 4), pxb = r & 15;
    const float* hb = Hs + ((pxb / TW) * Gm::IS + py * Gm::HWD + pxb % TW) * CIN;
    const int n_groups = (s.N + Gm::NIMG - 1) / Gm::NIMG;

    // work item = (block column tw fastest, block row th, image group g, column block nb): a mixed-radix counter advanced by the
    // decomposed grid size, as in k_conv3x3_halo_f32
    struct Item { int img0, oh0, ow0, n0; };
    struct Pos { int tw, th, g, nb; };
    auto split = [&](int item) { Pos p; p.tw = item % tiles_w; item /= tiles_w; p.th = item % tiles_h; item /= tiles_h; p.g = item % n_groups; p.nb = item / n_groups; return p; };
    const Pos stride = split((int)gridDim.x);
    auto advance = [&](Pos p) {
        p.tw += stride.tw; if (p.tw >= tiles_w) { p.tw -= tiles_w; ++p.th; }
        p.th += stride.th; if (p.th >= tiles_h) { p.th -= tiles_h; ++p.g; }
        p.g += stride.g; if (p.g >= n_groups) { p.g -= n_groups; ++p.nb; }
        p.nb += stride.nb;
        return p;
    };
    auto item_of = [&](const Pos& p) { return Item{p.g * Gm::NIMG, p.th * Gm::TH, p.tw * TW, p.nb * 32}; };

    Stage1Map<NL> hm;
    stage1_init<NL, CIN, TW>(hm, tid, s.H, s.W);
    float hv[NL];
    unsigned okm = 0;
    auto halo_load = [&](const Item& it) {
        const int base = ((it.img0 * s.H + it.oh0 - 1) * s.W + it.ow0 - 1) * CIN;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const bool ok = stage_ok(hm.pk[q], it.oh0 - 1, it.ow0 - 1, it.img0, s.H, s.W, s.N);
            okm |= (ok ? 1u : 0u) << q;
            hv[q] = X[ok ? (unsigned)(base + hm.rel[q]) : 0u];
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int q = 0; q < NL; ++q)
            if (hm.pk[q] >= 0) Hs[hm.lds[q]] = ((okm >> q) & 1u) ? hv[q] : 0.f;
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    RCNX_STAMP(0);
    RCNX_STAMP_HW();
    int stamp_slot = 1;
    Pos pos = split(item);
    Item cur = item_of(pos);
    halo_load(cur);
    float wreg[KS2];
    float bb = 0.f;                                                   // the lane's bias: loaded WITH the weights -- a load in the epilogue is waited
    int wn0 = -1;                                                     // for with vmcnt(0), i.e. together with the next block's halo just prefetched
    bool first = true;
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        if (cur.n0 != wn0) {                                          // this lane's B operands: W[k = 2 ks + h][n0 + r]
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const int k = 2 * ks + h;
                wreg[ks] = k < K ? Wk[(long long)k * s.Cout + cur.n0 + r] : 0.f;
            }
            bb = bias[cur.n0 + r];
            wn0 = cur.n0;
            // The weights are complete HERE, once per column block.  Left pending, the compiler's wait for them sat in front of every
            // MFMA of every item as "at most 13, 12, ... 0 loads outstanding" -- counted from the youngest load, which in the steady state
            // is the next block's halo just prefetched: every item waited for its own prefetch before its last MFMAs (ISA, round 4).
            __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0), expcnt and lgkmcnt untouched
        }
        const int nitem = item + gridDim.x;
        pos = advance(pos);
        const Item nxt = item_of(pos);
        if (!first) __syncthreads();                                  // the previous block's halo has been consumed
        first = false;
        halo_store();
        __syncthreads();
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // halo staged
        if (nitem < n_items) halo_load(nxt);
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // next loads issued
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const float a = hb[h ? conv1_koff<CIN, Gm::HWD>(2 * ks + 1) : conv1_koff<CIN, Gm::HWD>(2 * ks)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[ks], acc, 0, 0, 0);
        }
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // MFMAs issued
        const int img0 = cur.img0, oh0 = cur.oh0, ow0 = cur.ow0;
        {
            // the shared epilogue (same layout of the accumulator rows, one column tile): uniform row deltas, channel-pair stores for bf16
            const f32x16 a1[1] = {acc};
            const float bb1[1] = {bb};
            halo_epilogue<TW, 1, EPI, TY, TY>(a1, lane, wave, img0, oh0, ow0, cur.n0, s, bias, Y, pool_idx, nullptr, bb1);
        }
        RCNX_STAMP(stamp_slot); ++stamp_slot;                         // epilogue issued
        cur = nxt;
    }
    RCNX_STAMP(29);
}

// k_conv1_wgrad_f32: dW[k][co] = sum over pixels of patch element k times dZ[pixel][co], one 32 x 32 tile (rows k < 9 CIN real) per
// workgroup (co block, chunk of pixel blocks); the MFMA's row index is k -- lane r reads element koff(r) of pixel p + h, a lane
// constant plus an immediate -- and its contraction index the pixel, as in k_wgrad3x3_halo_f32.  Same slab layout.
template <int CIN, int TW, bool PDZ, typename TD = float>       // TD: storage type of dZ / of the pooled gradient and gate
__global__ __launch_bounds__(kThreads) void k_conv1_wgrad_f32(const float* __restrict__ X, const TD* __restrict__ dZ, float* __restrict__ slab, ConvShape s,
                                                              int tiles_w, int tiles_h, int blocks_per_chunk, PooledGradT<TD> pdz) {
    using Gm = HaloGeom<TW>;
    constexpr int K = 9 * CIN, BN = 32, NPX = 128;
    constexpr int HF = Gm::NPIX * CIN, NL = (HF + kThreads - 1) / kThreads;
    constexpr int ND = PDZ ? 1 : NPX * (BN / 4) / kThreads;
    constexpr int kHalo = (Gm::LPIX * CIN + 2 * CIN + 3) / 4 * 4;     // + one pixel pair of slack for the padding rows' reads, 16-byte aligned
    __shared__ __attribute__((aligned(16))) float smem[kHalo + NPX * BN];
    float* Hs = smem;
    float* Ds = smem + kHalo;
    static_assert(NPX * BN >= 4 * 16 * 64, "the wave partials fit the dZ image");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blockIdx.x * BN, chunk = blockIdx.y;
    const int total_blocks = tiles_w * tiles_h * ((s.N + Gm::NIMG - 1) / Gm::NIMG);
    const int b0 = chunk * blocks_per_chunk, b1 = b0 + blocks_per_chunk < total_blocks ? b0 + blocks_per_chunk : total_blocks;

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    Stage1Map<NL> hm;
    stage1_init<NL, CIN, TW>(hm, tid, s.H, s.W);
    // dZ chunk q of the thread: block pixel (tid >> 3) + 32 q = (row (tid >> 7) + 2 q, block column (tid >> 3) & 15); PDZ: pooled pixel
    const int dcolb = PDZ ? 2 * ((tid >> 3) & 7) : (tid >> 3) & 15;
    const int drow0 = PDZ ? tid >> 6 : tid >> 7;
    const int DH = PDZ ? s.H >> 1 : s.H, DW = PDZ ? s.W >> 1 : s.W;
    const int dximg = dcolb / TW, dxcol = PDZ ? (dcolb % TW) >> 1 : dcolb % TW;
    const int drel0 = ((dximg * DH + drow0) * DW + dxcol) * s.Cout + (tid & 7) * 4;
    float hv[NL];
    chunk4_t<TD> dv[ND], dp[1];
    unsigned di = 0, okm = 0;
    auto gload = [&](int blk) {
        int q0 = blk;
        const int tw = q0 % tiles_w; q0 /= tiles_w;
        const int th = q0 % tiles_h;
        const int img0 = (q0 / tiles_h) * Gm::NIMG;
        const int oh0 = th * Gm::TH, ow0 = tw * TW;
        const int hbase = ((img0 * s.H + oh0 - 1) * s.W + ow0 - 1) * CIN;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const bool ok = stage_ok(hm.pk[q], oh0 - 1, ow0 - 1, img0, s.H, s.W, s.N);
            okm |= (ok ? 1u : 0u) << q;
            hv[q] = X[ok ? (unsigned)(hbase + hm.rel[q]) : 0u];
        }
        const int dy0 = PDZ ? oh0 >> 1 : oh0, dx0 = PDZ ? ow0 >> 1 : ow0;
        const int dbase = ((img0 * DH + dy0) * DW + dx0) * s.Cout + n0;
#pragma unroll
        for (int q = 0; q < ND; ++q) {
            const bool ok = img0 + dximg < s.N && dy0 + drow0 + 2 * q < DH && dx0 + dxcol < DW;
            okm |= (ok ? 1u : 0u) << (16 + q);
            const unsigned off = ok ? (unsigned)(dbase + drel0 + 2 * q * DW * s.Cout) : 0u;
            if (PDZ) {
                dv[q] = *reinterpret_cast<const chunk4_t<TD>*>(pdz.dP + off);
                dp[q] = *reinterpret_cast<const chunk4_t<TD>*>(pdz.P + off);
                di = *reinterpret_cast<const unsigned*>(pdz.idx + off);
            } else {
                dv[q] = *reinterpret_cast<const chunk4_t<TD>*>(dZ + off);
            }
        }
    };
    auto lstore = [&]() {
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < NL; ++q)
            if (hm.pk[q] >= 0) Hs[hm.lds[q]] = ((okm >> q) & 1u) ? hv[q] : 0.f;
        if (PDZ) {
            const bool ok = (okm >> 16) & 1u;
            f32x4 v[4];
            unpool4x4(ok ? widen4(dv[0]) : f32x4{0, 0, 0, 0}, widen4(dp[0]), di, v);
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                *reinterpret_cast<f32x4*>(&Ds[((2 * drow0 + (pos >> 1)) * 16 + dcolb + (pos & 1)) * BN + c4]) = v[pos];
                colsum += v[pos];
            }
        } else {
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const f32x4 v = ((okm >> (16 + q)) & 1u) ? widen4(dv[q]) : f32x4{0, 0, 0, 0};
                *reinterpret_cast<f32x4*>(&Ds[(tid + kThreads * q) * 4]) = v;
                colsum += v;
            }
        }
    };
    if (tid < kHalo - Gm::LPIX * CIN) Hs[Gm::LPIX * CIN + tid] = 0.f;    // the slack is read (by rows that are dropped) but never staged
    float* out = slab + (long long)chunk * (K + 1) * s.Cout;
    float* Red = Ds;
    if constexpr (CIN == 1) {
        // One input channel: the patch has 9 entries, so the 32-row MFMA would carry 23 empty rows.  v_mfma_f32_16x16x4_f32 instead: rows =
        // patch entries (9 of 16), columns = 16 output channels (two MFMAs for the 32), contraction = FOUR pixels per instruction -- half
        // the matrix-pipe time per block (the MNIST-shape first layer's weight gradient at B = 4096 took 103 us on the 32-row form).
        using f32x4v = __attribute__((ext_vector_type(4))) float;
        const int r16 = lane & 15, kq = lane >> 4;
        int ko = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) ko = r16 == k ? conv1_koff<1, Gm::HWD>(k) : ko;
        const float* al = Hs + ko + kq;                               // patch entry r16 of pixel p + kq
        const float* dl = Ds + 32 * kq + r16;                         // dZ[pixel p + kq][16 t + r16]
        f32x4v acc4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (b0 < b1) gload(b0);
#pragma unroll 1
        for (int blk = b0; blk < b1; ++blk) {
            if (blk != b0) __syncthreads();
            lstore();
            __syncthreads();
            if (blk + 1 < b1) gload(blk + 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {                          // the wave's 32 pixels, four at a time: row 2 wave + (ks >> 2), block columns 4 (ks & 3) ..
                const int xb = 4 * (ks & 3);
                const float a = al[(xb / TW) * Gm::IS + (xb % TW) + (ks >> 2) * Gm::HWD + wave * (2 * Gm::HWD)];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dl[((ks >> 2) * 16 + xb) * 32 + wave * (2 * 16 * 32) + 16 * t], acc4[t], 0, 0, 0);
            }
        }
        // D layout of the 16x16x4 MFMA: lane l holds column l & 15, rows 4 (l >> 4) + i, i = 0..3
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) Red[((wave * 2 + t) * 4 + i) * 64 + lane] = acc4[t][i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + kThreads * u, ti = e >> 6, ln = e & 63;              // ti = t * 4 + i
            const int k = 4 * (ln >> 4) + (ti & 3), co = 16 * (ti >> 2) + (ln & 15);
            const float v = (Red[(0 * 8 + ti) * 64 + ln] + Red[(1 * 8 + ti) * 64 + ln]) + (Red[(2 * 8 + ti) * 64 + ln] + Red[(3 * 8 + ti) * 64 + ln]);
            if (k < K) out[(long long)k * s.Cout + n0 + co] = v;
        }
    } else {
    const int r = lane & 31, h = lane >> 5;
    // lane's A element: patch entry r (clamped for the padding rows 9 CIN .. 31, which are not written back) of pixel p + h
    int ko = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) ko = r == k ? conv1_koff<CIN, Gm::HWD>(k) : ko;
    const float* al = Hs + ko + h * CIN;
    const float* dl = Ds + 32 * h + r;
    if (b0 < b1) gload(b0);
#pragma unroll 1
    for (int blk = b0; blk < b1; ++blk) {
        if (blk != b0) __syncthreads();
        lstore();
        __syncthreads();
        if (blk + 1 < b1) gload(blk + 1);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int xb = 2 * (ks & 7);
            const float b = dl[((ks >> 3) * 16 + xb) * 32 + wave * (2 * 16 * 32)];
            const float a = al[((xb / TW) * Gm::IS + (xb % TW) + (ks >> 3) * Gm::HWD) * CIN + wave * (2 * Gm::HWD * CIN)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) Red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + kThreads * u, i = e >> 6, ln = e & 63;
        const int k = mfma32_row(ln, i);
        const float v = (Red[(0 * 16 + i) * 64 + ln] + Red[(1 * 16 + i) * 64 + ln]) + (Red[(2 * 16 + i) * 64 + ln] + Red[(3 * 16 + i) * 64 + ln]);
        if (k < K) out[(long long)k * s.Cout + n0 + (ln & 31)] = v;
    }
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(&Red[tid * 4]) = colsum;
    __syncthreads();
    if (tid < BN) {
        const int grp = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int u = grp; u < kThreads; u += BN / 4) t += Red[u * 4 + comp];
        out[(long long)K * s.Cout + n0 + tid] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_head_f32 -- the classifier head in one launch: logits = H W + b, softmax + cross-entropy, d logits, the gradient into the hidden
// layer (gated by its ReLU) and the per-workgroup partial of the logits layer's weight gradient.
//
// With a few hundred samples these are five tiny GEMMs and a reduction -- 512 x 256 x 32, 512 x 32 x 256, 256 x 512 x 32 on the
// CIFAR net -- each of which ran as its own launch on a handful of workgroups (forward 9.6 + split-K epilogue 5.4, loss 5.5, input
// gradient 14.8, weight gradient 10.6 us: latency, not arithmetic).  Here a workgroup owns 32 samples: their hidden activations,
// the weights in both orientations (the flipped copy is kept current by the update kernels) and d logits stay in LDS between the
// steps.  F = hidden width (multiple of 32, <= 256: LDS), classes <= 32 (one padded column block).  The weight-gradient partials go
// to slab[workgroup][F + 1][32] like k_conv_wgrad's, so k_reduce_all finishes; the loss like k_softmax_ce (last workgroup
// adds the partials in order).
template <bool GATE>
__global__ __launch_bounds__(kThreads) void k_head_f32(const float* __restrict__ Hin, const float* __restrict__ Wk, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, const int* __restrict__ labels, int B, int F, int C,
                                                       float* __restrict__ logits, float* __restrict__ dH, float* __restrict__ slab, float* loss_part,
                                                       unsigned* counter, float inv_b, float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float head_smem[];
    const int LDH = F + 1, LDT = F + 32;
    float* Hs = head_smem;                       // [32 samples][F + 1]: odd rows, read along samples (logits) and along features (weight gradient)
    float* Ws = Hs + 32 * LDH;                   // [F][32]
    float* WTs = Ws + F * 32;                    // [32 classes][F + 32]: the two half-waves' rows on different halves of the banks
    float* Ls = WTs + 32 * LDT;                  // [4 waves][32][32] partial logits
    float* Ds = Ls + 4 * 1024;                   // [32 samples][32]: d logits, B operand of the weight gradient
    float* DsA = Ds + 1024;                      // [32][33]: the same, A operand of the input gradient
    __shared__ float red[kThreads];
    __shared__ int last;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int s0 = blockIdx.x * 32, F4 = F >> 2;
    for (int e = tid; e < 32 * F4; e += kThreads) {
        const int row = e / F4, c4 = (e - row * F4) * 4;
        const f32x4 v = s0 + row < B ? *reinterpret_cast<const f32x4*>(Hin + (long long)(s0 + row) * F + c4) : f32x4{0, 0, 0, 0};
        float* d = &Hs[row * LDH + c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        *reinterpret_cast<f32x4*>(&WTs[row * LDT + c4]) = *reinterpret_cast<const f32x4*>(Wt + (long long)row * F + c4);
    }
    for (int e = tid; e < F * 8; e += kThreads) *reinterpret_cast<f32x4*>(&Ws[e * 4]) = *reinterpret_cast<const f32x4*>(Wk + e * 4);
    __syncthreads();
    // ---- logits: wave w contracts features [w F/4, (w + 1) F/4)
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int k0 = wave * (F >> 2);
        for (int ks = 0; ks < (F >> 3); ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hs[r * LDH + k0 + 2 * ks + h], Ws[(k0 + 2 * ks + h) * 32 + r], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) Ls[wave * 1024 + mfma32_row(lane, i) * 32 + r] = acc[i];
    }
    __syncthreads();
    // ---- softmax + cross-entropy: 8 lanes per sample, 4 classes each
    {
        const int sm = tid >> 3, cg = (tid & 7) * 4, s = s0 + sm;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((Ls[sm * 32 + cg + j] + Ls[1024 + sm * 32 + cg + j]) + (Ls[2048 + sm * 32 + cg + j] + Ls[3072 + sm * 32 + cg + j])) + bias[cg + j];
        if (logits && s < B) *reinterpret_cast<f32x4*>(logits + (long long)s * 32 + cg) = f32x4{v[0], v[1], v[2], v[3]};
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = (cg + j < C && v[j] > mx) ? v[j] : mx;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) { const float o = __shfl_xor(mx, off, 8); mx = o > mx ? o : mx; }
        float ex[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ex[j] = cg + j < C ? expf(v[j] - mx) : 0.f; sum += ex[j]; }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 8);
        const int y = s < B ? labels[s] : 0;
        float zy = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) zy += cg + j == y ? v[j] : 0.f;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) zy += __shfl_xor(zy, off, 8);
        if ((tid & 7) == 0) red[sm] = s < B ? -(zy - mx - logf(sum)) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = (s < B && cg + j < C) ? (ex[j] / sum - (cg + j == y ? 1.f : 0.f)) * inv_b : 0.f;
            Ds[sm * 32 + cg + j] = d;
            DsA[sm * 33 + cg + j] = d;
        }
    }
    __syncthreads();
    // ---- gradient into the hidden layer: dH[32][F] = D[32][32] W^T, feature blocks of 32 over the waves
    if (dH) {
        for (int blk = wave; blk < (F >> 5); blk += 4) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(DsA[r * 33 + 2 * ks + h], WTs[(2 * ks + h) * LDT + blk * 32 + r], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int sr = mfma32_row(lane, i);
                if (s0 + sr < B) {
                    float v = acc[i];
                    if (GATE) v = Hs[sr * LDH + blk * 32 + r] > 0.f ? v : 0.f;
                    dH[(long long)(s0 + sr) * F + blk * 32 + r] = v;
                }
            }
        }
    }
    // ---- weight-gradient partial: dW[F][32] = H^T D over this workgroup's samples; bias row = column sums of D
    float* out = slab + (long long)blockIdx.x * (F + 1) * 32;
    for (int blk = wave; blk < (F >> 5); blk += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Hs[(2 * ks + h) * LDH + blk * 32 + r], Ds[(2 * ks + h) * 32 + r], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) out[(long long)(blk * 32 + mfma32_row(lane, i)) * 32 + r] = acc[i];
    }
    if (tid < 32) {
        float t = 0.f;
        for (int sm = 0; sm < 32; ++sm) t += Ds[sm * 32 + tid];
        out[(long long)F * 32 + tid] = t;
    }
    // ---- loss: this workgroup's samples in order, then (last workgroup) the workgroups in order
    if (tid == 0) {
        float t = 0.f;
        for (int g = 0; g < 32; ++g) t += red[g];
        __hip_atomic_store(&loss_part[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    float t = 0.f;
    if (loss_out)
        for (int b = tid; b < (int)gridDim.x; b += kThreads) t += __hip_atomic_load(&loss_part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    red[tid] = t;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < kThreads; ++i) tot += red[i];
        if (loss_out) *loss_out = tot * inv_b;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace rcnx
