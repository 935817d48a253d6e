// convnet_halo_bf16.hpp -- Track X, bf16 MFMA path: the LDS-tiled 3x3 forward / input-gradient kernel as a software pipeline
// (round 3).  No reference counterpart.
//
// k_conv3x3_halo_bf16 (convnet_bf16.hpp) has the right data flow -- one staged halo per 8 x 16 pixel block serves all nine taps --
// but loads, waits, converts and stores every operand in place, with two barriers per filter row and one work item per workgroup:
// on the 224 x 224 net it ran at 15 % of the bf16 MFMA rate AND 4.7 x its HBM floor, i.e. on latency and integer instructions.
// k_conv3x3_halo_bf16p is the same kernel on the frame round 3 built for the fp32 path (convnet_halo.hpp): work items walked by a
// resident grid, phases (item, channel block, filter row) whose operands are loaded into registers one phase ahead of their use --
// across item boundaries too --, every thread's share of the staged grids decoded once (StageMap), the epilogues shared.  Same LDS
// images, same operand rounding (RNE to bf16 on the way into LDS, fp32 accumulation), same MFMA order as k_conv3x3_halo_bf16:
// bit-identical results.
#pragma once

#include "convnet_bf16.hpp"
#include "convnet_halo.hpp"

#ifndef RCNX_ABL
#define RCNX_ABL 0
#endif

namespace rcnx {

// TS: storage type of X, Y, EPI 3's gate tensor (passed through `bias`) and the pooled-resolution input (convnet.hpp, Chunk4): with
// __bf16 the halo chunks go from global memory to LDS as they are -- no conversion, half the bytes.
__device__ inline bf16x4 as_bf16x4(const f32x4& v) { return to_bf16x4(v); }
__device__ inline bf16x4 as_bf16x4(const bf16x4& v) { return v; }

// HaloGeom<16> with MG x 8 rows of output pixels
template <int MG> struct HaloRows { static constexpr int TH = 8 * MG, HH = TH + 2, HWD = 16 + 2, NPIX = HH * HWD; };

// MG: pixel-block rows per item in units of eight (1: 8 x 16 pixels, 2: 16 x 16).  With MG = 2 a wave computes TWO 32-pixel row groups
// against the same weights: the weights of a filter row are staged (L2 -> registers -> LDS) once per 256 pixels instead of once per 128,
// every B fragment read from LDS feeds two MFMAs, and the two barriers of a phase are paid per 48 MFMAs instead of 24 -- for 64 more
// accumulator registers (two workgroups per CU instead of three).
template <int CB, int BN, int EPI, bool PIN = false, typename TS = float, int MG = 1>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu((CB == 64 && BN == 64) || MG == 2 ? 2 : 3))) void k_conv3x3_halo_bf16p(
    const TS* __restrict__ X, const __bf16* __restrict__ WB, const float* __restrict__ bias, TS* __restrict__ Y, ConvShape s, int tiles_w,
    int tiles_h, int n_items, uint8_t* __restrict__ pool_idx, PooledGradT<TS> pin) {
    static_assert(CB % 16 == 0 && BN % 32 == 0, "channel blocks of the 32x32x16 MFMA");
    using Gm = HaloRows<MG>;
    constexpr int TW = 16, NT = BN / 32, LDC = CB + 8;                // halves per pixel / per weight row in LDS (16-byte aligned, bank-skewed)
    constexpr int CPP = CB / 4;                                       // f32x4 chunks per pixel of a channel block
    constexpr int PPW = TW / 2 + 2;                                   // PIN: pooled pixels under the halo: 6 rows x 10
    constexpr int GR = PIN ? Gm::TH / 2 + 2 : Gm::HH, GC = PIN ? PPW : Gm::HWD;    // the grid that is loaded: pooled pixels, or the halo itself
    constexpr int NH = (GR * GC * CPP + kThreads - 1) / kThreads;
    constexpr int BCH = 3 * BN * (CB / 8);                            // 16-byte chunks of one filter row's weights for a channel block
    constexpr int NB = (BCH + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) __bf16 Hs[Gm::NPIX * LDC];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3 * BN * LDC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Cin = s.Cin, K = 9 * Cin, nblk = s.Cout / BN;
    const int GH = PIN ? s.H >> 1 : s.H, GW = PIN ? s.W >> 1 : s.W;   // image size of the loaded grid
    const int r = lane & 31, h = lane >> 5;
    const int py = 2 * wave + (r >> 4), px = r & 15;                  // this lane's A row = output pixel (py, px) of the block

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
    auto item_of = [&](const Pos& p) { return Item{p.g, p.th * Gm::TH, p.tw * TW, p.nb * BN}; };

    // ---- staging: registers first (loads fly under the MFMAs), LDS after the barrier
    StageMap<NH> hm;
    stage_map_init<NH, GR, GC, 1, CPP>(hm, tid);
    chunk4_t<TS> hv[NH], hp[PIN ? NH : 1];
    unsigned hi[PIN ? NH : 1];
    unsigned okm = 0;
    auto halo_load = [&](const Item& it, int cb) {
        const int y0 = PIN ? (it.oh0 >> 1) - 1 : it.oh0 - 1, x0 = PIN ? (it.ow0 >> 1) - 1 : it.ow0 - 1;
        const int base = ((it.img0 * GH + y0) * GW + x0) * Cin + cb;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int pk = opaque(hm.pk[q]);
            const bool ok = stage_ok(pk, y0, x0, it.img0, GH, GW, s.N);
            okm |= (ok ? 1u : 0u) << q;
            const unsigned off = ok ? (unsigned)(base + stage_rel<CPP>(pk, tid, GH, GW, Cin)) : 0u;
            if (PIN) {
                hv[q] = *reinterpret_cast<const chunk4_t<TS>*>(pin.dP + off);
                hp[q] = *reinterpret_cast<const chunk4_t<TS>*>(pin.P + off);
                hi[q] = *reinterpret_cast<const unsigned*>(pin.idx + off);
            } else {
                hv[q] = *reinterpret_cast<const chunk4_t<TS>*>(X + off);
            }
        }
    };
    auto halo_store = [&]() {
        const int c4 = (tid % CPP) * 4;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const bool ok = (okm >> q) & 1u;
            const int pk = hm.pk[q];
            if (pk < 0) continue;
            if (PIN) {
                const int pr = pk & 255, pc = (pk >> 8) & 255;
                f32x4 v[4];
                unpool4x4(ok ? widen4(hv[q]) : f32x4{0, 0, 0, 0}, widen4(hp[q]), hi[q], v);
                __bf16* w0 = &Hs[((2 * pr - 1) * Gm::HWD + 2 * pc - 1) * LDC + c4];              // window position 0 (may lie outside the halo)
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const bool in = ((pos >> 1) ? pr < Gm::HH / 2 : pr > 0) && ((pos & 1) ? pc < Gm::HWD / 2 : pc > 0);
                    if (in) *reinterpret_cast<bf16x4*>(w0 + ((pos >> 1) * Gm::HWD + (pos & 1)) * LDC) = to_bf16x4(v[pos]);
                }
            } else {
                *reinterpret_cast<bf16x4*>(&Hs[(tid / CPP + (kThreads / CPP) * q) * LDC + c4]) = ok ? as_bf16x4(hv[q]) : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            }
        }
    };
    // weights of one filter row for a channel block: chunk q of the thread is 8 halves of row (kw * BN + co) of the [3 BN][CB] tile,
    // straight out of the transposed bf16 copy WB[Cout][K].  Row of chunk q = r0 + RQ q with r0 = tid / (CB/8) < RQ and RQ a multiple
    // or a divisor of BN, so (kw, co) of every q is the thread's own (kw0, co0) plus compile-time steps: one register for the global
    // offset, one for the LDS address.
    bf16x8 bv[NB];
    constexpr int RQ = kThreads / (CB / 8);
    static_assert(RQ % BN == 0 || BN % RQ == 0, "rows per step and column block must divide one another");
    const int br0 = tid / (CB / 8), bc8 = (tid % (CB / 8)) * 8;
    const int b_goff = (br0 % BN) * K + (br0 / BN) * Cin + bc8;       // (co0, kw0) of the thread
    const __bf16* const b_lds = &Bs[br0 * LDC + bc8];
    auto b_valid = [&](int q) { return NB * kThreads == BCH || tid + kThreads * q < BCH; };
    auto b_load = [&](const Item& it, int cb, int kh) {
        const __bf16* wp = WB + (long long)it.n0 * K + kh * 3 * Cin + cb + b_goff;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            // row step RQ q: RQ >= BN: kw += (RQ / BN) q; RQ < BN: co += RQ (q mod BN/RQ), kw += q / (BN/RQ)
            const int dkw = RQ >= BN ? (RQ / BN) * q : q / (BN / RQ), dco = RQ >= BN ? 0 : RQ * (q % (BN / RQ));
            if (b_valid(q)) bv[q] = *reinterpret_cast<const bf16x8*>(wp + (long long)dco * K + dkw * Cin);
        }
    };
    auto b_store = [&]() {
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (b_valid(q)) *reinterpret_cast<bf16x8*>(const_cast<__bf16*>(b_lds) + RQ * q * LDC) = bv[q];
    };

    const int nph = (Cin / CB) * 3;                                   // phases of one item: (channel block, filter row)
    int item = blockIdx.x;
    if (item >= n_items) return;
    Pos pos = split(item);
    Item cur = item_of(pos);
    halo_load(cur, 0);
    b_load(cur, 0, 0);
    bool first = true;
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        f32x16 acc[MG][NT];
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[g][t][q] = 0.f;
        const int nitem = item + gridDim.x;
        pos = advance(pos);
        const Item nxt = item_of(pos);
        int cb = 0, kh = 0;
#pragma unroll 1
        for (int ph = 0; ph < nph; ++ph) {
            // RCNX_ABL (diagnostic builds only, tools/ablate_halo_bf16.sh; never defined in the library): what a phase costs without its
            // global loads (1), without its LDS stores too (2), without its barriers too (3); 4: everything but the MFMAs
            const bool abl_first = first;
            if (!first && RCNX_ABL != 3) __syncthreads();             // the previous phase's operands have been consumed
            first = false;
            if (RCNX_ABL < 2 || RCNX_ABL >= 4 || abl_first) {
                if (kh == 0) halo_store();
                b_store();
            }
            if (RCNX_ABL != 3) __syncthreads();
            const int nkh = kh == 2 ? 0 : kh + 1, ncb = kh == 2 ? cb + CB : cb;
            if (RCNX_ABL == 0 || RCNX_ABL >= 4) {
                // ONE load site for "this item's next phase" and "the next item's first phase": as two sites the compiler loaded the
                // second one into other registers, copied them over behind an s_waitcnt vmcnt(0), and -- those registers doubling as
                // LDS-read destinations -- made every phase wait for most of its just-issued loads BEFORE its MFMAs (ISA, round 4)
                const bool same = ph + 1 < nph;
                const Item li = same ? cur : nxt;
                const int lcb = same ? ncb : 0, lkh = same ? nkh : 0;
                if (same || nitem < n_items) {
                    b_load(li, lcb, lkh);
                    if (lkh == 0) halo_load(li, lcb);
                }
            }
            if (RCNX_ABL != 4)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const __bf16* a = &Hs[((py + kh) * Gm::HWD + px + kw) * LDC + 8 * h];
                const __bf16* b = &Bs[(kw * BN + r) * LDC + 8 * h];
#pragma unroll
                for (int ks = 0; ks < CB / 16; ++ks) {
                    bf16x8 af[MG];
#pragma unroll
                    for (int g = 0; g < MG; ++g) {
                        if (RCNX_ABL == 5) { asm volatile("" : "=v"(af[g])); continue; }                 // 5: the MFMAs without their LDS reads
                        af[g] = *reinterpret_cast<const bf16x8*>(a + g * 8 * Gm::HWD * LDC + 16 * ks);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        bf16x8 bf;
                        if (RCNX_ABL == 5) asm volatile("" : "=v"(bf));
                        else bf = *reinterpret_cast<const bf16x8*>(b + 32 * t * LDC + 16 * ks);
                        if (RCNX_ABL == 6) {                                                             // 6: the LDS reads without the MFMAs
                            asm volatile("" :: "v"(bf), "v"(af[0]));
                            continue;
                        }
#pragma unroll
                        for (int g = 0; g < MG; ++g) acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], bf, acc[g][t], 0, 0, 0);
                    }
                }
            }
            kh = nkh; cb = ncb;
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) halo_epilogue<TW, NT, EPI, TS, TS>(acc[g], lane, wave, cur.img0, cur.oh0 + 8 * g, cur.ow0, cur.n0, s, bias, Y, pool_idx);
        cur = nxt;
    }
}

// k_conv3x3_halo_bf16_1cb -- the same for layers of exactly 32 input channels (one channel block).  A phase of the kernel above is then
// 6 / 12 MFMAs between two barriers and a restaged weight row -- on the 224 x 224 net's 32-channel layers (6.4 M pixels each) it ran
// at 2.2 x its HBM floor.  Here ALL nine taps' weights of a column block (9 x BN x 32 halves: 23 / 46 KB) are staged once and stay
// while the workgroup walks its items (the column block only changes when the layer has more than BN output channels); an item is its
// halo (prefetched under the previous item's MFMAs), 18 / 36 MFMAs, its epilogue: two barriers per item.
template <int BN, int EPI, bool PIN = false, typename TS = float>
__global__ __launch_bounds__(kThreads) void k_conv3x3_halo_bf16_1cb(const TS* __restrict__ X, const __bf16* __restrict__ WB, const float* __restrict__ bias,
                                                                   TS* __restrict__ Y, ConvShape s, int tiles_w, int tiles_h, int n_items,
                                                                   uint8_t* __restrict__ pool_idx, PooledGradT<TS> pin) {
    using Gm = HaloGeom<16>;
    constexpr int CB = 32, TW = 16, NT = BN / 32, LDC = CB + 8, CPP = CB / 4, K = 9 * CB;
    constexpr int PPW = TW / 2 + 2;
    constexpr int GR = PIN ? 6 : Gm::HH, GC = PIN ? PPW : Gm::HWD;
    constexpr int NH = (GR * GC * CPP + kThreads - 1) / kThreads;
    constexpr int BCH = BN * (K / 8), NB = (BCH + kThreads - 1) / kThreads;          // 16-byte chunks of the column block's weights: 36 per output channel
    __shared__ __attribute__((aligned(16))) __bf16 Hs[Gm::NPIX * LDC];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[9 * BN * LDC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nblk = s.Cout / BN;
    const int GH = PIN ? s.H >> 1 : s.H, GW = PIN ? s.W >> 1 : s.W;
    const int r = lane & 31, h = lane >> 5;
    const int py = 2 * wave + (r >> 4), px = r & 15;

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
    auto item_of = [&](const Pos& p) { return Item{p.g, p.th * Gm::TH, p.tw * TW, p.nb * BN}; };

    StageMap<NH> hm;
    stage_map_init<NH, GR, GC, 1, CPP>(hm, tid);
    chunk4_t<TS> hv[NH], hp[PIN ? NH : 1];
    unsigned hi[PIN ? NH : 1];
    unsigned okm = 0;
    auto halo_load = [&](const Item& it) {
        const int y0 = PIN ? (it.oh0 >> 1) - 1 : it.oh0 - 1, x0 = PIN ? (it.ow0 >> 1) - 1 : it.ow0 - 1;
        const int base = ((it.img0 * GH + y0) * GW + x0) * CB;
        okm = 0;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const int pk = hm.pk[q];
            const bool ok = stage_ok(pk, y0, x0, it.img0, GH, GW, s.N);
            okm |= (ok ? 1u : 0u) << q;
            const unsigned off = ok ? (unsigned)(base + stage_rel<CPP>(pk, tid, GH, GW, CB)) : 0u;
            if (PIN) {
                hv[q] = *reinterpret_cast<const chunk4_t<TS>*>(pin.dP + off);
                hp[q] = *reinterpret_cast<const chunk4_t<TS>*>(pin.P + off);
                hi[q] = *reinterpret_cast<const unsigned*>(pin.idx + off);
            } else {
                hv[q] = *reinterpret_cast<const chunk4_t<TS>*>(X + off);
            }
        }
    };
    auto halo_store = [&]() {
        const int c4 = (tid % CPP) * 4;
#pragma unroll
        for (int q = 0; q < NH; ++q) {
            const bool ok = (okm >> q) & 1u;
            const int pk = hm.pk[q];
            if (pk < 0) continue;
            if (PIN) {
                const int pr = pk & 255, pc = (pk >> 8) & 255;
                f32x4 v[4];
                unpool4x4(ok ? widen4(hv[q]) : f32x4{0, 0, 0, 0}, widen4(hp[q]), hi[q], v);
                __bf16* w0 = &Hs[((2 * pr - 1) * Gm::HWD + 2 * pc - 1) * LDC + c4];
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const bool in = ((pos >> 1) ? pr < Gm::HH / 2 : pr > 0) && ((pos & 1) ? pc < Gm::HWD / 2 : pc > 0);
                    if (in) *reinterpret_cast<bf16x4*>(w0 + ((pos >> 1) * Gm::HWD + (pos & 1)) * LDC) = to_bf16x4(v[pos]);
                }
            } else {
                *reinterpret_cast<bf16x4*>(&Hs[(tid / CPP + (kThreads / CPP) * q) * LDC + c4]) = ok ? as_bf16x4(hv[q]) : bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            }
        }
    };
    // the column block's weights: output channel co's nine taps x 32 channels are 288 consecutive halves of WB; chunk e = 8 of them
    auto weights_in = [&](int n0) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int e = tid + kThreads * q;
            if (NB * kThreads == BCH || e < BCH) {
                const int co = e / (K / 8), j = e - co * (K / 8);
                *reinterpret_cast<bf16x8*>(&Bs[((j >> 2) * BN + co) * LDC + (j & 3) * 8]) = *reinterpret_cast<const bf16x8*>(WB + (long long)(n0 + co) * K + j * 8);
            }
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    Pos pos = split(item);
    Item cur = item_of(pos);
    halo_load(cur);
    int loaded_n0 = -1;
    float bbr[NT];                                                    // the lane's bias values, loaded with the column block's weights (below)
#pragma unroll
    for (int t = 0; t < NT; ++t) bbr[t] = 0.f;
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
        if (!first) __syncthreads();                                  // the previous item's operands have been consumed
        first = false;
        if (cur.n0 != loaded_n0) {
            weights_in(cur.n0);
            if (EPI == 1 || EPI == 2 || EPI == 4) {
                // the bias with the weights, complete here: loaded in the epilogue it is the YOUNGEST load, and the wait for it (vmcnt counts
                // in order) is a wait for the next item's halo just prefetched -- every item paid its own prefetch latency (ISA, round 4)
#pragma unroll
                for (int t = 0; t < NT; ++t) bbr[t] = bias[cur.n0 + 32 * t + (lane & 31)];
                __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
            }
            loaded_n0 = cur.n0;
        }
        halo_store();
        __syncthreads();
        // (EPI 3, one column tile: the gate values are loaded ahead of the MFMAs -- sixteen registers; with two tiles they spill.  Issued
        // BEFORE the prefetch, and the prefetch unconditional (past the last item: this item's halo again, never used): the epilogue's wait
        // for the gates then allows exactly the prefetch's loads to stay in flight)
        constexpr bool GATE_AHEAD = EPI == 3 && NT == 1;
        float gate[NT][16];
        if constexpr (GATE_AHEAD) halo_gate_prefetch<TW, NT, TS>(gate, lane, wave, cur.img0, cur.oh0, cur.ow0, cur.n0, s, reinterpret_cast<const TS*>(bias));
        if (GATE_AHEAD) halo_load(nitem < n_items ? nxt : cur);
        else if (nitem < n_items) halo_load(nxt);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const __bf16* a = &Hs[((py + kh) * Gm::HWD + px + kw) * LDC + 8 * h];
                const __bf16* b = &Bs[((kh * 3 + kw) * BN + r) * LDC + 8 * h];
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
        __builtin_amdgcn_sched_barrier(0);                            // (nothing of the epilogue -- its waits least of all -- moves in front of the MFMAs)
        halo_epilogue<TW, NT, EPI, TS, TS>(acc, lane, wave, cur.img0, cur.oh0, cur.ow0, cur.n0, s, bias, Y, pool_idx, GATE_AHEAD ? gate : nullptr,
                                           (EPI == 1 || EPI == 2 || EPI == 4) ? bbr : nullptr);
        cur = nxt;
    }
}

}  // namespace rcnx
