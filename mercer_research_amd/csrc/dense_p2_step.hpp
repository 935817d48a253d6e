// dense_p2_step.hpp -- one train_batch step of the feature-sliced pipeline (dense_p2.hpp) as ONE launch.
//
// The two-kernel step is  k_p2_b (sample groups: slab sum, tail layers, deltas)  ->  k_p2_a (feature slices: dW_0, update,
// partial z_1 of the next batch): two dependent launches (1.6 us each inside a hipGraph) and, after each, a cold read of
// what the other kernel wrote.  The resident epoch kernel (dense_p2_persist.hpp) removed both launches but paid for two
// in-launch hand-offs per step, the big one being the 1.6 MB slab.  This form keeps the launch boundary where the big
// payload crosses (slab and parameters: written by launch j, read by launch j+1 with ordinary loads) and moves only the
// SMALL hand-off inside the launch:
//
//   blocks [0, B/8)            sample groups, exactly k_p2_b's work on batch j; they wait for nothing.  Their outputs
//                              (delta_1, a_1, delta_2, loss part: 2.5 KB per group) are stored write-through (sc1), the
//                              storing wave drains its stores, then publishes ONE flag word = the step's tag.
//   blocks [B/8, B/8 + G)      feature slices, k_p2_a's work.  Before looking at any flag a slice loads everything that
//                              does not depend on batch j's deltas -- its W_0 slice, its 16 features of batch j (for
//                              dW_0) and of batch j+1 (for the next partials) -- so those cold reads overlap the sample
//                              groups' work instead of following it.  Then: poll the flags of the sample groups it needs,
//                              sc1 loads of delta_1, MFMA, update, partial z_1 of batch j+1 into the slab (plain stores:
//                              the end of the launch publishes them).
//   the remaining blocks       tail tiles (db_0, [W_1 | b_1]): poll, sc1 loads, update in place; the first one also adds
//                              up the loss parts.
//
// Workgroups are dispatched in block order and the sample groups never wait, so the waiting blocks cannot starve them:
// unlike the resident kernel this form does NOT need every workgroup on the chip at once.  (The waits are bounded by the
// wall clock and raise a sticky error word all the same.)  The tag is read from a device word plus a per-node offset, so
// a captured graph replays with fresh tags: its last node advances the word.  Same arithmetic, same summation orders as
// the two-kernel pipeline -- the final cost is bit-identical (tests).  f32 only.
#pragma once

#include "dense_p2_persist.hpp"

namespace rcn {

struct StepBufs {
    float* slab;          // [B/8][G][8][32]   partial z_1 of the batch this launch's sample groups consume / next one produced
    float* a1;            // [B][32]   these four are written by the sample groups (sc1) and read inside the same launch (sc1)
    float* d1;            // [B][32]
    float* d2;            // [B][16]
    float* loss;          // [B/8]
    pw_t* d1w;            // [B][32]   delta_1 again, as self-validating words {value, tag}: the feature slices poll these directly
    unsigned* oflag;      // [B/8][16] tag of the newest complete outputs of sample group t, one 64-byte line per group
    const unsigned* tag;  // device word: tag of a step = *tag + joff
};

constexpr int kStepFlagStride = 16;          // words between two flags: pollers spread over 64-byte lines instead of one hot line

// ONE wave of a waiting workgroup polls every sample group's flag (lane t <- group t, t + 64, ...); the workgroup's other
// waves wait at the barrier that follows.  (Eight waves of 52 workgroups polling one line made each poll take ~2 us.)
__device__ inline bool step_wait_all(const unsigned* oflag, int NS, int lane, unsigned tag, long long timeout, const unsigned* err) {
    for (int t0 = 0; t0 < NS; t0 += 64) {
        const int t = t0 + lane;
        if (!px_wait_flags(oflag, (t < NS ? t : NS - 1) * kStepFlagStride, t < NS, tag, timeout, err)) return false;
    }
    return true;
}

inline int step_grid(const NetDesc& nd, size_t B) { return (int)(B / kP2Ts) + pipe_slices(nd) + pipe_extra_wgs(nd); }
inline bool step_supported(const NetDesc& nd, size_t B) { return p2_supported(nd, B); }

__global__ void k_add_u32(unsigned* w, unsigned v) { *w += v; }

// diagnostic build only (-DRCN_STAMPS): phase times of the 32nd and 33rd step of a graph, per workgroup (tools/stamps_step.py)
#define SSTAMP(i) do { if (joff == 32u) RCN_STAMP(0, i); else if (joff == 33u) RCN_STAMP(1, i); } while (0)

__global__ __launch_bounds__(kPersistThreads) void k_p2_step(
    NetDesc nd, float* __restrict__ params, const float* __restrict__ Xp, const float* __restrict__ Xn, const float* __restrict__ Ys, int B, int G,
    float scale, float loss_scale, float* __restrict__ loss_out, StepBufs bufs, unsigned joff, int do_fwd, unsigned* __restrict__ err, long long timeout,
    int first_look) {
    using T = float;
    using acc_t = Mfma16<T>::acc_t;
    using vec4 = Vec4<T>::type;
    __shared__ __attribute__((aligned(16))) float smem[kDenseWaves * kMtp * kRedTile + 16 * kP2H + 64 + 512];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const int F = nd.dims[0], H = nd.dims[1], C = nd.dims[2];
    const int NS = B / kP2Ts;
    const unsigned tag = *bufs.tag + joff;
    const auto r_d1 = PW_RSRC(bufs.d1, (size_t)B * kP2H * 4), r_a1 = PW_RSRC(bufs.a1, (size_t)B * kP2H * 4);
    const auto r_d2 = PW_RSRC(bufs.d2, (size_t)B * kP2C * 4), r_loss = PW_RSRC(bufs.loss, (size_t)NS * 4);
    const auto r_d1w = PW_RSRC(bufs.d1w, (size_t)B * kP2H * 8);
    const long long t_start = wall_clock64();
    auto fail = [&]() { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;      // an earlier step failed: touch nothing

    if ((int)blockIdx.x < NS) {
        // =============================================================== sample group t: samples 8 t .. 8 t + 7 (k_p2_b)
        constexpr int kPer = kP2MaxSlices / kP2BWaves;
        const int t = blockIdx.x, s0 = t * kP2Ts;
        vec4* zred = reinterpret_cast<vec4*>(smem);
        float* a1s = smem + kP2BWaves * 64 * 4;
        float* d2s = a1s + kP2H * kLd;
        float* frag = d2s + kP2C * kLd;
        SSTAMP(0);
        const vec4* sp = reinterpret_cast<const vec4*>(bufs.slab + (size_t)t * G * kP2Ts * kP2H) + lane;
        vec4 tq[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int g = wave + kP2BWaves * q;
            tq[q] = sp[(size_t)(g < G ? g : wave) * 64];
        }
        const float* W1 = params + nd.w_off[1];
        if (wave == 1 || wave == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ks = 4 * (wave - 1) + q, h = 4 * ks + g4;
                const float w = W1[(size_t)(h < H ? h : 0) * C + (n < C ? n : 0)];
                frag[ks * 64 + lane] = (h < H && n < C) ? w : 0.f;
            }
        } else if (wave == 3 || wave == 4) {
            const int mt = wave - 3;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = mt * 16 + n, c = 4 * ks + g4;
                const float w = W1[(size_t)(h < H ? h : 0) * C + (c < C ? c : 0)];
                frag[(8 + mt * 4 + ks) * 64 + lane] = (h < H && c < C) ? w : 0.f;
            }
        } else if (wave == 5) {
            const float* b1 = W1 + (size_t)C * H;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = Mfma16<T>::row(lane, i);
                const float bb = b1[c < C ? c : 0];
                frag[(16 + i) * 64 + lane] = c < C ? bb : 0.f;
            }
        } else if (wave == 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = Mfma16<T>::row(lane, i);
                frag[(20 + i) * 64 + lane] = Ys[(size_t)(s0 + (n & 7)) * C + (c < C ? c : 0)];
            }
        } else if (wave == 7) {
            const float* b0 = params + nd.w_off[0] + (size_t)H * F;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int h = 4 * (lane & 7) + i;
                const float b = b0[h < H ? h : 0];
                frag[(24 + i) * 64 + lane] = h < H ? b : 0.f;
            }
        }
        vec4 z = vec4{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < kPer; ++q)
            if (wave + kP2BWaves * q < G) z += tq[q];
        zred[wave * 64 + lane] = z;
        SSTAMP(1);
        __syncthreads();
        SSTAMP(2);
        if (wave != 0) return;
        z = ((zred[lane] + zred[64 + lane]) + (zred[128 + lane] + zred[192 + lane])) +
            ((zred[256 + lane] + zred[320 + lane]) + (zred[384 + lane] + zred[448 + lane]));
        {   // a_1 = sigmoid(z_1 + b_0)                                                                      rcn.rs:287-289
            const int s = lane >> 3, h0 = 4 * (lane & 7);
            float a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sg = sigmoid_fast(z[i] + frag[(24 + i) * 64 + lane]);   // unconditionally: as `c ? f(x) : 0` this is four branches
                a[i] = (h0 + i < H) ? sg : 0.f;
                a1s[(h0 + i) * kLd + s] = a[i];
            }
            px_store4(r_a1, ((s0 + s) * kP2H + h0) * 4, a[0], a[1], a[2], a[3]);
        }
        acc_t acc = acc_t{0, 0, 0, 0};                                // z_2 = W_1 a_1 + b_1
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float bv = a1s[(4 * ks + g4) * kLd + (n & 7)];
            acc = Mfma16<T>::mfma(frag[ks * 64 + lane], n < kP2Ts ? bv : 0.f, acc);
        }
        float lsum = 0.f;
        acc_t dv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = Mfma16<T>::row(lane, i);
            const float a2 = sigmoid_fast(acc[i] + frag[(16 + i) * 64 + lane]);
            const float diff = a2 - frag[(20 + i) * 64 + lane];
            const bool ok = c < C && n < kP2Ts;
            dv[i] = ok ? diff * (a2 * (1.f - a2)) : 0.f;              // rcn.rs:299
            lsum += ok ? diff * diff : 0.f;
            d2s[c * kLd + n] = dv[i];
        }
        if (n < kP2Ts) px_store4(r_d2, ((s0 + n) * kP2C + 4 * g4) * 4, dv[0], dv[1], dv[2], dv[3]);
#pragma unroll
        for (int mt = 0; mt < kMtp; ++mt) {                           // delta_1 = (W_1^T delta_2) (*) a_1 (1 - a_1)   rcn.rs:305-309
            acc_t ad = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ad = Mfma16<T>::mfma(frag[(8 + mt * 4 + ks) * 64 + lane], d2s[(4 * ks + g4) * kLd + n], ad);
            float o4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = a1s[(mt * 16 + Mfma16<T>::row(lane, i)) * kLd + (n & 7)];
                o4[i] = ad[i] * (a * (1.f - a));
            }
            if (n < kP2Ts) {
                px_store4(r_d1, ((s0 + n) * kP2H + mt * 16 + 4 * g4) * 4, o4[0], o4[1], o4[2], o4[3]);       // plain copy: the tail tile of b_0
                const int wo = ((s0 + n) * kP2H + mt * 16 + 4 * g4) * 8;                                      // tagged copy: the feature slices
                px_store4(r_d1w, wo, o4[0], __uint_as_float(tag), o4[1], __uint_as_float(tag));
                px_store4(r_d1w, wo + 16, o4[2], __uint_as_float(tag), o4[3], __uint_as_float(tag));
            }
        }
        lsum = wave_sum_lane0(lsum);
        if (lane == 0) px_store1(r_loss, t * 4, lsum);
        SSTAMP(3);
        px_drain();                                                   // only this wave stored: drain, then announce
        if (lane == 0) __hip_atomic_store(bufs.oflag + t * kStepFlagStride, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SSTAMP(4);
        return;
    }

    if ((int)blockIdx.x < NS + G) {
        // =============================================================== feature slice g of W_0 (k_p2_a)
        const int g = blockIdx.x - NS, f0 = g * 16;
        const int nf = F - f0 < 16 ? F - f0 : 16;
        float* red = smem;
        float* wsl = smem + kDenseWaves * kMtp * kRedTile;
        float* W0 = params + nd.w_off[0];
        const int ml = tid & 15, cl = (tid >> 4) & 15, mt = tid >> 8, m = mt * 16 + ml;
        const bool wvalid = m < H && cl < nf;
        const size_t off = (size_t)(f0 + (cl < nf ? cl : 0)) * H + (m < H ? m : 0);
        // ---- everything that does not depend on this batch's deltas, before the first look at a flag
        SSTAMP(0);
        float w = W0[off];
        const int ntile = B >> 4, kw = B >> 3;
        const float* __restrict__ cp = Xp + (size_t)g * B * 16;
        const float* __restrict__ cn = Xn + (size_t)g * B * 16;
        vec4 xn[2];
        if (do_fwd) {
#pragma unroll
            for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (wave + 8 * u) + n) * 16 + 4 * g4);
        }
        float bv0[8];
        {
            const float* xb = cp + (size_t)(wave * kw + g4) * 16 + n;
#pragma unroll
            for (int q = 0; q < 8; ++q) bv0[q] = xb[q * 64];
        }
        // ---- U: dW_0[:, slice] = sum_s delta_1[s] (x) x_s[slice]                                          rcn.rs:310
        acc_t acc[kMtp];
#pragma unroll
        for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
        // not before the sample groups can have finished: a poll round of all slices re-reads 3 MB through the fabric
        while (wall_clock64() - t_start < first_look) __builtin_amdgcn_s_sleep(8);
        bool bad = false;
        for (int kc = wave * kw; kc < (wave + 1) * kw && !bad; kc += 32) {
            float bv[8];
            if (kc == wave * kw) {
#pragma unroll
                for (int q = 0; q < 8; ++q) bv[q] = bv0[q];
            } else {
                const float* xb = cp + (size_t)(kc + g4) * 16 + n;
#pragma unroll
                for (int q = 0; q < 8; ++q) bv[q] = xb[q * 64];
            }
            // delta_1 rows n and 16 + n of samples kc + g4 + 4 q, polled until every word carries this step's tag
            float av[16];
            long long t0 = 0;
            for (unsigned it = 0;; ++it) {
                pu2 wv[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int at = (kc + g4 + 4 * q) * kP2H + n;
                    wv[q] = __builtin_amdgcn_raw_buffer_load_b64(r_d1w, at * 8, 0, 16);
                    wv[8 + q] = __builtin_amdgcn_raw_buffer_load_b64(r_d1w, (at + 16) * 8, 0, 16);
                }
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 16; ++q) ok = ok && wv[q][1] == tag;
                if (__all(ok ? 1 : 0)) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) av[q] = __uint_as_float(wv[q][0]);
                    break;
                }
                if ((it & 255u) == 255u && pw_give_up(t0, timeout, err)) { bad = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (bad) break;
            if (kc == wave * kw) SSTAMP(1);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[0] = Mfma16<T>::mfma(av[q], bv[q], acc[0]);
                acc[1] = Mfma16<T>::mfma(av[8 + q], bv[q], acc[1]);
            }
        }
        if (bad) fail();
        store_partials<T>(red, wave, lane, acc);
        SSTAMP(2);
        if (__syncthreads_or(bad ? 1 : 0)) return;                    // a timed-out wait: leave W_0 as it was
        SSTAMP(3);
        w = w - scale * sum_partials<T>(red, mt, cl, ml);             // rcn.rs:214
        if (wvalid) W0[off] = w;
        wsl[cl * kP2H + m] = wvalid ? w : 0.f;
        __syncthreads();
        SSTAMP(4);
        if (!do_fwd) return;
        // ---- F: partial z_1 of the next batch from the slice that is still in LDS
        float wf[4][kMtp];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < kMtp; ++t) wf[i][t] = wsl[(4 * g4 + i) * kP2H + t * 16 + n];
        for (int tb = 0; tb < ntile; tb += 16) {
            if (tb > 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) xn[u] = *reinterpret_cast<const vec4*>(cn + (size_t)(16 * (tb + wave + 8 * u) + n) * 16 + 4 * g4);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = 16 * (tb + wave + 8 * u) + n;
                acc_t a2[kMtp];
#pragma unroll
                for (int t = 0; t < kMtp; ++t) a2[t] = acc_t{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) a2[t] = Mfma16<T>::mfma(wf[i][t], xn[u][i], a2[t]);
                float* dst = bufs.slab + (((size_t)(s >> 3) * G + g) * kP2Ts + (s & 7)) * kP2H;
#pragma unroll
                for (int t = 0; t < kMtp; ++t) store4<T>(dst + t * 16, lane, a2[t]);
            }
        }
        SSTAMP(5);
        return;
    }

    {
        // =============================================================== tail tile e: db_0 or 16 columns of [W_1 | b_1]
        const int e = blockIdx.x - NS - G;
        const int jl = e == 0 ? 0 : 1, n0 = e == 0 ? F : (e - 1) * 16;
        const int Kin = nd.dims[jl], M = nd.dims[jl + 1];
        float* red = smem;
        const int c = n0 + n;
        const int mt_u = tid >> 8, o_u = tid & 255, cl_u = o_u >> 4, ml_u = o_u & 15;
        const int m_u = mt_u * 16 + ml_u, cc_u = n0 + cl_u;
        const bool pvalid = m_u < M && cc_u <= Kin;
        const int p_u = nd.w_off[jl] + (pvalid ? cc_u * M + m_u : 0);
        SSTAMP(0);
        float pv = params[p_u];
        const int kw = B >> 3;
        const int ldD = e == 0 ? kP2H : kP2C;
        acc_t acc[kMtp];
#pragma unroll
        for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
        __shared__ int s_bad2;
        if (tid == 0) s_bad2 = 0;
        __syncthreads();
        if (wave == 0 && !step_wait_all(bufs.oflag, NS, lane, tag, timeout, err)) s_bad2 = 1;
        __syncthreads();
        const bool bad = s_bad2 != 0;
        for (int kc = wave * kw; kc < (wave + 1) * kw && !bad; kc += 32) {
            const int r0 = n < M ? n : M - 1, r1 = 16 + n < M ? 16 + n : M - 1, ca = c < Kin ? c : Kin - 1;
            float v[24];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int smp = kc + g4 + 4 * q;
                v[q] = px_load1(e == 0 ? r_d1 : r_d2, (smp * ldD + r0) * 4);
                v[8 + q] = px_load1(e == 0 ? r_d1 : r_d2, (smp * ldD + r1) * 4);
                v[16 + q] = e != 0 ? px_load1(r_a1, (smp * kP2H + ca) * 4) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float b = e != 0 && c < Kin ? v[16 + q] : (c == Kin ? 1.f : 0.f);           // bias column: activation 1   rcn.rs:302,309
                acc[0] = Mfma16<T>::mfma(n < M ? v[q] : 0.f, b, acc[0]);
                acc[1] = Mfma16<T>::mfma(16 + n < M ? v[8 + q] : 0.f, b, acc[1]);
            }
        }
        if (bad) fail();
        store_partials<T>(red, wave, lane, acc);
        if (__syncthreads_or(bad ? 1 : 0)) return;
        SSTAMP(1);
        if (pvalid) params[p_u] = pv - scale * sum_partials<T>(red, mt_u, cl_u, ml_u);            // rcn.rs:214,221
        if (e == 0 && wave == 7 && loss_out) {                        // cost of this batch: parts added in group order
            float tot = 0.f;
            for (int t0 = 0; t0 < NS; t0 += 64) {
                const int t = t0 + lane;
                const float part = px_load1(r_loss, (t < NS ? t : NS - 1) * 4);
                for (int u = 0; u < 64 && t0 + u < NS; ++u) tot += __shfl(part, u, 64);
            }
            if (lane == 0) *loss_out = tot * loss_scale;
        }
    }
}

}  // namespace rcn
