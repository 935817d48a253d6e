// dense_xcd.hpp -- the feature-sliced pipeline of dense_p2.hpp as ONE resident kernel per epoch segment whose workgroups all sit
// on ONE XCD (the reference's own network shape class; f32 for batches of 1..256, and -- round 4 -- f64, the reference's own arithmetic
// type, for batches of 1..256 too: LDS holds 2 x the bytes per sample, the 256-sample instantiation one batch buffer instead of two).
//
// Why.  The two-kernel step (k_p2_b, k_p2_a) is bound by what surrounds its arithmetic: two dependent launch boundaries (~1.5 us
// each), W_0 and the batch leaving and re-entering the chip, and a 1.6 MB slab of partial sums written to and read back from
// memory every step (3.9x the algorithmic traffic by PMC).  Round 1 tried the obvious fix -- keep the workgroups resident and hand
// results over inside the launch -- and measured it SLOWER (11.7 us vs 9.6): its 85 workgroups were spread over all eight XCDs,
// whose L2s are not coherent with each other, so every hand-off had to be written through to memory and read back with L2 misses.
//
// What is different here.  An XCD has 32 CUs behind ONE coherent 4 MiB L2.  Blocks are dealt round-robin over the XCDs, so the
// blocks with blockIdx.x % 8 == 0 share one: the kernel is launched with 8 x 32 blocks, the 224 others return at once, and the 32
// WORKERS (one per CU of that XCD: each asks for > 80 KB of LDS) hand over through that L2 -- payloads as PLAIN stores (they stay
// dirty in L2 and are overwritten in place next step: no memory traffic at all), read with L1-bypassing (sc1) loads, announced by
// one flag word per producer after its waves drained their stores.  tools/ubench_xcd_ring.hip measured the two hand-offs of a
// step at 3.0-3.6 us together this way (5.3 us write-through, 5.9 us spread over the XCDs), with zero stale words in 2000 rounds;
// plain stores between DIFFERENT XCDs are stale every time, so placement is part of correctness here: every worker publishes its
// XCC_ID and checks all 32 before it trusts a plain store (a mismatch raises the sticky error word and nothing is written), and
// the host probes the placement once per context before it ever selects this path.
//
// Roles.  Worker w plays, every step, in this order:
//   sample group w   (all 32)          the work of k_p2_b for samples 8w..8w+7: wait for the slab parts of all feature workers and
//                                      the tail parameters -> sum in producer order -> a_1, layer 2, cost part, delta_2, delta_1.
//   feature worker w (w < NA)          the work of k_p2_a for the slice pair 2w, 2w+1 of W_0 (32 features): wait for delta_1 of all
//                                      sample groups -> dW_0 slice over the whole batch (MFMA, K split over the 8 waves) -> update
//                                      -> partial z_1 of the NEXT batch from the updated slice -> slab.  The slice of W_0 and the
//                                      batch's features live in LDS: the slice never leaves the CU until the launch ends, and a
//                                      batch is read from memory ONCE (for the forward of step j and the gradient of step j + 1),
//                                      prefetched a whole step ahead.
//   tail tile e      (w = NA + e)      db_0 or a 16-column tile of [W_1 | b_1]: gradient over the whole batch, update, and the new
//                                      values scattered into the operand-fragment image the sample groups read (p2_frag_scatter).
// Per step two hand-offs (slab: 25 producers -> 32 consumers; deltas / activations: 32 -> 28), no launch, no fence.
// Summation orders are fixed (producer order across slab parts, wave order inside a workgroup): results are bit-reproducible and
// agree with the two-kernel pipeline to f32 rounding of a different grouping (the oracle tolerances of tests/ hold for both).
//
// No hang by construction: every wait is bounded by the 100 MHz wall clock and raises a sticky error word that makes every
// worker leave at its next wait.  The parameters in memory are written only by a launch ALL of whose workers finished its last step
// (one closing flag round), and a launch that finds the word set leaves at once: after a failure the parameter vector is the state
// after the last launch that reports itself in `done` -- what the host's replay on the two-kernel pipeline starts from.
#pragma once

#include "dense_p2.hpp"
#include "dense_p2_dp.hpp"
#include "dp_push.hpp"

namespace rcn {

constexpr int kXcdWorkers = 32, kXcdThreads = 512, kXcdFlagStride = 32;      // flags: one 128-byte line each
constexpr int kXcdSl = 2;                                                      // 16-feature slices per feature worker

template <typename T>
struct XcdBufsT {
    T* slab;           // [B/8][NA][8][32]   partial z_1 of one batch, consumer-major
    T* d1;             // [B][32]            delta_1, a_1 [B][32], delta_2 [B][16], cost parts [B/8]: written by the sample groups
    T* a1;
    T* d2;
    T* loss;
    T* a2;             // [B][16]            two hidden layers only: a_2 and delta_3 (then d2 holds delta_2 of the middle layer)
    T* d3;
    T* fragimg;        // [28 | 40][64]      tail parameters as the sample groups' MFMA operand fragments
    unsigned* flagA;   // [32 x stride]      step tag of the newest complete slab part of feature worker w
    unsigned* flagB;   // [32 x stride]      step tag of the newest complete outputs of sample group w
    unsigned* flagT;   // [8 x stride]       step tag for which tail tile e's share of the fragment image is current
    unsigned* xcc;     // [32 x stride]      (launch tag << 4) | XCC_ID of worker w
    unsigned* flagD;   // [32 x stride]      tag of the last step of the newest launch worker w finished computing (before any write-back)
    unsigned* errd;    // [1]                device copy of the sticky error word
    unsigned* done;    // pinned host word: id of the newest launch whose workers ALL decided to commit (its parameters are in memory when it ends)
    unsigned* cw;      // [1]                the closing round's decision word: arrivals in the low half, the poison bit above (xcd_commit)
    unsigned* cdone;   // [32]               id of the newest launch whose write-back worker w completed (the host cross-checks `done` with it)
    long long* phase;  // [32][4]            diagnostic launches only (template parameter PH): per worker -- ticks spent inside the exchange, ticks of
                       //                    the whole step loop, steps, role (1 owner of its slice pair, 2 member, 3 tail tile)
};
using XcdBufs = XcdBufsT<float>;

inline int xcd_na(const NetDesc& nd) { return (pipe_slices(nd) + kXcdSl - 1) / kXcdSl; }
// one hidden layer (<= 32 units, <= 16 classes: the shape class of the two-kernel pipeline), or two (<= 32, <= 16 units, <= 16 classes)
inline bool xcd_two_hidden(const NetDesc& nd) {
    return nd.L == 3 && nd.dims[1] <= kP2H && nd.dims[2] <= kP2C && nd.dims[3] <= kP2C && pipe_slices(nd) <= kP2MaxSlices;
}
inline bool xcd_one_hidden(const NetDesc& nd) {
    return nd.L == 2 && nd.dims[1] <= kP2H && nd.dims[2] <= kP2C && pipe_slices(nd) <= kP2MaxSlices;
}
// The batch the kernel is instantiated for: the reference trains at batch_size 10 (rcn/src/main.rs:36-37) and BASELINE's CPU case is
// 32, the bench workload 256 -- any batch of 1..256 samples runs on the instantiation for the next of 32 / 64 / 128 / 256 (`BT`);
// the samples between B and BT are rows of zeros whose deltas are masked to zero, so they add nothing to any sum, and the update
// divides by the real batch.len() (rcn.rs:214).
constexpr int kXcdMaxB = 256;
// f64 (round 4): every LDS image is twice the bytes.  Up to 128 samples the layout is the f32 one (152 KB at 128 with the tail tiles'
// partials laid over the batch buffers they never use); the instantiation for 256 samples keeps ONE batch buffer instead of two (the
// next batch is fetched into registers under the gradient MFMAs and moves in behind them) and stages delta_1 in two halves: 152 KB too
constexpr int kXcdMaxB64 = 256;
inline int xcd_max_b(size_t esz) { return esz == 8 ? kXcdMaxB64 : kXcdMaxB; }
inline int xcd_bt(size_t B) { return B <= 32 ? 32 : B <= 64 ? 64 : B <= 128 ? 128 : 256; }
inline bool xcd_supported(const NetDesc& nd, size_t B, size_t esz = 4) {
    return (xcd_one_hidden(nd) || xcd_two_hidden(nd)) && B >= 1 && B <= (size_t)xcd_max_b(esz) && xcd_na(nd) + pipe_extra_wgs(nd) <= kXcdWorkers &&
           pipe_extra_wgs(nd) <= 8;
}
// workers of a launch: every feature worker and tail tile, and at least one worker per sample group of eight (BT / 8 <= 32)
inline int xcd_workers(const NetDesc& nd, int BT) {
    const int roles = xcd_na(nd) + pipe_extra_wgs(nd), groups = BT / kP2Ts;
    return roles > groups ? roles : groups;
}
inline size_t xcd_buf_bytes(const NetDesc& nd, size_t BT, size_t esz = 4) {
    const size_t NS = BT / kP2Ts, NA = (size_t)xcd_na(nd);
    return (NS * NA * kP2Ts * kP2H + 2 * BT * kP2H + 3 * BT * kP2C + NS + (size_t)kP3BFrag * 64) * esz +
           (size_t)(4 * kXcdWorkers + 8) * kXcdFlagStride * sizeof(unsigned) + 512 + 1024;   // flagA, flagB, xcc, flagD, flagT; decision word, committed ids; phase clocks
}
// LDS (elements): two batch buffers of a slice pair, delta_1 of the whole batch (rows padded to 48: conflict-free MFMA operand reads),
// the tail tiles' K-split partials, the slice pair of W_0, and the sample group's scratch (slab partial sums, a_1 / delta_2 tiles,
// target fragments): f32 148 KB of the CU's 160 at BT = 256 -- one worker per CU -- 94 / 66 / 51 KB at BT = 128 / 64 / 32;
// f64 152 / 130 / 101 KB at BT = 128 / 64 / 32
constexpr int kXcdD1Ld = 48;
// the tail tiles' partials over the batch buffers (a tail tile is never a feature worker): only where the LDS would not fit otherwise
template <typename T, int BT> constexpr bool xcd_red_aliased() { return sizeof(T) == 8 && BT >= 128; }
// ONE batch buffer and delta_1 in two halves: only where two buffers and the whole delta_1 do not fit (f64, 256 samples)
template <typename T, int BT> constexpr bool xcd_single_buffer() { return sizeof(T) == 8 && BT >= 256; }
constexpr size_t xcd_lds_floats(int BT, bool red_aliased = false, bool single = false) {
    return (single ? 1 : 2) * (size_t)kXcdSl * BT * 16 + (size_t)(single ? BT / 2 : BT) * kXcdD1Ld + (red_aliased ? 0 : (size_t)kDenseWaves * kMtp * kRedTile) +
           (size_t)kXcdSl * 16 * kP2H + (size_t)kP2BWaves * 64 * 4 + kP2H * kLd + 3 * kP2C * kLd + 4 * 64 + 64;
}
template <typename T, int BT> constexpr size_t xcd_lds_bytes() { return xcd_lds_floats(BT, xcd_red_aliased<T, BT>(), xcd_single_buffer<T, BT>()) * sizeof(T); }
static_assert(xcd_lds_bytes<double, 128>() + 64 <= 160 * 1024 && xcd_lds_bytes<float, 256>() + 64 <= 160 * 1024 && xcd_lds_bytes<double, 256>() + 64 <= 160 * 1024,
              "k_xcd_epoch: LDS of the largest instantiations");
static_assert((size_t)2 * kXcdSl * 128 * 16 >= (size_t)kDenseWaves * kMtp * kRedTile, "k_xcd_epoch: the partials fit the batch buffers they are laid over");

// diagnostic build only (-DRCN_STAMPS, tools/stamps_xcd.py): where each worker is at each point of the launch's last-but-one step
#ifdef RCN_STAMPS
#define XSTAMP(i) do { if (j == nb - 2 && lane == 0) g_rcn_stamps[0][w][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XCLOCK(k) do { if (lane == 0 && wave == 0) { g_rcn_stamps[1][w][2 * (k)] = __builtin_amdgcn_s_memtime(); g_rcn_stamps[1][w][2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define XSTAMP(i) do { } while (0)
#define XCLOCK(k) do { } while (0)
#endif

using xu4 = __attribute__((ext_vector_type(4))) unsigned;
#define XCD_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
// L1-bypassing (sc1, aux 16) reads of what another CU of this XCD stored in this launch; served by the shared L2.  byte_off is in BYTES
// of the buffer (an f64 quadruple is two 16-byte loads, an f64 word one 8-byte load).
template <typename T> __device__ inline typename Vec4<T>::type xcd_ld4(__amdgpu_buffer_rsrc_t r, int byte_off);
template <> __device__ inline Vec4<float>::type xcd_ld4<float>(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const xu4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    Vec4<float>::type f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}
template <> __device__ inline Vec4<double>::type xcd_ld4<double>(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const xu4 v[2] = {__builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16), __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16, 0, 16)};
    Vec4<double>::type f;
    __builtin_memcpy(&f, v, 32);
    return f;
}
template <typename T> __device__ inline T xcd_ld1(__amdgpu_buffer_rsrc_t r, int byte_off);
template <> __device__ inline float xcd_ld1<float>(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16);
    float f;
    __builtin_memcpy(&f, &v, 4);
    return f;
}
template <> __device__ inline double xcd_ld1<double>(__amdgpu_buffer_rsrc_t r, int byte_off) {
    typedef unsigned xu2 __attribute__((ext_vector_type(2)));
    const xu2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16);
    double f;
    __builtin_memcpy(&f, &v, 8);
    return f;
}
__device__ inline void xcd_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A flag is a PLAIN store like the payload it announces: it stays in the XCD's L2, which is where the polls (sc1 loads) look;
// a write-through (agent-scope atomic) store would drop the line from L2 and cost every poll a trip to memory (ubench: -0.2 us per
// hand-off).  Legal only between workers that share one XCD -- checked through the xcc table, which IS written write-through.
// (A workgroup-scope relaxed atomic store IS the plain global_store_dword; a `volatile` store compiles to a system-scope
// write-through flat_store sc0 sc1 plus a wait.)
__device__ inline void xcd_flag(unsigned* f, unsigned tag) { __hip_atomic_store(f, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void xcd_flag_wt(unsigned* f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The sticky error word lives twice: in pinned host memory (the host reads it after a synchronise without a copy in the stream) and in
// device memory (what the waits of this and every later launch look at: a launch that finds it set leaves at once, writing nothing).
// Round 4: the FIRST worker that gives up (compare-and-swap on the device word) also writes a record next to the host word -- which
// wait, who waited, for whom, at which step of which launch -- so that an expired wait names its site instead of reading "1".
enum XcdSite : unsigned {
    kXcdSitePlacement = 1,      // the placement vote: a worker's XCC answer never came (that worker is not resident), or the answers differ
    kXcdSiteTailFlag = 2,       // a sample group waiting for a tail tile's share of the fragment image (flagT)
    kXcdSiteSlabFlag = 3,       // a sample group waiting for a feature worker's slab part (flagA)
    kXcdSiteDeltaFlag = 4,      // a feature worker / tail tile waiting for a sample group's deltas (flagB)
    kXcdSitePushOwner = 5,      // data-parallel: the owner of a slice pair waiting for the other ranks' partial sums (reduce-scatter)
    kXcdSitePushMember = 6,     // data-parallel: a rank waiting for the owner's totals (all-gather)
    kXcdSitePushTail = 7,       // data-parallel: a tail parameter's all-to-all
    kXcdSitePushCost = 8,       // data-parallel: the cost's all-to-all
    kXcdSiteClosing = 9,        // the closing round: not every worker finished the launch's last step in time
};
constexpr int kXcdRecWords = 12;    // record at err_host + 4: site, worker, step, launch id, missing lo, missing hi, awaited tag, XCC_ID, rank, world, xsel, NW
// NOT inlined, on purpose (measured, A/B on one box, tools/ab_variants.py): inlined at its eleven sites the record's thirteen stores and their
// operands grew the step loop's code and cost the B = 256 step 1.4 % (6.53 against 6.44 us) although none of it ever runs -- instruction
// fetch of a loop whose cold branches are interleaved with its hot path; as one out-of-line copy the step is back at round 3's 6.44 us.
#ifdef RCN_AB_RAISE_INLINE          // (diagnostic builds of tools/ab_variants.py only: the inlined form this comment measures against)
__device__ inline void xcd_raise(unsigned* err_host, unsigned* err_dev, unsigned code, unsigned site, int worker, int step, unsigned launch,
                                 unsigned long long missing, unsigned tag, int rank = 0, int world = 1, int xsel = 0, int nw = 0) {
#else
__device__ __attribute__((noinline)) void xcd_raise(unsigned* err_host, unsigned* err_dev, unsigned code, unsigned site, int worker, int step, unsigned launch,
                                                    unsigned long long missing, unsigned tag, int rank = 0, int world = 1, int xsel = 0, int nw = 0) {
#endif
    unsigned expect = 0u;
    if (!__hip_atomic_compare_exchange_strong(err_dev, &expect, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    const unsigned rec[kXcdRecWords] = {site, (unsigned)worker, (unsigned)step, launch & 0x3fffffffu, (unsigned)missing, (unsigned)(missing >> 32), tag, id & 0xfu,
                                        (unsigned)rank, (unsigned)world, (unsigned)xsel, (unsigned)nw};
#pragma unroll
    for (int i = 0; i < kXcdRecWords; ++i) __hip_atomic_store(err_host + 4 + i, rec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err_host, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a workgroup barrier for data exchanged through LDS only: outstanding global stores are not waited for
__device__ inline void xcd_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ONE wave waits until every flag its lanes look at (lane < n0: f0[lane]; n0 <= lane < n0 + n1: f1[lane - n0]) carries a tag >= tag.
// Bounded: gives up after `timeout` ticks of the 100 MHz clock or when another worker raised the sticky error word.
// Returns 0 when every flag arrived, else the lanes (producers) whose flag was still behind when the wait gave up.
__device__ inline unsigned long long xcd_wait(const unsigned* f0, int n0, const unsigned* f1, int n1, unsigned tag, long long timeout, const unsigned* err) {
    const int lane = threadIdx.x & 63;
    const unsigned* p = lane < n0 ? f0 + lane * kXcdFlagStride : (lane < n0 + n1 ? f1 + (lane - n0) * kXcdFlagStride : nullptr);
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        const unsigned f = p ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
        if (__all((int)(f - tag) >= 0)) return 0ull;
        // (the rare part stays inline: as an out-of-line call it cost the B = 256 step 1.8 % -- 6.55 against 6.43 us, A/B on one box --
        // where moving the never-taken xcd_raise out of line had GAINED 1.4 %: a call inside the poll loop is paid on every exit)
        if ((it & 255u) == 255u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return __ballot((int)(f - tag) < 0);
        }
        // (no s_sleep between polls: measured 6.42 against 6.49 us per step with s_sleep(1); a SECOND poll in flight, issued before the
        // first is looked at, measured slower -- 6.64 -- the extra loads queue in front of the payload loads that follow)
    }
}

// The closing round as ONE decision every worker reads the same way (round 4; before, every worker waited for all flagD under its own
// clock, and a worker that arrived just after another had given up saw all flags set and wrote its slice: a torn parameter vector
// under a completed-looking launch).  Arrivals are counted in the low half of one word; a worker that gives up sets the poison bit
// by compare-and-swap, which succeeds only while the count is still short of NW.  So either the count reached NW unpoisoned -- then
// every worker, also one whose clock has run out, sees exactly that and writes back -- or the word was poisoned first, and then every
// worker, also one that arrives later, sees the poison and writes nothing.  Called by ONE lane per worker; worker 0 clears the word
// before it answers the placement vote (no worker arrives here before it has seen all answers).
constexpr unsigned kXcdPoison = 0x10000u;
__device__ inline bool xcd_commit(unsigned* cw, int NW, long long timeout, const unsigned* err, bool& gave_up, unsigned& seen) {
    unsigned v = __hip_atomic_fetch_add(cw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    long long t0 = 0;
    gave_up = false;
    for (unsigned it = 0;; ++it) {
        if (v & kXcdPoison) { seen = v; return false; }
        if ((int)(v & 0xffffu) == NW) { seen = v; return true; }
        if ((it & 63u) == 63u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                for (;;) {                                                // give up -- unless everybody has arrived in the meantime
                    if (v & kXcdPoison) { seen = v; return false; }
                    if ((int)(v & 0xffffu) == NW) { seen = v; return true; }
                    unsigned expect = v;
                    if (__hip_atomic_compare_exchange_strong(cw, &expect, v | kXcdPoison, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        gave_up = true;
                        seen = v;
                        return false;
                    }
                    v = expect;
                }
            }
        }
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The tail parameters as the sample groups' operand fragments, in the context's type.  f32: dense.hpp's image (shared with k_p2_b).
// f64: the same word / lane places for the A-operand words (the f64 MFMA reads A and B like the f32 one) and for b_0 (one word per
// element of the slab quadruple a lane sums); the words held per ACCUMULATOR element (b_1, b_2) follow the f64 accumulator map
// row = (lane >> 4) + 4 i (common.hpp) instead of 4 (lane >> 4) + i.
__device__ inline void xcd_frag_scatter2(int jl, int cc, int m, int H, float v, float* __restrict__ img) { p2_frag_scatter(jl, cc, m, H, v, img); }
__device__ inline void xcd_frag_scatter3(int cc, int m, int H2, float v, float* __restrict__ img) { p3_frag_scatter(cc, m, H2, v, img); }
__device__ inline void xcd_frag_scatter2(int jl, int cc, int m, int H, double v, double* __restrict__ img) {
    if (jl == 1 && cc < H) {
        const int h = cc, c = m;
        img[(h >> 2) * 64 + (h & 3) * 16 + c] = v;
        img[(8 + (h >> 4) * 4 + (c >> 2)) * 64 + (c & 3) * 16 + (h & 15)] = v;
    } else if (jl == 1) {
        const int c = m;                                                          // b_1[c]: element c >> 2 of the lanes with lane >> 4 == c & 3
#pragma unroll
        for (int nn = 0; nn < 16; ++nn) img[(16 + (c >> 2)) * 64 + (c & 3) * 16 + nn] = v;
    } else {
        const int h = m;                                                          // b_0[h] (bias column of W_0)
#pragma unroll
        for (int s = 0; s < 8; ++s) img[(24 + (h & 3)) * 64 + (h >> 2) + 8 * s] = v;
    }
}
__device__ inline void xcd_frag_scatter3(int cc, int m, int H2, double v, double* __restrict__ img) {
    if (cc < H2) {
        const int h2 = cc, c = m;
        img[(28 + (h2 >> 2)) * 64 + (h2 & 3) * 16 + c] = v;
        img[(32 + (c >> 2)) * 64 + (c & 3) * 16 + h2] = v;
    } else {
        const int c = m;                                                          // b_2[c]
#pragma unroll
        for (int nn = 0; nn < 16; ++nn) img[(36 + (c >> 2)) * 64 + (c & 3) * 16 + nn] = v;
    }
}

// The sigmoid of the resident kernel's single critical wave.  f32: hardware exp2 / rcp (common.hpp).  f64: the reference's
// 1 / (1 + E^-x) (rcn.rs:478-483) without the library calls' generality -- phase stamps put the f64 step's sample-group wave at 2.0 us of
// 6.4, most of it eight `exp` + IEEE divisions in two dependent rounds (~40 instructions each on a wave that issues alone): here
// e^t = 2^n e^r by Cody-Waite reduction (t = n ln2 + r, |r| <= ln2 / 2, ln2 in two pieces), e^r as its degree-13 Taylor polynomial
// (truncation 4e-18 relative), and the reciprocal as v_rcp_f64 + two Newton steps; |t| clamped to 708 (e^708 is finite; the true
// sigmoid there is 1 or 3e-308).  Relative error ~2e-16, far inside the f64 bar of 1e-11 per step (tests/test_gpu_round4.py).
__device__ inline float xcd_sigmoid(float x) { return sigmoid_fast(x); }
__device__ inline double xcd_sigmoid(double x) {
    double t = -x;
    t = t < -708.0 ? -708.0 : (t > 708.0 ? 708.0 : t);
    const double n = __builtin_rint(t * 1.44269504088896338700e+00);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, t);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;                                     // 1 / 13!
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double d = 1.0 + __builtin_ldexp(p, (int)n);
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return y;
}

// nb consecutive train_batch steps (rcn.rs:176-223) over the packed batches xs[j] (slice-major, k_pack_epoch), ys[j].
//
// DP = true: the data-parallel step (one rank per GPU, this rank's shard of every global batch in xs / ys).  Between the gradient
// MFMAs and the update the shards' partial gradients meet (dp_push.hpp): the feature workers' waves 0..3 push the four sums of a lane
// to the slice pair's owner rank (worker w -> rank w % world), which adds the ranks' rows in rank order and pushes the totals back --
// a reduce-scatter and an all-gather on self-validating {value, step} words, every poll local; the tail tiles' threads and the cost
// (element P) go all-to-all the same way.  Every rank applies the bit-identical update with the GLOBAL batch length (`scale` =
// eta / (B * world)).  xsel: which blocks are the workers (blockIdx.x % 8 == xsel) -- 0 on a GPU of its own; the one-GPU test
// harness gives each rank another XCD.
// the data-parallel form's arguments; the single-GPU instantiations carry an empty struct instead (no kernel-argument registers)
struct XcdDpOn { PushDesc pd; unsigned seq0; long long ptimeout; };
struct XcdDpOff {};
template <bool DP> struct XcdDpSel { using type = XcdDpOff; };
template <> struct XcdDpSel<true> { using type = XcdDpOn; };
template <bool DP> using XcdDpArgs = typename XcdDpSel<DP>::type;

// L3 = true: two hidden layers (dims F, H <= 32, H2 <= 16, C <= 16 -- the reference's own test net 784-10-10-10): the sample group
// runs one more 16 x 16 layer forward and backward (a_2, delta_3 through LDS; fragment words 28..39), the tail tiles cover
// [W_2 | b_2] as well (from a_2 / delta_3); everything about the big first layer is unchanged.
// GA = true: the gather form (rows fetched by the workers; opt-in, see below) -- its own instantiation, so that the default kernels do
// not carry its registers and branches.
// BT: the batch the kernel is built for (32 / 64 / 128 / 256); B <= BT is the real batch.len(): samples B .. BT-1 are rows of zeros in
// LDS and their deltas are masked, `scale` / `loss_scale` are formed from B on the host (rcn.rs:214).
// FULL: B == BT, known at compile time (measured: with the batch length a run-time value the BT = 256 step is 0.2 us slower -- the
// address arithmetic of the prefetch and of the targets, and the padding masks, sit on the step's critical waves).
// T: the context's arithmetic type.  f64 (the reference's own, rcn.rs:28,31,49) runs the same roles on v_mfma_f64_16x16x4_f64: the
// accumulator's row map differs (Mfma16<T>::row), so a feature-worker lane owns parameters hidden g4 + 4 i instead of 4 g4 + i, and
// 16-byte L2 reads become two per quadruple; single-GPU forms only (the exchange's words carry f32 values).
// PH (diagnostic, data-parallel form only; option "xcd_dp_phase"): every worker clocks the time it spends inside the exchange and the
// whole step loop (two reads of the 100 MHz clock per step -- most of a microsecond each, so the figures describe where a step waits,
// not what an unclocked step costs) and leaves them in bufs.phase: what rcn_hip_dp_phase_us reports after a first multi-GPU run.
template <typename T, int BT, bool FULL, bool DP, bool L3 = false, bool GA = false, bool PH = false>
__global__ __launch_bounds__(kXcdThreads) void k_xcd_epoch(
    NetDesc nd, T* __restrict__ params, const T* __restrict__ xs_all, const T* __restrict__ ys_all, int B_arg, int nb, int G, T scale,
    T loss_scale, T* __restrict__ loss_dev, XcdBufsT<T> bufs, unsigned tag0, unsigned* __restrict__ err, long long timeout, XcdDpArgs<DP> dp,
    int xsel, const int* __restrict__ gperm, unsigned launch_id) {
    constexpr bool gather = GA;
    const int B = FULL ? BT : B_arg;
    static_assert(!GA || FULL, "the gather form exists for the full batch only");
    static_assert(BT == 32 || BT == 64 || BT == 128 || BT == 256, "k_xcd_epoch: batch instantiations");
    static_assert(!GA || BT == 256, "the gather form exists for batch 256 only");
    static_assert(sizeof(T) == 4 || (!DP && !GA && BT <= kXcdMaxB64), "f64: single-GPU forms on the packed image");
    constexpr bool SB = xcd_single_buffer<T, BT>();                 // one batch buffer, delta_1 staged in DH parts (f64 at 256 samples)
    constexpr int DH = SB ? 2 : 1, BH = BT / DH;
    static_assert(!PH || DP, "the phase clocks exist for the data-parallel form");
    constexpr int ES = (int)sizeof(T), VS = 4 * ES;                  // bytes of an element / of a quadruple in the L2 buffers
    constexpr int RI = sizeof(T) == 4 ? 1 : 4;                       // Mfma16<T>::row(lane, i) = row(lane, 0) + RI * i
    constexpr size_t kXs = (size_t)kXcdSl * BT * 16;              // one LDS batch buffer of a slice pair
    constexpr int NS = BT / kP2Ts;                                  // sample groups of eight
    using acc_t = typename Mfma16<T>::acc_t;
    using vec4 = typename Vec4<T>::type;
    // the workers: the blocks of ONE residue class of blockIdx.x % 8 (they share an XCD, whichever the dispatch started on) -- xsel < 8:
    // that class; xsel >= 8 (several ranks sharing ONE device in the test harness): the class that landed on PHYSICAL XCD xsel - 8,
    // so that the ranks' kernels sit on different XCDs whatever XCD each rank's dispatch happened to start its round-robin on
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xfu;
    if (xsel < 8 ? (int)(blockIdx.x & 7) != xsel : (int)my_xcc != xsel - 8) return;      // the other seven XCDs' blocks
    const int w = (int)(blockIdx.x >> 3);
    const int F = nd.dims[0], H = nd.dims[1], Cm = nd.dims[2], C = nd.dims[L3 ? 3 : 2];     // Cm: the units after W_1 (the classes, or h2)
    const int NA = (G + kXcdSl - 1) / kXcdSl, NT = 1 + nd.tile_start[nd.L] - nd.tile_start[1];    // = pipe_extra_wgs(nd)
    const int NW = NA + NT > NS ? NA + NT : NS;                     // workers of this launch (host: xcd_workers)
    if (w >= NW) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    __shared__ int s_abort;
    T* smem = reinterpret_cast<T*>(smem_dyn);
    T* xbuf = smem;                                                 // [2 | SB: 1][kXcdSl][B][16]
    T* d1s = xbuf + (SB ? 1 : 2) * kXs;                             // delta_1 of the batch [BH][48] (32 used)
    constexpr bool RA = xcd_red_aliased<T, BT>();                   // (f64, BT >= 128: laid over the batch buffers a tail tile never uses)
    T* red = RA ? xbuf : d1s + BH * kXcdD1Ld;                       // tail tiles: [wave][mt][16 x kLd]
    T* wsl = d1s + BH * kXcdD1Ld + (RA ? 0 : kDenseWaves * kMtp * kRedTile);     // [slice][feature 0..15][32]
    vec4* zred = reinterpret_cast<vec4*>(wsl + kXcdSl * 16 * kP2H); // [8 waves][64 lanes]
    T* a1s = reinterpret_cast<T*>(zred) + kP2BWaves * 64 * 4;       // a_1 tile  [hidden 32][kLd]
    T* d2s = a1s + kP2H * kLd;                                      // delta_2   [class 16][kLd]
    T* frag = d2s + kP2C * kLd;                                     // [4][lane]: the targets per accumulator element
    T* a2s = frag + 4 * 64;                                         // two hidden layers: a_2 [h2 16][kLd], delta_3 [class 16][kLd]
    T* d3s = a2s + kP2C * kLd;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g4 = lane >> 4;
    const bool is_a = w < NA, is_t = w >= NA && w < NA + NT, is_s = w < NS;
    int dp_rank = 0, dp_world = 1;                                  // (for the time-out record)
    if constexpr (DP) { dp_rank = dp.pd.pd.rank; dp_world = dp.pd.pd.world; }
    const int e = w - NA;                                           // tail tile index when is_t
    const int nsl = is_a ? (G - kXcdSl * w < kXcdSl ? G - kXcdSl * w : kXcdSl) : 0;
    const size_t xs_stride = (size_t)G * B * 16, ys_stride = (size_t)B * C;
    const auto r_slab = XCD_RSRC(bufs.slab, (size_t)NS * NA * kP2Ts * kP2H * ES);
    const auto r_d1 = XCD_RSRC(bufs.d1, (size_t)BT * kP2H * ES), r_a1 = XCD_RSRC(bufs.a1, (size_t)BT * kP2H * ES);
    const auto r_d2 = XCD_RSRC(bufs.d2, (size_t)BT * kP2C * ES), r_loss = XCD_RSRC(bufs.loss, (size_t)NS * ES);
    const auto r_img = XCD_RSRC(bufs.fragimg, (size_t)(L3 ? kP3BFrag : kP2BFrag) * 64 * ES);
    const auto r_a2 = XCD_RSRC(bufs.a2, (size_t)BT * kP2C * ES), r_d3 = XCD_RSRC(bufs.d3, (size_t)BT * kP2C * ES);

    if (tid == 0) s_abort = 0;
    __syncthreads();
    // ---- placement: every worker says where it runs; nobody trusts a plain store before it has seen the answers of all workers.
    // Before it answers, a worker sets the flags it owns to "one before this launch's first tag": the protocol then does not depend on
    // what the workspace held (zeros from its creation, the tags of an earlier launch -- or, seen once in round 3 right after another
    // context had been destroyed, flag words of that context's workspace at the same address, which let sample groups of the new
    // context's first step run ahead of their slab).  The answer carries the launch's process-wide id, so an answer left by any
    // earlier launch of any context never counts.
    // (an answer: launch id << 8 | residue class of the block << 4 | XCC_ID -- all workers of a launch must agree on the low byte)
    const unsigned vtag = launch_id & 0x00ffffffu;
    if (tid == 0) {
        const unsigned id = ((blockIdx.x & 7u) << 4) | my_xcc;
        xcd_flag(bufs.flagA + w * kXcdFlagStride, tag0 - 1u);
        xcd_flag(bufs.flagB + w * kXcdFlagStride, tag0 - 1u);
        xcd_flag(bufs.flagD + w * kXcdFlagStride, tag0 - 1u);
        if (w < 8) xcd_flag(bufs.flagT + w * kXcdFlagStride, tag0 - 1u);
        if (w == 0) __hip_atomic_store(bufs.cw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the closing round's decision word (xcd_commit)
        xcd_drain();
        xcd_flag_wt(bufs.xcc + w * kXcdFlagStride, (vtag << 8) | id);
    }
    if (wave == 0) {
        long long t0 = 0;
        bool ok = true;
        unsigned mine = 0, v = 0;
        const bool stale = __hip_atomic_load(bufs.errd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;     // an earlier launch failed: the
        if (stale && lane == 0) s_abort = 1;                          // parameters in memory are not this launch's starting point -> leave
        if (!stale)
        for (unsigned it = 0;; ++it) {
            v = lane < NW ? __hip_atomic_load(bufs.xcc + lane * kXcdFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (vtag << 8);
            if (__all((v >> 8) == vtag)) break;
            if ((it & 255u) == 255u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > timeout || __hip_atomic_load(bufs.errd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        mine = __shfl(v, w < 64 ? w : 0, 64) & 0xffu;
        const bool same = __all(lane >= NW || (v & 0xffu) == mine);
        // (missing: the workers whose answer never came -- or, code 2, the workers that answered from another XCD / residue class than this one)
        const unsigned long long absent = ok ? __ballot(lane < NW && (v & 0xffu) != mine) : __ballot(lane < NW && (v >> 8) != vtag);
        if (lane == 0 && !stale && !(ok && same)) {
            s_abort = 1;
            xcd_raise(err, bufs.errd, ok ? 2u : 1u, kXcdSitePlacement, w, -1, launch_id, absent, vtag, dp_rank, dp_world, xsel, NW);      // 2: workers on different XCDs; 1: a wait expired
        }
    }
    __syncthreads();
    if (s_abort) return;
    // (test hook, option "xcd_fault_launch": bit 31 of the launch id makes worker 1 leave here, as a worker that never became resident
    // would -- every wait for it expires and the launch fails without having written anything)
    if ((launch_id >> 31) != 0u && w == 1) return;
    // (second test hook, bit 30: worker 1 reaches the closing round late -- after every other worker's wait there has expired)
    const bool late_closer = ((launch_id >> 30) & 1u) != 0u && w == 1;

    // ---- feature worker: its slice pair of W_0 into LDS and registers; the first two batches into the two LDS buffers.
    // The gradient of the slice pair is four 16 x 16 tiles (slice sl, hidden half mt); wave u owns tile u & 3 for the samples of K-half
    // u >> 2, and waves 0..3 also own the tile's parameters, in the accumulator's layout: lane (n, g4), element i <-> hidden
    // 16 mt + Mfma16<T>::row(lane, i) (f32: 4 g4 + i, four consecutive parameters; f64: g4 + 4 i), feature 16 sl + n.
    // gather form (no packed image: xs_all = X[rows][F] as stored, ys_all = Y[rows][C], gperm = the epoch's order or NULL for the
    // stored order): element i of this worker's LDS image of batch b -- slice i / 4B of the pair, sample (i / 4) % B, 16-byte chunk
    // i % 4 -- straight from row gperm[b B + sample]; four neighbouring threads fetch one 64-byte run of a row, and the sample's
    // other slice comes from the same 128-byte line of that row (F % 4 == 0: a chunk is inside the row or past it)
    auto xrow4 = [&](int b, int i) -> vec4 {
        const int sl = i >> 10, smp = (i >> 2) & 255, col = (kXcdSl * w + sl) * 16 + 4 * (i & 3);      // B == 256
        const unsigned at = (unsigned)(b * B + smp);
        const unsigned row = gperm ? (unsigned)gperm[at] : at;
        const vec4 v = *reinterpret_cast<const vec4*>(xs_all + (size_t)row * F + (col < F ? col : 0));
        return (sl < nsl && col < F) ? v : vec4{0, 0, 0, 0};
    };
    T* W0 = params + nd.w_off[0];
    const int mt = tid >> 8;                                        // (tail tiles: M-tile of this thread's parameter)
    const int tl = wave & 3, kh = wave >> 2, usl = tl >> 1, umt = tl & 1;
    T wcur[4] = {0, 0, 0, 0};
    bool wvalid[4] = {false, false, false, false};
    unsigned woff0 = 0;                                              // this lane's four parameters in W_0: woff0 + RI i (f32: consecutive, hidden 4 g4 + i)
    if (is_a) {
        if (kh == 0) {
            const int f0 = (kXcdSl * w + usl) * 16;
            const int nf = usl < nsl ? (F - f0 < 16 ? F - f0 : 16) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hid = umt * 16 + Mfma16<T>::row(lane, i);
                wvalid[i] = hid < H && n < nf;
                if (i == 0) woff0 = wvalid[0] ? (unsigned)((f0 + n) * H + hid) : 0u;      // (wvalid[0] false => all four are: hid grows with i)
                const T v = W0[wvalid[i] ? woff0 + RI * i : 0u];
                wcur[i] = wvalid[i] ? v : (T)0;
            }
            store4<T>(wsl + (usl * 16 + n) * kP2H + umt * 16, lane, acc_t{wcur[0], wcur[1], wcur[2], wcur[3]});
        }
        // batches 0 and 1 of this launch: [slice pair][sample][16] is one contiguous run of nsl * B * 16 floats per batch
        // (in LDS a slice holds BT rows: rows B .. BT-1, and a missing second slice, are zeros)
        for (int b = 0; b < (SB ? 1 : 2) && b < nb; ++b) {
            vec4* dst = reinterpret_cast<vec4*>(xbuf + (size_t)(b & 1) * kXs);
            if constexpr (gather) {
                for (int i = tid; i < kXcdSl * BT * 4; i += kXcdThreads) dst[i] = xrow4(b, i);
            } else {
                const vec4* src = reinterpret_cast<const vec4*>(xs_all + (size_t)b * xs_stride + (size_t)(kXcdSl * w) * B * 16);
                for (int i = tid; i < kXcdSl * BT * 4; i += kXcdThreads) {
                    const int sl = i / (BT * 4), r = i % (BT * 4);
                    dst[i] = (sl < nsl && (r >> 2) < B) ? src[sl * B * 4 + r] : vec4{0, 0, 0, 0};
                }
            }
        }
    }
    // tail tile: this thread's parameter (column cc of [W_jl | b_jl], row tm), kept in a register until the launch ends
    T tcur = 0;
    size_t tp = 0;
    bool tvalid = false;
    // tail tile e -> (layer jl, first column n0 of [W_jl | b_jl]): e == 0 the bias column of W_0, then the tiles of layer 1, then of layer 2
    const int nt1 = nd.tile_start[2] - nd.tile_start[1];            // 16-column tiles of [W_1 | b_1]
    auto tile_of = [&](int& jl, int& n0) {
        if (e == 0) { jl = 0; n0 = F; }
        else if (!L3 || e - 1 < nt1) { jl = 1; n0 = (e - 1) * 16; }
        else { jl = 2; n0 = (e - 1 - nt1) * 16; }
    };
    if (is_t) {
        int jl, n0;
        tile_of(jl, n0);
        const int Kin = nd.dims[jl], M = nd.dims[jl + 1];
        const int o = tid & 255, tm = mt * 16 + (o & 15), cc = n0 + (o >> 4);
        tvalid = tm < M && cc <= Kin;
        tp = tvalid ? (size_t)nd.w_off[jl] + (size_t)cc * M + tm : 0;
        tcur = params[tp];
    }
    __syncthreads();

    // partial z_1 of batch `jn` (LDS buffer jn & 1) from the slice pair in LDS -> this worker's part of the slab
    auto forward = [&](int jn) {
        const T* xb = xbuf + (SB ? (size_t)0 : (size_t)(jn & 1) * kXs);
        constexpr int NTILE = BT / 16, UT = NTILE > kDenseWaves ? NTILE / kDenseWaves : 1;      // sample tiles, and how many a wave takes
        T wf[kXcdSl][4][kMtp];
#pragma unroll
        for (int sl = 0; sl < kXcdSl; ++sl)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < kMtp; ++t) wf[sl][i][t] = wsl[(sl * 16 + 4 * g4 + i) * kP2H + t * 16 + n];
        if (NTILE >= kDenseWaves || wave < NTILE)
#pragma unroll
        for (int u = 0; u < UT; ++u) {                               // BT = 256: 16 sample tiles, two per wave; BT = 32: two tiles, waves 0 and 1
            const int s = 16 * (wave + 8 * u) + n;
            vec4 xv[kXcdSl];
#pragma unroll
            for (int sl = 0; sl < kXcdSl; ++sl) xv[sl] = *reinterpret_cast<const vec4*>(xb + ((size_t)sl * BT + s) * 16 + 4 * g4);
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < kXcdSl; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(wf[sl][i][t], xv[sl][i], acc[t]);
            T* dst = bufs.slab + (((size_t)(s >> 3) * NA + w) * kP2Ts + (s & 7)) * kP2H;
#pragma unroll
            for (int t = 0; t < kMtp; ++t) store4<T>(dst + t * 16, lane, acc[t]);
        }
        xcd_drain();                                                  // every storing wave, before the barrier in front of the flag
    };

    if (is_a) {
        forward(0);
        __syncthreads();
        if (tid == 0) xcd_flag(bufs.flagA + w * kXcdFlagStride, tag0);
    } else if (is_t) {
        // this tile's share of the operand-fragment image, from the parameter vector (the image's pads are the zeros it was created with)
        if (tvalid) {
            int jl, n0;
            tile_of(jl, n0);
            const int o = tid & 255;
            if (L3 && jl == 2) xcd_frag_scatter3(n0 + (o >> 4), mt * 16 + (o & 15), Cm, tcur, bufs.fragimg);
            else xcd_frag_scatter2(jl, n0 + (o >> 4), mt * 16 + (o & 15), H, tcur, bufs.fragimg);
        }
        xcd_drain();
        __syncthreads();
        if (tid == 0) xcd_flag(bufs.flagT + e * kXcdFlagStride, tag0);
    }

    const bool live = n < kP2Ts && (FULL || w * kP2Ts + n < B);                 // sample group: this lane's column of the 16-wide tiles is a sample of the batch
    unsigned yrow = 0;                                                // gather form: the row of this lane's sample (8 w + n % 8) in batch j
    if (gather) {
        const unsigned at = (unsigned)(w * kP2Ts + (n & 7));
        yrow = gperm ? (unsigned)gperm[at] : at;
    }
    long long ph_x = 0, ph_t0 = 0;                                    // (PH) ticks inside the exchange; the loop's start
    if constexpr (PH) { if (tid == 0) ph_t0 = wall_clock64(); }
    for (int j = 0; j < nb; ++j) {
        const unsigned tag = tag0 + (unsigned)j;
        if (j == 8) XCLOCK(0);
        if (j == nb - 8) XCLOCK(1);
        const bool more = j + 1 < nb;
        // (1) the batch after next, a whole step ahead: into registers now, into the LDS buffer step j's gradient frees
        constexpr int XR = (kXcdSl * BT * 4 + kXcdThreads - 1) / kXcdThreads;      // 16-byte pieces of a batch's slice pair per thread: 4 at BT = 256
        vec4 xr[XR];
        const bool pre = is_a && (SB ? j + 1 < nb : j + 2 < nb);       // (SB: the NEXT batch, fetched further down, under the gradient MFMAs)
        auto prefetch = [&](int jb) {
            const vec4* src = reinterpret_cast<const vec4*>(xs_all + (size_t)jb * xs_stride + (size_t)(kXcdSl * w) * B * 16);
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                const int i = tid + r * kXcdThreads, sl = i / (BT * 4), rr = i % (BT * 4);
                xr[r] = (sl < nsl && (rr >> 2) < B) ? src[sl * B * 4 + rr] : vec4{0, 0, 0, 0};
            }
        };
        if (pre && !SB) {
            if constexpr (gather) {
#pragma unroll
                for (int r = 0; r < XR; ++r) xr[r] = xrow4(j + 2, tid + r * kXcdThreads);
            } else {
                const vec4* src = reinterpret_cast<const vec4*>(xs_all + (size_t)(j + 2) * xs_stride + (size_t)(kXcdSl * w) * B * 16);
                if (FULL) {                                           // the full batch: the image's slice pair is the LDS image
#pragma unroll
                    for (int r = 0; r < XR; ++r) {
                        const int i = tid + r * kXcdThreads;
                        xr[r] = i < nsl * BT * 4 ? src[i] : vec4{0, 0, 0, 0};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < XR; ++r) {
                        const int i = tid + r * kXcdThreads, sl = i / (BT * 4), rr = i % (BT * 4);
                        xr[r] = (sl < nsl && (rr >> 2) < B) ? src[sl * B * 4 + rr] : vec4{0, 0, 0, 0};
                    }
                }
            }
        }

        // =============================================================== sample group w: samples 8w .. 8w+7 of batch j
        if (is_s) {
            const int s0 = w * kP2Ts;
            const int srow = (FULL || s0 + (n & 7) < B) ? s0 + (n & 7) : B - 1;                             // (a padding sample reads a live row; its delta is masked)
            const T* Ys = gather ? ys_all + (size_t)yrow * C
                                 : ys_all + (size_t)j * ys_stride + (size_t)srow * C;             // this lane's sample's targets
            T fr[L3 ? kP3BFrag : kP2BFrag];
            if (wave == 0) {
                XSTAMP(0);
                // the tail tiles finish well before the feature workers: their flags first, the 24 parameter words of the image
                // fetched under the wait for the slab
                const unsigned long long mT = xcd_wait(bufs.flagT, NT, nullptr, 0, tag, timeout, bufs.errd);
#pragma unroll
                for (int q = 0; q < (L3 ? kP3BFrag : kP2BFrag); ++q) fr[q] = (q >= 20 && q < 24) ? (T)0 : xcd_ld1<T>(r_img, (q * 64 + lane) * ES);
                const unsigned long long mA = mT ? 0ull : xcd_wait(bufs.flagA, NA, nullptr, 0, tag, timeout, bufs.errd);
                if ((mT | mA) != 0ull && lane == 0) {
                    s_abort = 1;
                    xcd_raise(err, bufs.errd, 1u, mT ? kXcdSiteTailFlag : kXcdSiteSlabFlag, w, j, launch_id, mT ? mT : mA, tag, dp_rank, dp_world, xsel, NW);
                }
                XSTAMP(1);
            } else if (wave == 6) {                                   // targets per accumulator element: no flag to wait for
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = Mfma16<T>::row(lane, i);
                    frag[i * 64 + lane] = Ys[c < C ? c : 0];
                }
                if (gather && more) {                                 // the next step's row of this lane's sample: its index a step ahead
                    const unsigned at = (unsigned)((j + 1) * B + s0 + (n & 7));
                    yrow = gperm ? (unsigned)gperm[at] : at;
                }
            }
            __syncthreads();
            if (s_abort) return;
            // slab: this group's NA x 1 KB, contiguous; wave q sums producers q, q + 8, q + 16, q + 24 in that order
            vec4 z = vec4{0, 0, 0, 0};
            {
                vec4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = wave + kP2BWaves * q;
                    t[q] = xcd_ld4<T>(r_slab, (int)((((size_t)w * NA + (p < NA ? p : wave)) * 64 + lane) * VS));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (wave + kP2BWaves * q < NA) z += t[q];
            }
            zred[wave * 64 + lane] = z;
            if (wave == 7) XSTAMP(2);
            __syncthreads();
            if (wave == 0) {
                XSTAMP(3);
                {   // fixed order across waves (as k_p2_b)
                    typedef T h2 __attribute__((ext_vector_type(2)));
                    vec4 r[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] = zred[q * 64 + lane];
                    h2 lo[8], hi[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { lo[q] = h2{r[q][0], r[q][1]}; hi[q] = h2{r[q][2], r[q][3]}; }
                    const h2 zl = ((lo[0] + lo[1]) + (lo[2] + lo[3])) + ((lo[4] + lo[5]) + (lo[6] + lo[7]));
                    const h2 zh = ((hi[0] + hi[1]) + (hi[2] + hi[3])) + ((hi[4] + hi[5]) + (hi[6] + hi[7]));
                    z = vec4{zl[0], zl[1], zh[0], zh[1]};
                }
#pragma unroll
                for (int q = 20; q < 24; ++q) fr[q] = frag[(q - 20) * 64 + lane];
                // a_1 = sigmoid(z_1 + b_0); lane <- sample lane >> 3, hidden 4 (lane & 7) + i                        rcn.rs:287-289
                {
                    const int s = lane >> 3, h0 = 4 * (lane & 7);
                    vec4 a;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto sg = xcd_sigmoid(z[i] + fr[24 + i]);
                        a[i] = (h0 + i < H) ? sg : (T)0;
                        a1s[(h0 + i) * kLd + s] = a[i];
                    }
                    *reinterpret_cast<vec4*>(bufs.a1 + (size_t)(s0 + s) * kP2H + h0) = a;
                }
                // z_2 = W_1 a_1 + b_1, a_2 = sigmoid, delta_2 = (a_2 - y) (*) a_2 (1 - a_2)                          rcn.rs:287-289, 299
                acc_t acc = acc_t{0, 0, 0, 0};
                {
                    T bv[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) bv[ks] = a1s[(4 * ks + g4) * kLd + (n & 7)];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) acc = Mfma16<T>::mfma(fr[ks], n < kP2Ts ? bv[ks] : (T)0, acc);      // (as two interleaved chains: 6.48 vs 6.49 us, nothing)
                }
                T lsum = 0;
                acc_t dv;
                if constexpr (!L3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = Mfma16<T>::row(lane, i);
                        const T a2 = xcd_sigmoid(acc[i] + fr[16 + i]);
                        const T diff = a2 - fr[20 + i];
                        const bool ok = c < C && live;
                        dv[i] = ok ? diff * (a2 * ((T)1 - a2)) : (T)0;
                        lsum += ok ? diff * diff : (T)0;
                        d2s[c * kLd + n] = dv[i];
                    }
                } else {
                    // a_2 = sigmoid(z_2 + b_1): element i of the accumulator is unit h2 = 4 g4 + i of sample n                 rcn.rs:287-289
                    acc_t a2r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int h2 = Mfma16<T>::row(lane, i);
                        const T a = xcd_sigmoid(acc[i] + fr[16 + i]);
                        a2r[i] = (h2 < Cm && n < kP2Ts) ? a : (T)0;
                        a2s[h2 * kLd + n] = a2r[i];
                    }
                    if (n < kP2Ts) store4<T>(bufs.a2 + (size_t)(s0 + n) * kP2C, lane, a2r);
                    // z_3 = W_2 a_2 + b_2, a_3 = sigmoid, delta_3 = (a_3 - y) (*) a_3 (1 - a_3)                               rcn.rs:287-289, 299
                    acc_t acc3 = acc_t{0, 0, 0, 0};
                    {
                        T bv[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) bv[ks] = a2s[(4 * ks + g4) * kLd + (n & 7)];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) acc3 = Mfma16<T>::mfma(fr[28 + ks], n < kP2Ts ? bv[ks] : (T)0, acc3);
                    }
                    acc_t d3v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = Mfma16<T>::row(lane, i);
                        const T a3 = xcd_sigmoid(acc3[i] + fr[36 + i]);
                        const T diff = a3 - fr[20 + i];
                        const bool ok = c < C && live;
                        d3v[i] = ok ? diff * (a3 * ((T)1 - a3)) : (T)0;
                        lsum += ok ? diff * diff : (T)0;
                        d3s[c * kLd + n] = d3v[i];
                    }
                    if (n < kP2Ts) store4<T>(bufs.d3 + (size_t)(s0 + n) * kP2C, lane, d3v);
                    // delta_2 = (W_2^T delta_3) (*) a_2 (1 - a_2)                                                              rcn.rs:305-309
                    acc_t ad2 = acc_t{0, 0, 0, 0};
                    {
                        T dvv[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) dvv[ks] = d3s[(4 * ks + g4) * kLd + n];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) ad2 = Mfma16<T>::mfma(fr[32 + ks], dvv[ks], ad2);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int h2 = Mfma16<T>::row(lane, i);
                        dv[i] = ad2[i] * (a2r[i] * ((T)1 - a2r[i]));                 // a2r is zero outside the live units / samples
                        d2s[h2 * kLd + n] = dv[i];
                    }
                }
                if (n < kP2Ts) store4<T>(bufs.d2 + (size_t)(s0 + n) * kP2C, lane, dv);
                // delta_1 = (W_1^T delta_2) (*) a_1 (1 - a_1)                                                          rcn.rs:305-309
                {
                    T dvv[4], av[kMtp][4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) dvv[ks] = d2s[(4 * ks + g4) * kLd + n];
#pragma unroll
                    for (int t = 0; t < kMtp; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) av[t][i] = a1s[(t * 16 + Mfma16<T>::row(lane, i)) * kLd + (n & 7)];
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) {
                        acc_t ad = acc_t{0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) ad = Mfma16<T>::mfma(fr[8 + t * 4 + ks], dvv[ks], ad);
                        acc_t o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = ad[i] * (av[t][i] * ((T)1 - av[t][i]));
                        if (n < kP2Ts) store4<T>(bufs.d1 + (size_t)(s0 + n) * kP2H + t * 16, lane, o);
                    }
                }
                lsum = wave_sum_lane0(lsum);
                if (lane == 0) bufs.loss[w] = lsum;
                XSTAMP(4);
                xcd_drain();                                              // only this wave stored
                if (lane == 0) xcd_flag(bufs.flagB + w * kXcdFlagStride, tag);
                XSTAMP(5);
            }
        }

        // =============================================================== second half of step j: needs every sample group's outputs
        if (is_a || is_t) {
            if (wave == 1) {
                const unsigned long long mB = xcd_wait(bufs.flagB, NS, nullptr, 0, tag, timeout, bufs.errd);
                if (mB != 0ull && lane == 0) {
                    s_abort = 1;
                    xcd_raise(err, bufs.errd, 1u, kXcdSiteDeltaFlag, w, j, launch_id, mB, tag, dp_rank, dp_world, xsel, NW);
                }
            }
        }
        if (wave == 1) XSTAMP(6);
        __syncthreads();                                                   // (also orders wave 0's LDS scratch against the next step's)
        if (s_abort) return;
        if (wave == 1) XSTAMP(7);

        if (is_a) {
            // ---- U: dW_0[:, slice pair] = sum_s delta_1[s] (x) x_s[slice pair]; W_0 <- W_0 - (eta/B) dW_0            rcn.rs:310, 214
            // delta_1 of the whole batch into LDS first: 32 KB as 16-byte L1-bypassing loads, 4 per thread (as 4-byte loads straight
            // into MFMA operands it was 16 per lane and the slower part of this phase)
            // (SB -- f64 at 256 samples: delta_1 in DH = 2 parts of BH samples, each staged and contracted in turn into the same accumulators,
            // and the next batch fetched into registers under the first part's MFMAs)
            acc_t acc0 = acc_t{0, 0, 0, 0}, acc1 = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int dh = 0; dh < DH; ++dh) {
            {
                constexpr int DR = (BH * 8 + kXcdThreads - 1) / kXcdThreads;
                vec4 dv[DR];
#pragma unroll
                for (int r = 0; r < DR; ++r) dv[r] = xcd_ld4<T>(r_d1, (dh * BH * 8 + tid + r * kXcdThreads) * VS);    // (past the buffer: zeros, not stored)
#pragma unroll
                for (int r = 0; r < DR; ++r) {
                    const int idx = tid + r * kXcdThreads;                 // vec4 index in [BH][8]
                    if (BH * 8 >= kXcdThreads || idx < BH * 8) *reinterpret_cast<vec4*>(d1s + (idx >> 3) * kXcdD1Ld + (idx & 7) * 4) = dv[r];
                }
            }
            __syncthreads();
            if (wave == 1 && dh == 0) XSTAMP(8);
            if constexpr (SB) { if (dh == 0 && pre) prefetch(j + 1); }
            // Four 16 x 16 tiles, 64 k-steps each: 256 MFMAs = 2048 cycles of the CU's four matrix pipes whichever way they are cut.
            // Waves 0..3 (one per SIMD) each run ONE tile over the WHOLE batch -- no K-split, so no cross-wave reduction, no partials
            // in LDS, no second barrier -- on two interleaved accumulators (a single chain would be paced by the 40-cycle dependent
            // latency instead of the 32-cycle issue rate).  Waves 4..7 have no arithmetic here; they move the prefetched batch into LDS.
            if (kh == 0) {
                const T* xb = xbuf + (SB ? (size_t)0 : (size_t)(j & 1) * kXs) + (size_t)usl * BT * 16 + (size_t)dh * BH * 16;
                constexpr int NQB = BH / 32;                               // blocks of eight k-steps (four samples each)
                // hand-pipelined: the operands of block qb + 1 are requested before the MFMAs of block qb issue, and fences keep the
                // scheduler from folding that back into load-wait-MFMA triples (which ran at ~100 cycles per MFMA instead of 32)
                const T* ap = d1s + g4 * kXcdD1Ld + umt * 16 + n;
                const T* bp = xb + g4 * 16 + n;
                T av[2][8], bv[2][8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { av[0][q] = ap[4 * q * kXcdD1Ld]; bv[0][q] = bp[4 * q * 16]; }
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    if (qb + 1 < NQB) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            av[(qb + 1) & 1][q] = ap[4 * (8 * (qb + 1) + q) * kXcdD1Ld];
                            bv[(qb + 1) & 1][q] = bp[4 * (8 * (qb + 1) + q) * 16];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc0 = Mfma16<T>::mfma(av[qb & 1][q], bv[qb & 1][q], acc0);
                        acc1 = Mfma16<T>::mfma(av[qb & 1][q + 1], bv[qb & 1][q + 1], acc1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (dh + 1 < DH) __syncthreads();                              // (every wave is past its reads of this part before the next one is staged)
            }
            if (kh == 0) {
                T gsum[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) gsum[i] = acc0[i] + acc1[i];
                if constexpr (DP) {
                    // this shard's four sums per lane meet the other ranks' at the slice pair's owner rank, the totals come back
                    // (dp_push.hpp: reduce-scatter + all-gather on pushed, self-validating words; rank-order sums)
                    const unsigned seq = dp.seq0 + (unsigned)j;
                    unsigned wo = woff0;                                     // opaque: the peers' addresses of these words are formed
                    asm volatile("" : "+v"(wo));                             // here, per step, instead of living in registers all loop long
                    const size_t i0 = (size_t)nd.w_off[0] + wo;
                    const bool gw[4] = {wvalid[0], wvalid[1], wvalid[2], wvalid[3]};
                    unsigned pmiss = 0u;
                    long long ph0 = 0;
                    if constexpr (PH) { if (tid == 0) ph0 = wall_clock64(); }
                    const bool ok = push_reduce4(dp.pd, w, seq, i0, gw, gsum, dp.ptimeout, &pmiss);
                    if constexpr (PH) { if (tid == 0) ph_x += wall_clock64() - ph0; }
                    if (!ok) {
                        s_abort = 1;
                        xcd_raise(err, bufs.errd, 1u, push_owner_of(w, dp_world) == dp_rank ? kXcdSitePushOwner : kXcdSitePushMember, w, j, launch_id,
                                  (unsigned long long)pmiss | ((unsigned long long)i0 << 32), seq, dp_rank, dp_world, xsel, NW);      // (high half: the lane's first parameter index)
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const T wn = wcur[i] - scale * gsum[i];
                    wcur[i] = wvalid[i] ? wn : (T)0;
                }
                store4<T>(wsl + (usl * 16 + n) * kP2H + umt * 16, lane, acc_t{wcur[0], wcur[1], wcur[2], wcur[3]});
                if (wave == 1) XSTAMP(9);
            }
            // (DP: a barrier that orders LDS only -- __syncthreads would also wait for the exchange's system-scope stores to reach
            // memory, 0.8 us on the step's critical path; they finish under the next forward and are covered by its drain)
            if constexpr (DP) xcd_barrier_lds(); else __syncthreads();
            if (DP && s_abort) return;
            if (wave == 1) XSTAMP(10);
            // the LDS buffer of batch j is free now (every wave is past its reads): the batch after next moves in; its first reader is
            // the forward of the NEXT iteration, several barriers away
            if (pre) {
                vec4* dst = reinterpret_cast<vec4*>(xbuf + (SB ? (size_t)0 : (size_t)(j & 1) * kXs));
#pragma unroll
                for (int r = 0; r < XR; ++r)
                    if (kXcdSl * BT * 4 >= kXcdThreads || tid + r * kXcdThreads < kXcdSl * BT * 4) dst[tid + r * kXcdThreads] = xr[r];
            }
            if constexpr (SB) __syncthreads();                             // (one buffer: the forward below is its first reader)
            if (more) {
                forward(j + 1);
                if (wave == 1) XSTAMP(11);
                if (wave == 0) XSTAMP(13);
                if (wave == 4) XSTAMP(14);
                if (wave == 7) XSTAMP(15);
                __syncthreads();
                if (tid == 0) xcd_flag(bufs.flagA + w * kXcdFlagStride, tag + 1);
                if (wave == 0) XSTAMP(12);
            }
        } else if (is_t) {
            // ---- tail tile e: e == 0 the bias column of W_0 (db_0 = sum_s delta_1); e >= 1 a 16-column tile of [W_1 | b_1]
            if (e == 0 && wave == 0 && (loss_dev || DP)) {
                // the batch's cost: the sample groups' NS (<= 32) shares, one per lane, summed by a fixed butterfly (every lane ends
                // with the same bits); lane 0 publishes.  One load latency, not NS of them, between flagB and this tile's gradient
                T t = lane < NS ? xcd_ld1<T>(r_loss, lane * ES) : (T)0;
#pragma unroll
                for (int sh = 16; sh >= 1; sh >>= 1) t += __shfl_xor(t, sh, 64);
                t *= loss_scale;
                if (lane == 0) {
                if constexpr (DP) {                                          // the global cost: every shard's share, element P of the exchange
                    const unsigned seq = dp.seq0 + (unsigned)j;
                    T g = t;
                    unsigned pmiss = 0u;
                    if (dp.pd.pd.world == 1 || push_all1(dp.pd, seq, (size_t)nd.P, t, dp.ptimeout, g, &pmiss)) t = g;
                    else { s_abort = 1; xcd_raise(err, bufs.errd, 1u, kXcdSitePushCost, w, j, launch_id, (unsigned long long)pmiss | ((unsigned long long)nd.P << 32), seq, dp_rank, dp_world, xsel, NW); }
                }
                if (loss_dev) loss_dev[j] = t;
                }
            }
            int jl, n0;
            tile_of(jl, n0);
            const int Kin = nd.dims[jl], M = nd.dims[jl + 1];
            const int c = n0 + n;                                          // this lane's column of [W | b]
            const auto r_act = e == 0 ? r_d1 : ((L3 && jl == 2) ? r_a2 : r_a1);    // A_prev: unused for the bias-only tile (every column >= Kin)
            const auto r_del = e == 0 ? r_d1 : ((L3 && jl == 2) ? r_d3 : r_d2);
            const int ldD = e == 0 ? kP2H : kP2C, ldA = (L3 && jl == 2) ? kP2C : kP2H;
            acc_t acc[kMtp];
#pragma unroll
            for (int t = 0; t < kMtp; ++t) acc[t] = acc_t{0, 0, 0, 0};
            {
                constexpr int NQ = BT / 32;                                // k-steps of four samples per wave: the batch over eight waves
                const int kc = wave * (BT >> 3);
                const int cc_ld = (e != 0 && c < Kin) ? c : 0;
                T bv[NQ], av[NQ][kMtp];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int s = kc + 4 * q + g4;
                    bv[q] = xcd_ld1<T>(r_act, (int)(((size_t)s * ldA + cc_ld) * ES));
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) {
                        const int row = t * 16 + n;
                        av[q][t] = xcd_ld1<T>(r_del, (int)(((size_t)s * ldD + (row < M ? row : M - 1)) * ES));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const T b = (e != 0 && c < Kin) ? bv[q] : (c == Kin ? (T)1 : (T)0);     // bias column: activation 1 (rcn.rs:302,309)
#pragma unroll
                    for (int t = 0; t < kMtp; ++t) acc[t] = Mfma16<T>::mfma(t * 16 + n < M ? av[q][t] : (T)0, b, acc[t]);
                }
            }
            store_partials<T>(red, wave, lane, acc);
            __syncthreads();
            {
                const int o = tid & 255, tcl = o >> 4, tml = o & 15, tm = mt * 16 + tml, cc = n0 + tcl;
                if (tvalid) {
                    T gsum = sum_partials<T>(red, mt, tcl, tml);
                    if constexpr (DP) {
                        const unsigned seq = dp.seq0 + (unsigned)j;
                        size_t tpo = tp;                                     // (opaque, as in the feature workers' exchange)
                        asm volatile("" : "+v"(tpo));
                        T g = gsum;
                        unsigned pmiss = 0u;
                        long long ph0 = 0;
                        if constexpr (PH) { if (tid == 0) ph0 = wall_clock64(); }
                        const bool pok = dp.pd.pd.world == 1 || push_all1(dp.pd, seq, tpo, gsum, dp.ptimeout, g, &pmiss);
                        if constexpr (PH) { if (tid == 0) ph_x += wall_clock64() - ph0; }
                        if (pok) gsum = g;
                        else { s_abort = 1; xcd_raise(err, bufs.errd, 1u, kXcdSitePushTail, w, j, launch_id, (unsigned long long)pmiss | ((unsigned long long)tpo << 32), seq, dp_rank, dp_world, xsel, NW); }
                    }
                    tcur -= scale * gsum;                                    // rcn.rs:214,221
                    int tmo = tm;                                            // opaque: the image addresses are recomputed per step, not
                    asm volatile("" : "+v"(tmo));                            // kept in two dozen register pairs across the whole loop
                    if (L3 && jl == 2) xcd_frag_scatter3(cc, tmo, Cm, tcur, bufs.fragimg);
                    else xcd_frag_scatter2(jl, cc, tmo, H, tcur, bufs.fragimg);
                }
            }
            xcd_drain();
            __syncthreads();
            if (DP && s_abort) return;
            if (more && tid == 0) xcd_flag(bufs.flagT + e * kXcdFlagStride, tag + 1);
            if (wave == 0) XSTAMP(12);
        }
    }

    if constexpr (PH) {
        if (tid == 0) {
            bufs.phase[w * 4 + 0] = ph_x;
            bufs.phase[w * 4 + 1] = wall_clock64() - ph_t0;
            bufs.phase[w * 4 + 2] = nb;
            bufs.phase[w * 4 + 3] = is_a ? (push_owner_of(w, dp_world) == dp_rank ? 1 : 2) : (is_t ? 3 : 0);
        }
    }
    // ---- all or nothing: a worker writes its parameters back only if EVERY worker finished the last step -- decided on ONE word that all
    // workers read the same way (xcd_commit): a wait that expires anywhere leaves the whole parameter vector as the launch found it,
    // and no worker that arrives later can write its slice under it
    {
        const unsigned tagD = tag0 + (unsigned)nb - 1u;
        __syncthreads();
        if (tid == 0) {
            if (late_closer) {                                               // (test hook: arrive after the others' waits have expired)
                const long long t0 = wall_clock64();
                while (wall_clock64() - t0 < 3 * timeout) __builtin_amdgcn_s_sleep(8);
            }
            xcd_flag(bufs.flagD + w * kXcdFlagStride, tagD);                  // (diagnostic only: who had arrived -- the host reads the table after a failure)
            bool gave_up;
            unsigned seen;
            const bool commit = xcd_commit(bufs.cw, NW, timeout, bufs.errd, gave_up, seen);
            if (!commit) {
                s_abort = 1;
                if (gave_up) xcd_raise(err, bufs.errd, 1u, kXcdSiteClosing, w, nb - 1, launch_id, (unsigned long long)(seen & 0xffffu), tagD, dp_rank, dp_world, xsel, NW);
            }
        }
        __syncthreads();
        if (s_abort) return;
        if (w == 0 && tid == 0) __hip_atomic_store(bufs.done, launch_id & 0x3fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- the slice pair of W_0 and the tail parameters go back to the parameter vector
    if (is_a) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wvalid[i]) W0[woff0 + RI * i] = wcur[i];
    } else if (is_t && tvalid) {
        params[tp] = tcur;
    }
    // (cross-check for the host, read after the stream has drained: this worker went through its write-back of this launch -- no drain, no
    // barrier here: a kernel does not end before its stores have, and the host looks only behind a synchronise)
    if (tid == 0) __hip_atomic_store(bufs.cdone + w, launch_id & 0x3fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// placement probe: the same grid and LDS footprint as k_xcd_epoch; block b reports (XCC_ID, CU slot) -- the host selects the
// resident kernel only if the 32 blocks with b % 8 == 0 report one XCC_ID
__global__ __launch_bounds__(kXcdThreads) void k_xcd_probe(unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        smem_dyn[0] = (unsigned char)id;
        out[blockIdx.x] = 0x100u | (id & 0xfu);
    }
}

}  // namespace rcn
