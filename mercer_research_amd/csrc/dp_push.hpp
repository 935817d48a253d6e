// dp_push.hpp -- the gradient exchange of the resident kernel's data-parallel form (dense_xcd.hpp, DP = true): a reduce-scatter and an
// all-gather on self-validating words, PUSHED over xGMI.
//
// What it replaces, and why.  Rounds 1-2 exchanged by PULL: a rank stored its tagged words {value, step} in its own memory and polled
// the same words of all seven peers over the links (dense_p2_dp.hpp: ll_store / ll_gather_sum4 -- still the exchange of the
// two-kernel pipeline).  Per step and rank that is 7 x P words of 8 bytes IN over seven links (1.3 MB for the default net: ~190 KB per
// link, 2.5 us at the links' ~77 GB/s per direction before any latency), and every poll is a full xGMI round trip.  Here:
//
//   reduce-scatter   every slice pair of W_0 has ONE owner rank (feature worker w -> rank w % world).  The other ranks push their four
//                    partial sums per lane straight into the owner's memory (posted stores: half a round trip), at the parameter's
//                    index in the row of the sending rank; the owner polls its OWN memory, adds the rows in rank order with its own
//                    value at its own place,
//   all-gather       and pushes the sum into every peer's memory, where the lane that needs it polls locally.
//
// Every rank applies the owner's bits, so replicas stay bit-identical without a broadcast, and at a group of one the sum is the
// rank's own value.  Bytes per step and rank: (world-1)/world x P words out in each phase -- 1.75 P words at eight ranks instead of 7 P
// (47 KB per link instead of 190 KB); latency: two posted-store hops plus two local polls instead of one round trip per poll.
// The tail parameters (b_0, [W_1 | b_1]: ~340 values) and the cost stay all-to-all, pushed: one hop, every rank adds all rows itself.
//
// A word is {value bits, step number} written by ONE 8-byte system-scope store: single-copy atomic, so a matching tag implies the
// value is this step's -- no flag, no fence.  Reuse needs no barrier: rows alternate by step parity; a sender overwrites the word of
// step s at step s + 2, which it reaches only after it has received the all-gather word of step s + 1 for the same parameter, which
// the owner sent only after it had consumed every row of step s + 1, hence of step s; the owner overwrites an all-gather word of step
// s at step s + 2 only after it has the member's partial of step s + 2, which the member sent after consuming the word of step s.
// Every poll is bounded by the wall clock (read lazily, see ll_gather_sum) and fails the launch instead of hanging it.
// rcn_hip_dp_init admits the form by its own known-answer exchange on exactly these primitives (k_push_selftest) and its own vote.
#pragma once

#include "dense_p2_dp.hpp"

namespace rcn {

// Region (u64 words) behind the older protocols' regions of every rank's exported buffer, at byte offset `off`:
//   rs[parity 2][source rank 8][stride]   partial sums pushed to this rank by `source`, at their parameter index
//   ag[parity 2][stride]                  rank-ordered sums pushed to this rank by the parameters' owners
struct PushDesc { P2PDesc pd; size_t stride; size_t off; };
inline size_t push_region_bytes(size_t stride) { return (size_t)2 * (kP2PMaxWorld + 1) * stride * sizeof(u64); }

__device__ inline u64* push_rs(const PushDesc& d, int at_rank, unsigned par, int src) {
    return reinterpret_cast<u64*>(reinterpret_cast<char*>(d.pd.buf[at_rank]) + d.off) + ((size_t)par * kP2PMaxWorld + (size_t)src) * d.stride;
}
__device__ inline u64* push_ag(const PushDesc& d, int at_rank, unsigned par) {
    return reinterpret_cast<u64*>(reinterpret_cast<char*>(d.pd.buf[at_rank]) + d.off) + ((size_t)2 * kP2PMaxWorld + par) * d.stride;
}
__device__ inline void push_store(u64* p, float v, unsigned seq) {
    unsigned b;
    __builtin_memcpy(&b, &v, 4);
    __hip_atomic_store(p, ((u64)seq << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline int push_owner_of(int worker, int world) { return worker % world; }

// OWNER of four consecutive parameters idx0 .. idx0 + 3 (the resident kernel's accumulator layout): collect the other ranks' rows out of
// this rank's own memory -- batches of four ranks, all missing words of a batch in flight together, the answers looked at afterwards
// (dense_p2_dp.hpp: why) -- add in rank order, then push the sums to every peer.  own[]: in this rank's partial sums, out the totals.
// *miss_out (nullable; written only when the wait expires): the ranks whose rows were still missing.
__device__ inline bool push_owner4(const PushDesc& d, unsigned seq, size_t idx0, const bool (&want)[4], float (&own)[4], long long timeout_ticks,
                                   unsigned* miss_out = nullptr) {
    const unsigned par = seq & 1u;
    const int world = d.pd.world, rank = d.pd.rank;
    const unsigned peers = ((1u << world) - 1u) & ~(1u << rank);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long long t0 = 0;
    unsigned it = 0;
#pragma unroll
    for (int q0 = 0; q0 < kP2PMaxWorld; q0 += 4) {
        if (q0 >= world) break;                                       // (uniform)
        float x[4][4];
        unsigned miss[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) miss[i] = want[i] ? (peers >> q0) & 15u : 0u;
        for (; (miss[0] | miss[1] | miss[2] | miss[3]) != 0u; ++it) {
            u64 w[4][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const u64* words = push_rs(d, rank, par, (q0 + qq) < world ? q0 + qq : 0) + idx0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((miss[i] >> qq) & 1u) w[i][qq] = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (((miss[i] >> qq) & 1u) && (unsigned)(w[i][qq] >> 32) == seq) {
                        const unsigned b = (unsigned)w[i][qq];
                        __builtin_memcpy(&x[i][qq], &b, 4);
                        miss[i] &= ~(1u << qq);
                    }
            if ((miss[0] | miss[1] | miss[2] | miss[3]) != 0u && (it & 31u) == 31u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > timeout_ticks) {
                    if (miss_out) *miss_out = (miss[0] | miss[1] | miss[2] | miss[3]) << q0;
                    return false;
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = q0 + qq;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += q < world ? (q == rank ? own[i] : x[i][qq]) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (want[i]) own[i] = acc[i];
#pragma unroll
    for (int q = 0; q < kP2PMaxWorld; ++q) {
        if (q >= world) break;                                        // (uniform)
        if (q == rank) continue;
        u64* dst = push_ag(d, q, par) + idx0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (want[i]) push_store(dst + i, own[i], seq);
    }
    return true;
}

// MEMBER (not the owner): push this rank's four partial sums into the owner's row for this rank, then wait for the totals in this rank's
// own memory.
__device__ inline bool push_member4(const PushDesc& d, int owner, unsigned seq, size_t idx0, const bool (&want)[4], float (&own)[4], long long timeout_ticks,
                                    unsigned* miss_out = nullptr) {
    const unsigned par = seq & 1u;
    {
        u64* dst = push_rs(d, owner, par, d.pd.rank) + idx0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (want[i]) push_store(dst + i, own[i], seq);
    }
    const u64* src = push_ag(d, d.pd.rank, par) + idx0;
    unsigned miss = (want[0] ? 1u : 0u) | (want[1] ? 2u : 0u) | (want[2] ? 4u : 0u) | (want[3] ? 8u : 0u);
    long long t0 = 0;
    for (unsigned it = 0; miss != 0u; ++it) {
        u64 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((miss >> i) & 1u) w[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (((miss >> i) & 1u) && (unsigned)(w[i] >> 32) == seq) {
                const unsigned b = (unsigned)w[i];
                __builtin_memcpy(&own[i], &b, 4);
                miss &= ~(1u << i);
            }
        if (miss != 0u && (it & 31u) == 31u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout_ticks) {
                if (miss_out) *miss_out = 1u << owner;                // (the totals come from the owner)
                return false;
            }
        }
    }
    return true;
}

// the two together, for the lanes of feature worker `worker`
__device__ inline bool push_reduce4(const PushDesc& d, int worker, unsigned seq, size_t idx0, const bool (&want)[4], float (&own)[4], long long timeout_ticks,
                                    unsigned* miss_out = nullptr) {
    if (d.pd.world == 1) return true;                                  // (0 + own = own: the single-GPU kernel's bits)
    const int owner = push_owner_of(worker, d.pd.world);              // (uniform per workgroup)
    return owner == d.pd.rank ? push_owner4(d, seq, idx0, want, own, timeout_ticks, miss_out) : push_member4(d, owner, seq, idx0, want, own, timeout_ticks, miss_out);
}

// ONE value, all-to-all: push it into every peer's row for this rank, collect the peers' out of this rank's own rows, add in rank order.
__device__ inline bool push_all1(const PushDesc& d, unsigned seq, size_t idx, float own, long long timeout_ticks, float& sum, unsigned* miss_out = nullptr) {
    const unsigned par = seq & 1u;
    const int world = d.pd.world, rank = d.pd.rank;
#pragma unroll
    for (int q = 0; q < kP2PMaxWorld; ++q) {
        if (q >= world) break;
        if (q != rank) push_store(push_rs(d, q, par, rank) + idx, own, seq);
    }
    float v[kP2PMaxWorld];
    unsigned ready = 1u << rank;
    const unsigned all = (1u << world) - 1u;
    long long t0 = 0;
    for (unsigned it = 0; ready != all; ++it) {
        u64 raw[kP2PMaxWorld];
        const unsigned miss = all & ~ready;
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q)
            if ((miss >> q) & 1u) raw[q] = __hip_atomic_load(push_rs(d, rank, par, q) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int q = 0; q < kP2PMaxWorld; ++q)
            if (((miss >> q) & 1u) && (unsigned)(raw[q] >> 32) == seq) {
                const unsigned b = (unsigned)raw[q];
                __builtin_memcpy(&v[q], &b, 4);
                ready |= 1u << q;
            }
        if (ready != all && (it & 31u) == 31u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout_ticks) {
                if (miss_out) *miss_out = all & ~ready;
                return false;
            }
        }
    }
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < kP2PMaxWorld; ++q) g += q < world ? (q == rank ? own : v[q]) : 0.f;
    sum = g;
    return true;
}

// Known-answer exchange on exactly these primitives (collective): the first half of the index range as the feature workers use it
// (four consecutive values per thread, one owner per 256-thread block), the second half all-to-all; integer patterns, exact sums.
__global__ __launch_bounds__(256) void k_push_selftest(PushDesc d, unsigned seq, unsigned* __restrict__ err, long long timeout_ticks,
                                                       unsigned* __restrict__ mismatches) {
    if (*err != 0u) return;
    const size_t half = (d.stride / 8) * 4;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t idx0 = 4 * t;
    if (idx0 + 3 < half) {
        float own[4];
        const bool want[4] = {true, true, true, (t & 7u) != 7u};        // (a value that is not exchanged, as padded rows are)
#pragma unroll
        for (int i = 0; i < 4; ++i) own[i] = p2p_pattern(d.pd.rank, seq, idx0 + i);
        if (!push_reduce4(d, (int)blockIdx.x, seq, idx0, want, own, timeout_ticks)) { *err = 1u + (unsigned)d.pd.rank; return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float expect = 0.f;
            for (int q = 0; q < d.pd.world; ++q) expect += p2p_pattern(q, seq, idx0 + i);
            if (!want[i]) expect = p2p_pattern(d.pd.rank, seq, idx0 + i);
            if (own[i] != expect) atomicAdd(mismatches, 1u);
        }
    }
    const size_t idx = half + t;
    if (idx < d.stride) {
        const float own = p2p_pattern(d.pd.rank, seq, idx);
        float got = own;
        if (d.pd.world > 1 && !push_all1(d, seq, idx, own, timeout_ticks, got)) { *err = 1u + (unsigned)d.pd.rank; return; }
        float expect = 0.f;
        for (int q = 0; q < d.pd.world; ++q) expect += p2p_pattern(q, seq, idx);
        if (got != expect) atomicAdd(mismatches, 1u);
    }
}

}  // namespace rcn
