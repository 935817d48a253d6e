// common.hpp -- shared definitions for the gfx950 rcn kernels (device + host side of the library).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rcn {

constexpr int kMaxLayers = 8;       // dense layers (feedforward_cfg.len()+1)
constexpr int kMaxConvPool = 8;     // conv/pool stack depth
constexpr int kWave = 64;           // CDNA wavefront
constexpr int kTileS = 16;          // samples per MFMA tile (N of v_mfma_*_16x16x4)
constexpr int kLd = 17;             // padded leading dimension of 16-wide LDS tiles (bank-conflict break)

// Dense network description handed to kernels by value.
// Flat parameter buffer: [W_0 | b_0 | W_1 | b_1 | ...], W_l column-major (rows = dims[l+1], cols = dims[l]),
// i.e. [W_l | b_l] is ONE column-major rows x (cols+1) matrix whose last column is the bias.
struct NetDesc {
    int L;                          // number of dense layers
    int dims[kMaxLayers + 1];       // dims[0] = F, dims[L] = classes
    int w_off[kMaxLayers];          // offset of W_l in the flat buffer (b_l follows at w_off + rows*cols)
    int P;                          // total parameter count
    int act_off[kMaxLayers + 1];    // prefix sums of dims[1..]: per-sample offset of layer l's vector in a
                                    // [sum d] row; saved activations / deltas are stored [layer][B][d_l] with
                                    // layer base = B * act_off[l]
    int tile_start[kMaxLayers + 1]; // wgrad grid: first workgroup of layer l (16-column tiles of [W_l|b_l])
};

// Conv/pool stack description for the fused feature kernel.
struct FeatDesc {
    int H, W;                       // input image rows / cols
    int n;                          // number of layers
    int kind[kMaxConvPool];         // 0 conv, 1 pool
    int arg[kMaxConvPool];          // padding (conv) / pooling (pool)
    int F;                          // flattened feature length
    int max_elems;                  // largest per-image element count of any stage (for LDS sizing)
};

template <typename T> struct Mfma16;

// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; D: col = l&15, row = 4*(l>>4) + i
template <> struct Mfma16<float> {
    using acc_t = __attribute__((ext_vector_type(4))) float;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static inline int row(int lane, int i) { return ((lane >> 4) << 2) + i; }
};

// v_mfma_f64_16x16x4_f64: same A/B maps; D: col = l&15, row = (l>>4) + 4*i   (NOT the f32 map)
template <> struct Mfma16<double> {
    using acc_t = __attribute__((ext_vector_type(4))) double;
    __device__ static inline acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static inline int row(int lane, int i) { return (lane >> 4) + (i << 2); }
};

// sigmoid as the reference writes it: 1 / (1 + E^(-x))   (rcn.rs:478-483)
__device__ inline float  sigmoid_ref(float x)  { return 1.0f / (1.0f + expf(-x)); }
__device__ inline double sigmoid_ref(double x) { return 1.0 / (1.0 + exp(-x)); }
// f32 only: hardware exp2 / rcp (v_exp_f32, v_rcp_f32; ~1 ulp each, i.e. ~1e-7 relative on the result, inside the
// 2e-6 activation tolerance of the f32 path) -- 4 instructions instead of ~35 on a path that is issue-bound.
// (__builtin_amdgcn_rcpf IS v_rcp_f32; HIP's __frcp_rn is the correctly rounded reciprocal, i.e. a ten-instruction
// IEEE division, which is what this used to compile to.)
__device__ inline float  sigmoid_fast(float x)  { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline double sigmoid_fast(double x) { return sigmoid_ref(x); }

// Sum over the wave, result in lane 0, in exactly the order of the shuffle-down tree
//     for (off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64)
// so the bits do not depend on which form runs.  f32: two lane swaps (v_permlane32_swap / v_permlane16_swap, new in
// gfx950) and four DPP row shifts -- six VALU instructions instead of six ds_bpermute round trips (~100 clocks each) at
// the very end of a kernel whose last wave is the critical path.
__device__ inline float wave_sum_lane0(float v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);     // lane i: v[i], v[i + 32]
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);         // lane i < 16: v[i], v[i + 16]
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));    // row_shl:8  lane i += lane i + 8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));
    return v;
}
__device__ inline double wave_sum_lane0(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- diagnostic build only (-DRCN_STAMPS, never shipped): per-workgroup phase timestamps (100 MHz s_memrealtime)
#ifdef RCN_STAMPS
__device__ unsigned long long g_rcn_stamps[2][512][16];
#define RCN_STAMP(kid, i)                                                                                                  \
    do {                                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 512) g_rcn_stamps[kid][blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define RCN_STAMP(kid, i) do { } while (0)
#endif

}  // namespace rcn
