// rcn_hipx_api.hip -- C ABI of include/rcn_hipx.h (Track X: trainable conv net; no reference counterpart) over convnet.hpp.
#include "../../include/rcn_hipx.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <tuple>
#include <vector>

#include "convnet.hpp"
#include "convnet_bf16.hpp"
#include "convnet_halo.hpp"
#include "convnet_halo_bf16.hpp"

using namespace rcnx;

namespace {

struct Buf {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Layer {
    int kind;
    int H, W, Cin;          // input of the layer
    int oH, oW, Cout;       // output (logical Cout)
    int CoutP;              // padded to a multiple of 32
    int K;                  // contraction length (3*3*Cin or Cin*H*W for dense)
    long long w_off = 0, b_off = 0;       // padded flat layout
    long long lw_off = 0, lb_off = 0;     // logical flat layout
    bool pool_follows = false;
    Buf out, idx, dout;     // activation (post-ReLU / pooled), pool arg-max, gradient wrt the layer's OUTPUT
    Buf slab;               // partial [W | b] tiles of the weight-gradient kernels (reduced for all layers at once: k_reduce_all)
    long long wbf_off = -1, wbb_off = -1;   // bf16 mode: this layer's transposed bf16 weight copies in net->wb16 (forward / input-gradient operand)
};

struct Key { const void* x; const void* y; int B; float lr; const void* loss;
    bool operator<(const Key& o) const { return std::tie(x, y, B, lr, loss) < std::tie(o.x, o.y, o.B, o.lr, o.loss); } };

}  // namespace

// Kernel-selection knobs, PER NET (round 4; before, most were read from the environment into function-local statics at first use:
// frozen process-wide, two nets of one process could not differ, and rcn_hipx_plan reported whatever the first call had latched).
// The environment variable of the same meaning only seeds the default when a net (or a plan) is created.
struct XOptions {
    int halo = 1;             // "halo"            RCN_HIPX_HALO            bf16: the LDS-tiled 3x3 kernels (0: implicit GEMM only)
    int bf16_pipe = 1;        // "bf16_pipe"       RCN_HIPX_BF16_PIPE       bf16: the software-pipelined LDS-tiled kernel (k_conv3x3_halo_bf16p)
    int bf16_1cb = 1;         // "bf16_1cb"        RCN_HIPX_BF16_1CB        bf16: the resident-weights form for 32-channel layers
    int bf16_rows16 = 0;      // "bf16_rows16"     RCN_HIPX_BF16_ROWS16     bf16 storage: 16 x 16 pixel blocks (two row groups per wave) where the map's height allows
    int halo_wgrad = 1;       // "halo_wgrad"      RCN_HIPX_HALO_WGRAD      the LDS-tiled weight-gradient kernels
    int fuse_pool_bwd = 1;    // "fuse_pool_bwd"   RCN_HIPX_FUSE_POOL_BWD   gradient kernels unpool while staging (no k_pool_bwd)
    int head = 1;             // "head"            RCN_HIPX_HEAD            the classifier head as one launch (k_head_f32)
    int xcd_remap = 0;        // "xcd_remap"       RCN_HIPX_XCD_REMAP       implicit-GEMM weight gradient: XCD-aware block order
    int pix_per_chunk = 0;    // "pix_per_chunk"   RCN_HIPX_PIX_PER_CHUNK   pixels per weight-gradient chunk (0: by the workgroup target)
    int wg_target = 4096;     // "wg_target"       RCN_HIPX_WG_TARGET       workgroups aimed at by the implicit-GEMM weight gradient
    int wgh_f32_target = 512; // "wgh_f32_target"  RCN_HIPX_WGH_F32_TARGET  ... by the fp32 LDS-tiled weight gradient
    int wgh_target = 256;     // "wgh_target"      RCN_HIPX_WGH_TARGET      ... by the bf16 LDS-tiled weight gradient
    int wgb_policy = 1;       // "wgb_policy"      RCN_HIPX_WGB_POLICY      bf16 implicit-GEMM weight gradient: narrower tiles / shorter chunks below 2 waves per SIMD
    int wgf_policy = 1;       // "wgf_policy"      RCN_HIPX_WGF_POLICY      fp32: the same
    int halo_f32_slots = 0;   // "halo_f32_slots"  RCN_HIPX_HALO_F32_SLOTS  resident workgroups assumed for the looping kernels (0: asked from the runtime)
};
namespace {
struct XOptDesc { const char* name; const char* env; int XOptions::*field; int lo, hi; };
const XOptDesc kXOptTable[] = {
    {"halo", "RCN_HIPX_HALO", &XOptions::halo, 0, 1},
    {"bf16_pipe", "RCN_HIPX_BF16_PIPE", &XOptions::bf16_pipe, 0, 1},
    {"bf16_1cb", "RCN_HIPX_BF16_1CB", &XOptions::bf16_1cb, 0, 1},
    {"bf16_rows16", "RCN_HIPX_BF16_ROWS16", &XOptions::bf16_rows16, 0, 1},
    {"halo_wgrad", "RCN_HIPX_HALO_WGRAD", &XOptions::halo_wgrad, 0, 1},
    {"fuse_pool_bwd", "RCN_HIPX_FUSE_POOL_BWD", &XOptions::fuse_pool_bwd, 0, 1},
    {"head", "RCN_HIPX_HEAD", &XOptions::head, 0, 1},
    {"xcd_remap", "RCN_HIPX_XCD_REMAP", &XOptions::xcd_remap, 0, 1},
    {"pix_per_chunk", "RCN_HIPX_PIX_PER_CHUNK", &XOptions::pix_per_chunk, 0, 1 << 20},
    {"wg_target", "RCN_HIPX_WG_TARGET", &XOptions::wg_target, 1, 1 << 20},
    {"wgh_f32_target", "RCN_HIPX_WGH_F32_TARGET", &XOptions::wgh_f32_target, 1, 1 << 20},
    {"wgh_target", "RCN_HIPX_WGH_TARGET", &XOptions::wgh_target, 1, 1 << 20},
    {"wgb_policy", "RCN_HIPX_WGB_POLICY", &XOptions::wgb_policy, 0, 1},
    {"wgf_policy", "RCN_HIPX_WGF_POLICY", &XOptions::wgf_policy, 0, 1},
    {"halo_f32_slots", "RCN_HIPX_HALO_F32_SLOTS", &XOptions::halo_f32_slots, 0, 1 << 20},
};
void seed_options(XOptions& o) {
    for (const XOptDesc& d : kXOptTable) {
        const char* e = std::getenv(d.env);
        if (!e || !*e) continue;
        const long long v = std::atoll(e);
        if (v >= d.lo && v <= d.hi) o.*(d.field) = (int)v;
    }
}
}  // namespace

struct rcn_hipx_net {
    XOptions opt;
    int device = 0, in_h = 0, in_w = 0, in_c = 0, max_batch = 0, classes = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    // The backward pass can run a layer's weight gradient on a second stream beside the input-gradient chain: the two only share dZ,
    // and each of these kernels leaves CUs idle while it ramps up and drains (rcn_hipx_set_overlap: 1 = every layer, 2 = dense layers).
    // OFF by default: measured no gain on the CIFAR step (fp32 0.420 / 0.418 / 0.418 ms for 0 / 1 / 2; 7 % slower while a reduction
    // launch per layer still ran on the second stream) -- the kernels are sized to fill the chip on their own.
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> events; size_t ev_next = 0;
    int overlap = 0;
    bool dry = false;                       // rcn_hipx_plan: walk a step's dispatch decisions, record what WOULD be launched, touch no device
    std::string plan;
    std::vector<Layer> L;
    long long n_pad = 0, n_log = 0;
    Buf params, wt, dz, loss_part, grad_tmp, dlogits, skbuf, wb;      // wt: tap-flipped transposed weights, laid out like params (w_off)
    Buf* slab_sel = nullptr;                // where the weight-gradient launch in progress puts its partial tiles (a layer's slab)
    ReduceJobs jobs{};                      // the step's pending slab reductions
    Buf wb16; PrepJobs prep{}; long long prep_blocks = 0;      // bf16 mode: every layer's bf16 operand copies, made by ONE launch per step
    int precision = RCN_HIPX_FP32;          // GEMM operand precision of forward / dgrad (rcn_hipx_set_precision)
    // RCN_HIPX_BF16_STORED: bf16 operands AND the convolutional stage's activations / gradients (every conv and pool layer's out and dout)
    // kept in memory as bf16.  precision == RCN_HIPX_BF16 then too: what is ROUNDED does not change, only where.
    bool store16 = false;
    int tiling = RCN_HIPX_TILING_AUTO;      // fp32 3x3 kernels: implicit GEMM only / by shape / LDS-tiled wherever they apply (rcn_hipx_set_tiling)
    std::map<Key, hipGraphExec_t> graphs;
    // the backward pass as a resumable walk (rcn_hipx_gradients_begin_dev / _bucket_dev: a data-parallel step whose all-reduce of one bucket
    // of layers overlaps the backward pass of the layers below it)
    struct BwState {
        std::vector<char> gated;            // layer's dout already holds dZ (ReLU gate applied by the producer)
        std::vector<PooledGrad> pooled;     // layer's dZ exists only at pooled resolution
        bool side_busy = false;
        int next = -1;                      // the next layer the walk handles (it runs from the last layer down to 0)
        const float* x = nullptr; int B = 0; float* grad = nullptr;
        int taken = 0;                      // buckets handed out since rcn_hipx_gradients_begin_dev
        std::vector<int> lo;                // bucket k ends with layer lo[k] (a layer with parameters; lo.back() == the first such layer)
        std::vector<long long> off, len;    // its slice of the padded flat gradient
    } bw;
    std::string err;
};

namespace {

int fail(rcn_hipx_net* n, int code, const std::string& m) { if (n) n->err = m; return code; }
// dry run (rcn_hipx_plan): note the launch that the code in front of this call has decided on and tell the caller to return
bool dry_note(rcn_hipx_net* n, const char* fmt, ...) {
    if (!n->dry) return false;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    n->plan += buf;
    n->plan += "\n";
    return true;
}
#define XTRY(net, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(net, e_ == hipErrorOutOfMemory ? -7 : -4, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define RTRY(expr) do { int s_ = (expr); if (s_ != 0) return s_; } while (0)

void drop_graphs(rcn_hipx_net* n);
// scratch buffers that captured graphs point into: one that grows moves, and every cached graph would replay on freed memory --
// drop them, they are re-captured on demand (sizes are settled by the eager step that precedes every capture)
hipError_t scratch_ensure(rcn_hipx_net* n, Buf& b, size_t bytes) {
    if (n->dry) return hipSuccess;
    const void* before = b.p;
    const hipError_t e = b.ensure(bytes);
    if (e == hipSuccess && before && b.p != before) drop_graphs(n);
    return e;
}

struct Dev { int prev = -1; explicit Dev(int d) { (void)hipGetDevice(&prev); if (prev != d) (void)hipSetDevice(d); else prev = -1; } ~Dev() { if (prev >= 0) (void)hipSetDevice(prev); } };

int grid1d(long long total, int block) { long long g = (total + block - 1) / block; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// Y = act(conv(X) + b) as implicit GEMM; `ks` = 1 or 3; epi 0 raw / 1 bias / 2 bias + relu.
// Few output tiles and a long contraction (the dense layers: M = batch) -> split-K over gridDim.z into raw partial tiles
// (slab `skbuf`) that k_splitk_epilogue sums in order.
// epi 4 (bias + ReLU + the following 2x2 max-pool, written to Y = pooled map and pool_idx) exists only in the LDS-tiled bf16 kernel:
// callers ask conv_pool_fusable() first
static bool halo_enabled(const rcn_hipx_net* n) { return n->opt.halo != 0; }
// RCN_HIPX_HALO_F32 only seeds a new net's tiling mode (rcn_hipx_create); rcn_hipx_set_tiling changes it per net
static int halo_f32_default() { const char* e = std::getenv("RCN_HIPX_HALO_F32"); const int v = e ? std::atoi(e) : 1; return v < 0 || v > 2 ? 1 : v; }

// workgroups of `kernel` (256 threads, static LDS only) the device holds at once: the grid of a kernel whose workgroups loop over
// work items.  Asked from the runtime once per kernel.
long long resident_slots(rcn_hipx_net* n, const void* kernel) {
    if (n->opt.halo_f32_slots > 0) return n->opt.halo_f32_slots;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, long long> cache;
    const std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, const void*> key{n->device, kernel};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, n->device) != hipSuccess || cus < 1) cus = 256;
    const long long v = (long long)per_cu * cus;
    cache.emplace(key, v);
    return v;
}

// split-K factor of the implicit-GEMM kernels: few output tiles and a long contraction
int splitk_z(long long M, int Cout, int bn, int nkt) {
    const long long tiles = ((M + kBM - 1) / kBM) * (Cout / bn);
    int Z = 1;
    if (tiles < 256 && nkt >= 8) { Z = (int)(512 / tiles); if (Z > nkt / 4) Z = nkt / 4; if (Z < 1) Z = 1; }
    // every partial is summed by ONE thread per element in k_splitk_epilogue: 392 of them on a 128 x 32 output (the 50176 -> 10 layer of the
    // 224 x 224 net) made that kernel 101 us for 4096 sums
    if (Z > 64) Z = 64;
    return Z;
}

// Pixel-block geometry of the fp32 LDS-tiled kernels for an H x W map: 8 x 16 blocks of one image, or 8 x 8 blocks of two images
// side by side where that wastes fewer MFMA rows (8-, 24-, 56-pixel-wide maps).  Not used when less than 70 % of a block's rows
// are real pixels (the implicit-GEMM kernels have no such waste).
struct HaloPlan { bool ok; int tw; };
HaloPlan halo_plan(const rcn_hipx_net* n, const ConvShape& s) {
    const double uh = (double)s.H / ((s.H + 7) / 8 * 8);
    const double u16 = (double)s.W / ((s.W + 15) / 16 * 16);
    const double u8 = (double)s.W / ((s.W + 7) / 8 * 8) * ((double)s.N / ((s.N + 1) / 2 * 2));
    const int tw = u8 > u16 ? 8 : 16;
    // the LDS-tiled kernels address with 32-bit element offsets
    const bool fits = (long long)(s.N + 1) * s.H * s.W * (s.Cin > s.Cout ? s.Cin : s.Cout) < 0x7fffffffLL;
    return HaloPlan{fits && (n->tiling == RCN_HIPX_TILING_LDS || uh * (tw == 8 ? u8 : u16) >= 0.7), tw};
}
// the first layer's own kernels (k_conv1_*_f32): 1 or 3 input channels
bool conv1_f32_shape(const rcn_hipx_net* n, const ConvShape& s) { return n->tiling != RCN_HIPX_TILING_GEMM && (s.Cin == 1 || s.Cin == 3) && s.Cout % 32 == 0 && halo_plan(n, s).ok; }
bool conv_halo_f32_shape(const rcn_hipx_net* n, const ConvShape& s) { return n->tiling != RCN_HIPX_TILING_GEMM && s.Cin % 32 == 0 && s.Cout % 32 == 0 && halo_plan(n, s).ok; }
// split-K factor of a fp32 3x3 layer as launch_conv decides it (the LDS-tiled kernels have no split-K form)
int conv3_f32_z(const rcn_hipx_net* n, const ConvShape& s) {
    if (n->tiling == RCN_HIPX_TILING_LDS && conv_halo_f32_shape(n, s)) return 1;
    return splitk_z((long long)s.N * s.H * s.W, s.Cout, s.Cout % 64 == 0 ? 64 : 32, 9 * s.Cin / 32);
}

// does the 2x2 max-pool that follows this 3x3 convolution run in the convolution kernel's epilogue (EPI 4: LDS-tiled kernels only)?
bool conv_pool_fusable(const rcn_hipx_net* n, const ConvShape& s) {
    if (s.H % 2 || s.W % 2) return false;
    if (conv1_f32_shape(n, s)) return true;                            // the first layer's kernels serve both precisions
    if (n->precision == RCN_HIPX_BF16) return halo_enabled(n) && (s.Cin == 32 || s.Cin % 64 == 0);
    return conv_halo_f32_shape(n, s);
}

// can the LDS-tiled kernel run this 3x3 convolution (as launch_conv would decide)?  Mirrors launch_conv's split-K rule.
bool conv_halo_runs(const rcn_hipx_net* n, const ConvShape& s) {
    const long long M = (long long)s.N * s.H * s.W;
    if (n->precision != RCN_HIPX_BF16) return conv_halo_f32_shape(n, s) && conv3_f32_z(n, s) == 1;
    if (!halo_enabled(n) || !(s.Cin == 32 || s.Cin % 64 == 0) || s.Cout % 32) return false;
    const int bn = s.Cout % 128 == 0 ? 128 : s.Cout % 64 == 0 ? 64 : 32;
    return n->store16 || splitk_z(M, s.Cout, bn, 9 * s.Cin / 32) == 1;
}

// Storage of a launch's tensors (RCN_HIPX_BF16_STORED): x16 -- the input X (and a pooled-resolution input) is bf16; y16 -- the output Y and
// epilogue 3's gate tensor (both belong to the layer below in the input-gradient pass) are bf16.  The host passes every tensor as
// float*; the launch sites cast.
struct Store { bool x16 = false, y16 = false; };
// layer i's out / dout (i < 0: the net's input, always fp32)
bool stage16(const rcn_hipx_net* n, int i) { return n->store16 && i >= 0 && (n->L[i].kind == RCN_HIPX_CONV3X3_RELU || n->L[i].kind == RCN_HIPX_MAXPOOL2); }
static const char* kStoreGap = "bf16 storage (RCN_HIPX_BF16_STORED) covers nets whose convolutions run on the LDS-tiled kernels with fused pooling: this layer does not (rcn_hipx_plan shows the kernels chosen)";

int launch_conv(rcn_hipx_net* n, const float* X, const float* Wk, const float* bias, float* Y, ConvShape s, int ks, int epi, uint8_t* pool_idx = nullptr,
                const PooledGrad* pin = nullptr, bool force_fp32 = false, const __bf16* wb_ready = nullptr, Store st = Store{}) {
    const long long M = (long long)s.N * s.H * s.W;
    const bool smallc = ks * ks * s.Cin <= 32 && s.Cin % 32 != 0;   // the per-element gather loader: only where a k-tile is not 32 whole channels
    if (!smallc && s.Cin % 32) return fail(n, -3, "input channels must be a multiple of 32 (or the whole 3x3xCin patch <= 32)");
    if (s.Cout % 32) return fail(n, -3, "output channels must be a multiple of 32");
    // bf16 mode rounds the operands of every GEMM EXCEPT the first layer's (its whole 3 x 3 x Cin patch is one k-block: nothing of the
    // MFMA rate to gain, and its own fp32 kernels are the fast ones) and the fused classifier head's (head_fusable)
    const bool bf16 = n->precision == RCN_HIPX_BF16 && !(ks == 3 && smallc) && !force_fp32;
    const int bn = (bf16 && s.Cout % 128 == 0) ? 128 : (s.Cout % 64 == 0) ? 64 : 32;
    const int nkt = smallc ? 1 : ks * ks * s.Cin / 32;
    // (bf16 storage: a 3x3 layer never splits K -- whatever the batch size, it stays on the LDS-tiled kernels, the ones that take bf16 tensors)
    const int Z = (epi == 4 || (n->store16 && ks == 3)) ? 1 : (!bf16 && ks == 3 && !smallc) ? conv3_f32_z(n, s) : splitk_z(M, s.Cout, bn, nkt);
    float* out = Y;
    int kepi = epi;
    if (Z > 1) {
        XTRY(n, scratch_ensure(n, n->skbuf, (size_t)Z * M * s.Cout * sizeof(float)));
        out = (float*)n->skbuf.p; kepi = 0;
    }
    const dim3 grid((unsigned)((M + kBM - 1) / kBM), (unsigned)(s.Cout / bn), (unsigned)Z);
    if (bf16) {
        // operands rounded to bf16: the weights once here, transposed to [Cout][Kp]; the activations inside the kernel
        const int K = ks * ks * s.Cin, Kp = (K + 31) / 32 * 32;
        const __bf16* WB = wb_ready;                                    // made for all layers at the start of the step (prep_bf16_weights)
        if (!WB) {
            XTRY(n, scratch_ensure(n, n->wb, (size_t)s.Cout * Kp * sizeof(__bf16)));
            if (!n->dry) hipLaunchKernelGGL(k_prep_weights_bf16, dim3(grid1d((long long)s.Cout * Kp, 256)), dim3(256), 0, n->stream, Wk, K, s.Cout, (__bf16*)n->wb.p, Kp);
            WB = (const __bf16*)n->wb.p;
        }
        // thin 3x3 layers: the LDS-tiled kernel (one halo per 8x16 output block serves all nine taps)
        if (epi == 4 && !(ks == 3 && conv_pool_fusable(n, s))) return fail(n, -3, "internal: fused conv+pool epilogue requested for a layer the LDS-tiled kernel does not cover");
        if (halo_enabled(n) && ks == 3 && !smallc && Z == 1 && (s.Cin == 32 || s.Cin % 64 == 0)) {
            const int hbn = (s.Cout % 64 == 0) ? 64 : 32;
            const int tw = (s.W + kHaloTW - 1) / kHaloTW, th = (s.H + kHaloTH - 1) / kHaloTH;
            const dim3 hgrid((unsigned)(tw * th * s.N), (unsigned)(s.Cout / hbn));
            const PooledGrad pg = pin ? *pin : PooledGrad{nullptr, nullptr, nullptr};
            const int pipe = n->opt.bf16_pipe;
            if (pipe && (long long)(s.N + 1) * s.H * s.W * (s.Cin > s.Cout ? s.Cin : s.Cout) < 0x7fffffffLL) {
                // the pipelined form (convnet_halo_bf16.hpp): work items (pixel block, column block) on a resident grid, operands loaded a
                // phase ahead; same LDS images, rounding and MFMA order as k_conv3x3_halo_bf16 below
                const long long items = (long long)tw * th * s.N * (s.Cout / hbn);
#define HBP_LAUNCH_T(CI_, BN_, EPI_, PIN_, TS_) do { const long long slots = resident_slots(n, (const void*)k_conv3x3_halo_bf16p<CI_, BN_, EPI_, PIN_, TS_>); \
                hipLaunchKernelGGL((k_conv3x3_halo_bf16p<CI_, BN_, EPI_, PIN_, TS_>), dim3((unsigned)(items < slots ? items : slots)), dim3(kThreads), 0, n->stream, (const TS_*)X, WB, bias, (TS_*)out, s, tw, th, (int)items, pool_idx, \
                                   PooledGradT<TS_>{(const TS_*)pg.dP, (const TS_*)pg.P, pg.idx}); } while (0)
#define HBP_LAUNCH(CI_, BN_, EPI_, PIN_) do { if (st.x16) HBP_LAUNCH_T(CI_, BN_, EPI_, PIN_, __bf16); else HBP_LAUNCH_T(CI_, BN_, EPI_, PIN_, float); } while (0)
#define HBP_EPI(CI_, BN_) do { if (pin) { if (kepi == 3) HBP_LAUNCH(CI_, BN_, 3, true); else if (kepi == 0) HBP_LAUNCH(CI_, BN_, 0, true); else return fail(n, -3, "internal: pooled-resolution input with a forward epilogue"); } \
                               else if (kepi == 0) HBP_LAUNCH(CI_, BN_, 0, false); else if (kepi == 1) HBP_LAUNCH(CI_, BN_, 1, false); else if (kepi == 2) HBP_LAUNCH(CI_, BN_, 2, false); \
                               else if (kepi == 3) HBP_LAUNCH(CI_, BN_, 3, false); else HBP_LAUNCH(CI_, BN_, 4, false); } while (0)
#define HB1_LAUNCH_T(BN_, EPI_, PIN_, TS_) do { const long long slots = resident_slots(n, (const void*)k_conv3x3_halo_bf16_1cb<BN_, EPI_, PIN_, TS_>); \
                hipLaunchKernelGGL((k_conv3x3_halo_bf16_1cb<BN_, EPI_, PIN_, TS_>), dim3((unsigned)(items < slots ? items : slots)), dim3(kThreads), 0, n->stream, (const TS_*)X, WB, bias, (TS_*)out, s, tw, th, (int)items, pool_idx, \
                                   PooledGradT<TS_>{(const TS_*)pg.dP, (const TS_*)pg.P, pg.idx}); } while (0)
#define HB1_LAUNCH(BN_, EPI_, PIN_) do { if (st.x16) HB1_LAUNCH_T(BN_, EPI_, PIN_, __bf16); else HB1_LAUNCH_T(BN_, EPI_, PIN_, float); } while (0)
#define HB1_EPI(BN_) do { if (pin) { if (kepi == 3) HB1_LAUNCH(BN_, 3, true); else if (kepi == 0) HB1_LAUNCH(BN_, 0, true); else return fail(n, -3, "internal: pooled-resolution input with a forward epilogue"); } \
                          else if (kepi == 0) HB1_LAUNCH(BN_, 0, false); else if (kepi == 1) HB1_LAUNCH(BN_, 1, false); else if (kepi == 2) HB1_LAUNCH(BN_, 2, false); \
                          else if (kepi == 3) HB1_LAUNCH(BN_, 3, false); else HB1_LAUNCH(BN_, 4, false); } while (0)
                const int onecb = n->opt.bf16_1cb;
                if (st.x16 != st.y16) return fail(n, -3, kStoreGap);
                // 16 x 16 pixel blocks (k_conv3x3_halo_bf16p<..., MG = 2>): bf16 tensors, 64-wide column blocks, a height that 16-row blocks
                // cover with no more padding than 8-row blocks do, and still at least one item per resident workgroup
                const int th2 = (s.H + 15) / 16;
                const long long items2 = (long long)tw * th2 * s.N * (s.Cout / hbn);
                const bool mg2 = n->opt.bf16_rows16 && st.x16 && hbn == 64 && !(s.Cin == 32 && onecb && hbn == 32) && th2 * 16 == th * kHaloTH && items2 >= 512;
                if (mg2 && dry_note(n, "  conv3x3 %dx%dx%d->%d epi %d%s: k_conv3x3_halo_bf16p (16 x 16 pixel blocks), %lld items", s.H, s.W, s.Cin, s.Cout, kepi, pin ? " pooled-in" : "", items2)) return 0;
                if (mg2) {
#define HB2_LAUNCH(CI_, EPI_, PIN_) do { const long long slots = resident_slots(n, (const void*)k_conv3x3_halo_bf16p<CI_, 64, EPI_, PIN_, __bf16, 2>); \
                    hipLaunchKernelGGL((k_conv3x3_halo_bf16p<CI_, 64, EPI_, PIN_, __bf16, 2>), dim3((unsigned)(items2 < slots ? items2 : slots)), dim3(kThreads), 0, n->stream, (const __bf16*)X, WB, bias, (__bf16*)out, s, tw, th2, (int)items2, pool_idx, \
                                       PooledGradT<__bf16>{(const __bf16*)pg.dP, (const __bf16*)pg.P, pg.idx}); } while (0)
#define HB2_EPI(CI_) do { if (pin) { if (kepi == 3) HB2_LAUNCH(CI_, 3, true); else if (kepi == 0) HB2_LAUNCH(CI_, 0, true); else return fail(n, -3, "internal: pooled-resolution input with a forward epilogue"); } \
                          else if (kepi == 0) HB2_LAUNCH(CI_, 0, false); else if (kepi == 1) HB2_LAUNCH(CI_, 1, false); else if (kepi == 2) HB2_LAUNCH(CI_, 2, false); \
                          else if (kepi == 3) HB2_LAUNCH(CI_, 3, false); else HB2_LAUNCH(CI_, 4, false); } while (0)
                    if (s.Cin == 32) HB2_EPI(32); else HB2_EPI(64);
#undef HB2_EPI
#undef HB2_LAUNCH
                    XTRY(n, hipGetLastError());
                    return 0;
                }
                if (items <= 0x7fffffffLL && dry_note(n, "  %s %dx%dx%d->%d epi %d%s: %s, %lld items", ks == 3 ? "conv3x3" : "dense", s.H, s.W, s.Cin, s.Cout, kepi, pin ? " pooled-in" : "",
                                                      (s.Cin == 32 && onecb && hbn == 32) ? "k_conv3x3_halo_bf16_1cb<32>" : "k_conv3x3_halo_bf16p", items)) return 0;
                if (items <= 0x7fffffffLL) {
                    // one channel block and one 32-wide column tile: all nine taps' weights stay in LDS (with a 64-wide tile the 46 KB of weights
                    // cost a third workgroup per CU: measured 190 vs 158 us on the 32 -> 64 layer of the 224 x 224 net)
                    if (s.Cin == 32 && onecb && hbn == 32) HB1_EPI(32);
                    else if (s.Cin == 32) { if (hbn == 64) HBP_EPI(32, 64); else HBP_EPI(32, 32); }
                    else { if (hbn == 64) HBP_EPI(64, 64); else HBP_EPI(64, 32); }
                    XTRY(n, hipGetLastError());
                    return 0;
                }
#undef HB1_EPI
#undef HB1_LAUNCH
#undef HB1_LAUNCH_T
#undef HBP_EPI
#undef HBP_LAUNCH
#undef HBP_LAUNCH_T
            }
            if (st.x16 || st.y16) return fail(n, -3, kStoreGap);
#define HALO_CASE(CI_, BN_, EPI_) do { if (pin) hipLaunchKernelGGL((k_conv3x3_halo_bf16<CI_, BN_, EPI_, true>), hgrid, dim3(kThreads), 0, n->stream, X, WB, bias, out, s, tw, th, pool_idx, pg); \
                                       else hipLaunchKernelGGL((k_conv3x3_halo_bf16<CI_, BN_, EPI_, false>), hgrid, dim3(kThreads), 0, n->stream, X, WB, bias, out, s, tw, th, pool_idx, pg); } while (0)
#define HALO_EPI(CI_, BN_) do { if (kepi == 0) HALO_CASE(CI_, BN_, 0); else if (kepi == 1) HALO_CASE(CI_, BN_, 1); else if (kepi == 2) HALO_CASE(CI_, BN_, 2); else if (kepi == 3) HALO_CASE(CI_, BN_, 3); else HALO_CASE(CI_, BN_, 4); } while (0)
            if (dry_note(n, "  conv3x3 %dx%dx%d->%d epi %d%s: k_conv3x3_halo_bf16", s.H, s.W, s.Cin, s.Cout, kepi, pin ? " pooled-in" : "")) return 0;
            if (s.Cin == 32) { if (hbn == 64) HALO_EPI(32, 64); else HALO_EPI(32, 32); }
            else { if (hbn == 64) HALO_EPI(64, 64); else HALO_EPI(64, 32); }
#undef HALO_EPI
#undef HALO_CASE
            XTRY(n, hipGetLastError());
            return 0;
        }
        if (pin) return fail(n, -3, "internal: pooled-resolution input requested for a layer the LDS-tiled kernel does not cover");
        if (st.x16 || st.y16) {
            // the dense layer on top of the convolutional stage: its forward pass and weight gradient READ a bf16 map, its input gradient
            // WRITES one (gated by the map, epilogue 3, or raw into a pooled gradient, epilogue 0).  A split-K launch leaves float partials.
            if (ks != 1 || smallc || (st.x16 && st.y16)) return fail(n, -3, kStoreGap);
            const bool y16k = st.y16 && Z == 1;                     // the kernel itself writes the bf16 tensor
#define CONVS_CASE(BN_, EPI_) do { if (st.x16) hipLaunchKernelGGL((k_conv_fwd_bf16<1, false, BN_, EPI_, __bf16, float>), grid, dim3(kThreads), 0, n->stream, (const __bf16*)X, WB, bias, out, s); \
                                   else if (y16k) hipLaunchKernelGGL((k_conv_fwd_bf16<1, false, BN_, EPI_, float, __bf16>), grid, dim3(kThreads), 0, n->stream, X, WB, bias, (__bf16*)out, s); \
                                   else hipLaunchKernelGGL((k_conv_fwd_bf16<1, false, BN_, EPI_>), grid, dim3(kThreads), 0, n->stream, X, WB, bias, out, s); } while (0)
#define CONVS_EPI(BN_) do { if (kepi == 0) CONVS_CASE(BN_, 0); else if (kepi == 1) CONVS_CASE(BN_, 1); else if (kepi == 2) CONVS_CASE(BN_, 2); else CONVS_CASE(BN_, 3); } while (0)
            if (dry_note(n, "  dense %d->%d epi %d: k_conv_fwd_bf16<1, tile, %d>%s, %s", s.Cin, s.Cout, epi, bn, Z > 1 ? (" split-K " + std::to_string(Z) + " + k_splitk_epilogue").c_str() : "",
                         st.x16 ? "bf16 input map" : "bf16 output map")) return 0;
            if (bn == 128) CONVS_EPI(128); else if (bn == 64) CONVS_EPI(64); else CONVS_EPI(32);
#undef CONVS_EPI
#undef CONVS_CASE
            XTRY(n, hipGetLastError());
            if (Z > 1) {
                if (st.y16) hipLaunchKernelGGL(k_splitk_epilogue<__bf16>, dim3(grid1d(M * s.Cout, 256)), dim3(256), 0, n->stream, (const float*)n->skbuf.p, bias, (__bf16*)Y, M * s.Cout, s.Cout, Z, epi);
                else hipLaunchKernelGGL(k_splitk_epilogue<float>, dim3(grid1d(M * s.Cout, 256)), dim3(256), 0, n->stream, (const float*)n->skbuf.p, bias, Y, M * s.Cout, s.Cout, Z, epi);
                XTRY(n, hipGetLastError());
            }
            return 0;
        }
#define CONVB_CASE(KS_, SM_, BN_, EPI_) hipLaunchKernelGGL((k_conv_fwd_bf16<KS_, SM_, BN_, EPI_>), grid, dim3(kThreads), 0, n->stream, X, WB, bias, out, s)
#define CONVB_EPI(KS_, SM_, BN_) do { if (kepi == 0) CONVB_CASE(KS_, SM_, BN_, 0); else if (kepi == 1) CONVB_CASE(KS_, SM_, BN_, 1); else if (kepi == 2) CONVB_CASE(KS_, SM_, BN_, 2); else CONVB_CASE(KS_, SM_, BN_, 3); } while (0)
#define CONVB_BN(KS_, SM_) do { if (bn == 128) CONVB_EPI(KS_, SM_, 128); else if (bn == 64) CONVB_EPI(KS_, SM_, 64); else CONVB_EPI(KS_, SM_, 32); } while (0)
        if (dry_note(n, "  %s %dx%dx%d->%d epi %d: k_conv_fwd_bf16<%d, %s, %d>%s", ks == 3 ? "conv3x3" : "dense", s.H, s.W, s.Cin, s.Cout, epi, ks, smallc ? "gather" : "tile", bn,
                     Z > 1 ? (" split-K " + std::to_string(Z) + " + k_splitk_epilogue").c_str() : "")) return 0;
        if (ks == 3) { if (smallc) CONVB_BN(3, true); else CONVB_BN(3, false); }
        else { if (smallc) CONVB_BN(1, true); else CONVB_BN(1, false); }
#undef CONVB_BN
#undef CONVB_EPI
#undef CONVB_CASE
    } else {
        if (epi == 4 && !(ks == 3 && conv_pool_fusable(n, s))) return fail(n, -3, "internal: fused conv+pool epilogue requested for a layer the LDS-tiled kernel does not cover");
        if (st.x16 || (st.y16 && !(ks == 3 && smallc && (epi == 2 || epi == 4) && conv1_f32_shape(n, s)))) return fail(n, -3, kStoreGap);
        if (ks == 3 && smallc && (epi == 2 || epi == 4) && conv1_f32_shape(n, s)) {
            // first layer (convnet_halo.hpp): weights in registers, the block's input halo in LDS
            const int tw = halo_plan(n, s).tw, nimg = 16 / tw;
            const int tiles_w = (s.W + tw - 1) / tw, tiles_h = (s.H + 7) / 8;
            const long long items = (long long)tiles_w * tiles_h * ((s.N + nimg - 1) / nimg) * (s.Cout / 32);
            if (items > 0x7fffffffLL) return fail(n, -3, "too many pixel blocks in one layer");
#define C1_LAUNCH_T(CIN_, TW_, EPI_, TY_) do { const long long slots = resident_slots(n, (const void*)k_conv1_fwd_f32<CIN_, TW_, EPI_, TY_>); \
            hipLaunchKernelGGL((k_conv1_fwd_f32<CIN_, TW_, EPI_, TY_>), dim3((unsigned)(items < slots ? items : slots)), dim3(kThreads), 0, n->stream, X, Wk, bias, (TY_*)Y, s, tiles_w, tiles_h, (int)items, pool_idx); } while (0)
#define C1_LAUNCH(CIN_, TW_, EPI_) do { if (st.y16) C1_LAUNCH_T(CIN_, TW_, EPI_, __bf16); else C1_LAUNCH_T(CIN_, TW_, EPI_, float); } while (0)
#define C1_EPI(CIN_, TW_) do { if (epi == 4) C1_LAUNCH(CIN_, TW_, 4); else C1_LAUNCH(CIN_, TW_, 2); } while (0)
            if (dry_note(n, "  conv3x3 %dx%dx%d->%d epi %d: k_conv1_fwd_f32<%d, %d>, %lld items", s.H, s.W, s.Cin, s.Cout, epi, s.Cin, tw, items)) return 0;
            if (s.Cin == 3) { if (tw == 16) C1_EPI(3, 16); else C1_EPI(3, 8); }
            else { if (tw == 16) C1_EPI(1, 16); else C1_EPI(1, 8); }
#undef C1_EPI
#undef C1_LAUNCH
#undef C1_LAUNCH_T
            XTRY(n, hipGetLastError());
            return 0;
        }
        if (ks == 3 && !smallc && Z == 1 && conv_halo_f32_shape(n, s)) {
            // LDS-tiled (convnet_halo.hpp): one staged halo per block of 128 output pixels serves all nine taps
            const int tw = halo_plan(n, s).tw, nimg = 16 / tw;
            const int tiles_w = (s.W + tw - 1) / tw, tiles_h = (s.H + 7) / 8;
            // work items = (pixel block, bn-wide column block); at most as many workgroups as the chip holds at once (three per CU), each
            // taking items blockIdx.x, + gridDim.x, ...
            const long long items = (long long)tiles_w * tiles_h * ((s.N + nimg - 1) / nimg) * (s.Cout / bn);
            if (items > 0x7fffffffLL) return fail(n, -3, "too many pixel blocks in one layer");
            const PooledGrad pg = pin ? *pin : PooledGrad{nullptr, nullptr, nullptr};
#define HF_LAUNCH(TW_, BN_, EPI_, PIN_) do { const long long slots = resident_slots(n, (const void*)k_conv3x3_halo_f32<TW_, BN_, EPI_, PIN_>); \
            hipLaunchKernelGGL((k_conv3x3_halo_f32<TW_, BN_, EPI_, PIN_>), dim3((unsigned)(items < slots ? items : slots)), dim3(kThreads), 0, n->stream, X, Wk, bias, out, s, tiles_w, tiles_h, (int)items, pool_idx, pg); } while (0)
            // a pooled-resolution input only occurs in the input-gradient pass (EPI 0 / 3)
#define HF_EPI(TW_, BN_) do { if (pin) { if (kepi == 3) HF_LAUNCH(TW_, BN_, 3, true); else if (kepi == 0) HF_LAUNCH(TW_, BN_, 0, true); else return fail(n, -3, "internal: pooled-resolution input with a forward epilogue"); } \
                              else if (kepi == 0) HF_LAUNCH(TW_, BN_, 0, false); else if (kepi == 1) HF_LAUNCH(TW_, BN_, 1, false); else if (kepi == 2) HF_LAUNCH(TW_, BN_, 2, false); \
                              else if (kepi == 3) HF_LAUNCH(TW_, BN_, 3, false); else HF_LAUNCH(TW_, BN_, 4, false); } while (0)
            if (dry_note(n, "  conv3x3 %dx%dx%d->%d epi %d%s: k_conv3x3_halo_f32<%d, %d>, %lld items", s.H, s.W, s.Cin, s.Cout, kepi, pin ? " pooled-in" : "", tw, bn, items)) return 0;
            if (tw == 16) { if (bn == 64) HF_EPI(16, 64); else HF_EPI(16, 32); }
            else { if (bn == 64) HF_EPI(8, 64); else HF_EPI(8, 32); }
#undef HF_EPI
#undef HF_LAUNCH
            XTRY(n, hipGetLastError());
            return 0;
        }
        if (pin) return fail(n, -3, "internal: pooled-resolution input requested for a layer the LDS-tiled kernel does not cover");
#define CONV_CASE(KS_, SM_, BN_, EPI_) hipLaunchKernelGGL((k_conv_fwd<KS_, SM_, BN_, EPI_>), grid, dim3(kThreads), 0, n->stream, X, Wk, bias, out, s)
#define CONV_EPI(KS_, SM_, BN_) do { if (kepi == 0) CONV_CASE(KS_, SM_, BN_, 0); else if (kepi == 1) CONV_CASE(KS_, SM_, BN_, 1); else if (kepi == 2) CONV_CASE(KS_, SM_, BN_, 2); else CONV_CASE(KS_, SM_, BN_, 3); } while (0)
#define CONV_BN(KS_, SM_) do { if (bn == 64) CONV_EPI(KS_, SM_, 64); else CONV_EPI(KS_, SM_, 32); } while (0)
        if (dry_note(n, "  %s %dx%dx%d->%d epi %d: k_conv_fwd<%d, %s, %d>%s", ks == 3 ? "conv3x3" : "dense", s.H, s.W, s.Cin, s.Cout, epi, ks, smallc ? "gather" : "tile", bn,
                     Z > 1 ? (" split-K " + std::to_string(Z) + " + k_splitk_epilogue").c_str() : "")) return 0;
        if (ks == 3) { if (smallc) CONV_BN(3, true); else CONV_BN(3, false); }
        else { if (smallc) CONV_BN(1, true); else CONV_BN(1, false); }
#undef CONV_BN
#undef CONV_EPI
#undef CONV_CASE
    }
    XTRY(n, hipGetLastError());
    if (Z > 1) {
        hipLaunchKernelGGL(k_splitk_epilogue<float>, dim3(grid1d(M * s.Cout, 256)), dim3(256), 0, n->stream, (const float*)n->skbuf.p, bias, Y, M * s.Cout, s.Cout, Z, epi);
        XTRY(n, hipGetLastError());
    }
    return 0;
}

static int xcd_remap(const rcn_hipx_net* n) { return n->opt.xcd_remap; }
// Pixels per weight-gradient chunk.  Every chunk costs one (K+1) x Cout partial tile written to the slab and read back by
// k_reduce_all, and a chunk is worked on by `tiles` workgroups (k-blocks x n-tiles), so the chunk size aims at a
// total number of workgroups -- wide layers need few chunks -- with 1024 pixels as the floor (measured best on the small
// CIFAR / MNIST nets, where parallelism is what matters).
static int pix_per_chunk(const rcn_hipx_net* n, long long M, long long tiles) {
    const int v = n->opt.pix_per_chunk >= 128 ? n->opt.pix_per_chunk / 128 * 128 : 0;
    if (v) return v;
    const long long target = n->opt.wg_target;
    long long pix = (M * tiles / target + 127) / 128 * 128;
    if (pix < 1024) pix = 1024;
    if (pix > 32768) pix = 32768;
    return (int)pix;
}
#define kPixPerChunk (pix_per_chunk(n, M, (long long)(smallc ? 1 : K / 32) * (s.Cout / bn)))

static bool wgrad_halo_on(const rcn_hipx_net* n) { return n->opt.halo_wgrad != 0; }
bool wgrad_halo_f32_runs(const rcn_hipx_net* n, const ConvShape& s, int ks) { return ks == 3 && ((ks * ks * s.Cin > 32 && conv_halo_f32_shape(n, s)) || conv1_f32_shape(n, s)); }
bool wgrad_halo_runs(const rcn_hipx_net* n, const ConvShape& s, int ks) {
    if (ks == 3 && ks * ks * s.Cin <= 32 && conv1_f32_shape(n, s)) return wgrad_halo_on(n);      // first layer: fp32 kernels in either precision
    if (n->precision != RCN_HIPX_BF16) return wgrad_halo_on(n) && wgrad_halo_f32_runs(n, s, ks);
    return n->precision == RCN_HIPX_BF16 && wgrad_halo_on(n) && ks == 3 && ks * ks * s.Cin > 32 && (s.Cin == 32 || s.Cin % 64 == 0) && s.H >= kHaloTH / 2 && s.W >= kHaloTW / 2;
}

// st.x16: X is a bf16 tensor; st.y16: dZ (and a pooled-resolution dZ) is
int launch_wgrad(rcn_hipx_net* n, const float* X, const float* dZ, ConvShape s, int ks, int* chunks_out, const PooledGrad* pdz = nullptr, Store st = Store{}) {
    const long long M = (long long)s.N * s.H * s.W;
    const int K = ks * ks * s.Cin;
    const bool smallc = K <= 32 && s.Cin % 32 != 0;
    const int bn = (s.Cout % 64 == 0) ? 64 : 32, bn0 = bn;
    if (M > 0x7fff0000LL) return fail(n, -3, "too many output pixels in one layer (N*H*W must stay below 2^31)");
    const int chunks = (int)((M + kPixPerChunk - 1) / kPixPerChunk), chunks0 = chunks;
    XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)chunks * (K + 1) * s.Cout * sizeof(float)));
    if (pdz && !wgrad_halo_runs(n, s, ks)) return fail(n, -3, "internal: pooled-resolution dZ requested for a layer the LDS-tiled weight-gradient kernel does not cover");
    if (wgrad_halo_runs(n, s, ks) && smallc) {
        // first layer (convnet_halo.hpp): one 32 x 32 tile (rows = patch entries) per (co block, chunk of pixel blocks)
        const int tw = halo_plan(n, s).tw, nimg = 16 / tw;
        const int tiles_w = (s.W + tw - 1) / tw, tiles_h = (s.H + 7) / 8;
        const long long blocks = (long long)tiles_w * tiles_h * ((s.N + nimg - 1) / nimg);
        long long want = (1024 + s.Cout / 32 - 1) / (s.Cout / 32);
        if (want > blocks) want = blocks;
        const int bpc = (int)((blocks + want - 1) / want);
        const int hchunks = (int)((blocks + bpc - 1) / bpc);
        XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)hchunks * (K + 1) * s.Cout * sizeof(float)));
        const dim3 hgrid((unsigned)(s.Cout / 32), (unsigned)hchunks);
        const PooledGrad pg = pdz ? *pdz : PooledGrad{nullptr, nullptr, nullptr};
#define W1_CASE_T(CIN_, TW_, TD_) do { const PooledGradT<TD_> pgt{(const TD_*)pg.dP, (const TD_*)pg.P, pg.idx}; \
                                if (pdz) hipLaunchKernelGGL((k_conv1_wgrad_f32<CIN_, TW_, true, TD_>), hgrid, dim3(kThreads), 0, n->stream, X, (const TD_*)dZ, (float*)(*n->slab_sel).p, s, tiles_w, tiles_h, bpc, pgt); \
                                else hipLaunchKernelGGL((k_conv1_wgrad_f32<CIN_, TW_, false, TD_>), hgrid, dim3(kThreads), 0, n->stream, X, (const TD_*)dZ, (float*)(*n->slab_sel).p, s, tiles_w, tiles_h, bpc, pgt); } while (0)
#define W1_CASE(CIN_, TW_) do { if (st.y16) W1_CASE_T(CIN_, TW_, __bf16); else W1_CASE_T(CIN_, TW_, float); } while (0)
        if (st.x16) return fail(n, -3, kStoreGap);
        if (dry_note(n, "  wgrad conv3x3 %dx%dx%d->%d%s: k_conv1_wgrad_f32<%d, %d>, %d chunks", s.H, s.W, s.Cin, s.Cout, pdz ? " pooled-dZ" : "", s.Cin, tw, hchunks)) { *chunks_out = hchunks; return 0; }
        if (s.Cin == 3) { if (tw == 16) W1_CASE(3, 16); else W1_CASE(3, 8); }
        else { if (tw == 16) W1_CASE(1, 16); else W1_CASE(1, 8); }
#undef W1_CASE
#undef W1_CASE_T
        XTRY(n, hipGetLastError());
        *chunks_out = hchunks;
        return 0;
    }
    if ((st.x16 || st.y16) && n->precision != RCN_HIPX_BF16) return fail(n, -3, kStoreGap);
    if (n->precision != RCN_HIPX_BF16 && wgrad_halo_runs(n, s, ks)) {
        // fp32 LDS-tiled (convnet_halo.hpp): workgroup = (32 input channels, 32 output channels, chunk of pixel blocks), all nine taps.
        // Every chunk costs one (K+1) x Cout partial written and read back by the reduce whatever the number of (ci, co) workgroups
        // that share it, so: as few chunks as fill the chip twice over.
        const int tw = halo_plan(n, s).tw, nimg = 16 / tw;
        const int tiles_w = (s.W + tw - 1) / tw, tiles_h = (s.H + 7) / 8;
        const long long blocks = (long long)tiles_w * tiles_h * ((s.N + nimg - 1) / nimg);
        const int target = n->opt.wgh_f32_target;
        const long long combos = (long long)(s.Cin / 32) * (s.Cout / 32);
        long long want = (target + combos - 1) / combos;
        if (want > blocks) want = blocks;
        if (want > 32768) want = 32768;
        const int bpc = (int)((blocks + want - 1) / want);
        const int hchunks = (int)((blocks + bpc - 1) / bpc);
        XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)hchunks * (K + 1) * s.Cout * sizeof(float)));
        const dim3 hgrid((unsigned)(s.Cin / 32), (unsigned)(s.Cout / 32), (unsigned)hchunks);
        const PooledGrad pg = pdz ? *pdz : PooledGrad{nullptr, nullptr, nullptr};
#define WGF_CASE(TW_) do { if (pdz) hipLaunchKernelGGL((k_wgrad3x3_halo_f32<TW_, true>), hgrid, dim3(kThreads), 0, n->stream, X, dZ, (float*)(*n->slab_sel).p, s, tiles_w, tiles_h, bpc, pg); \
                           else hipLaunchKernelGGL((k_wgrad3x3_halo_f32<TW_, false>), hgrid, dim3(kThreads), 0, n->stream, X, dZ, (float*)(*n->slab_sel).p, s, tiles_w, tiles_h, bpc, pg); } while (0)
        if (dry_note(n, "  wgrad conv3x3 %dx%dx%d->%d%s: k_wgrad3x3_halo_f32<%d>, %d chunks x %lld tiles", s.H, s.W, s.Cin, s.Cout, pdz ? " pooled-dZ" : "", tw, hchunks, combos)) { *chunks_out = hchunks; return 0; }
        if (tw == 16) WGF_CASE(16); else WGF_CASE(8);
#undef WGF_CASE
        XTRY(n, hipGetLastError());
        *chunks_out = hchunks;
        return 0;
    }
    if (wgrad_halo_runs(n, s, ks)) {
        // LDS-tiled: input halo + dZ block staged once per 8x16 pixel block, nine waves = nine filter taps (convnet_bf16.hpp)
        const int tw = (s.W + kHaloTW - 1) / kHaloTW, th = (s.H + kHaloTH - 1) / kHaloTH;
        const long long blocks = (long long)tw * th * s.N;
        const int hb = s.Cin == 32 ? 32 : 64, hbn = (s.Cout % 64 == 0) ? 64 : 32;
        // Pixel blocks per chunk: every chunk costs one (K+1) x Cout partial tile written and read back by the reduce, so aim at
        // `target` workgroups in total (tiles per chunk x chunks) rather than at a fixed chunk count -- wide layers have many
        // tiles per chunk and need few chunks.
        // (256 = one per CU: the 576-thread workgroup with its 64+ accumulator registers per wave is alone on its CU anyway, and every
        // chunk fewer is a partial [W | b] less to write and reduce: synth-224 bf16 5.20 ms at 512, 5.05 at 256, 5.49 at 384 -- 1.5 per CU)
        const int target = n->opt.wgh_target;
        const long long tiles = (long long)(s.Cin / hb) * (s.Cout / hbn);
        int bpc = (int)((blocks * tiles + target - 1) / target);
        if (bpc < 8) bpc = blocks < 8 ? (int)blocks : 8;
        const int hchunks = (int)((blocks + bpc - 1) / bpc);
        XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)hchunks * (K + 1) * s.Cout * sizeof(float)));
        const dim3 hgrid((unsigned)(s.Cin / hb), (unsigned)(s.Cout / hbn), (unsigned)hchunks);
        const PooledGrad pg = pdz ? *pdz : PooledGrad{nullptr, nullptr, nullptr};
#define WGH_CASE_T(CB_, BN_, TS_) do { const PooledGradT<TS_> pgt{(const TS_*)pg.dP, (const TS_*)pg.P, pg.idx}; \
                                if (pdz) hipLaunchKernelGGL((k_wgrad3x3_halo_bf16<CB_, BN_, true, TS_>), hgrid, dim3(kWgHaloThreads), 0, n->stream, (const TS_*)X, (const TS_*)dZ, (float*)(*n->slab_sel).p, s, tw, th, bpc, hchunks, pgt); \
                                else hipLaunchKernelGGL((k_wgrad3x3_halo_bf16<CB_, BN_, false, TS_>), hgrid, dim3(kWgHaloThreads), 0, n->stream, (const TS_*)X, (const TS_*)dZ, (float*)(*n->slab_sel).p, s, tw, th, bpc, hchunks, pgt); } while (0)
#define WGH_CASE(CB_, BN_) do { if (st.x16) WGH_CASE_T(CB_, BN_, __bf16); else WGH_CASE_T(CB_, BN_, float); } while (0)
        if (st.x16 != st.y16) return fail(n, -3, kStoreGap);
        if (dry_note(n, "  wgrad conv3x3 %dx%dx%d->%d%s: k_wgrad3x3_halo_bf16<%d, %d>, %d chunks x %lld tiles", s.H, s.W, s.Cin, s.Cout, pdz ? " pooled-dZ" : "", hb, hbn, hchunks, tiles)) { *chunks_out = hchunks; return 0; }
        if (hb == 32) { if (hbn == 64) WGH_CASE(32, 64); else WGH_CASE(32, 32); }
        else { if (hbn == 64) WGH_CASE(64, 64); else WGH_CASE(64, 32); }
#undef WGH_CASE
#undef WGH_CASE_T
        XTRY(n, hipGetLastError());
        *chunks_out = hchunks;
        return 0;
    }
    if (n->precision == RCN_HIPX_BF16 && !smallc) {
        // bf16 operands, transposed LDS reads (convnet_bf16.hpp); NKB waves per workgroup, one 32-row k-block each
        const int nkb = K / 32;
        const int nk = nkb % 4 == 0 ? 4 : nkb % 3 == 0 ? 3 : nkb % 2 == 0 ? 2 : 1;
        // A wave owns one 32-row k-block x bn columns over the chunk's pixels, so a dense layer behind a pooled map is FEW waves (MNIST shape
        // 3136 -> 128 at B = 4096: 98 x 2 x 4 chunks = 784 on the chip's 1024 SIMDs, 62 us for 7 us of traffic).  Below two waves per SIMD
        // take 32-wide column blocks (no more slab, X re-read from L2), below one per SIMD also shorter chunks (down to 256 pixels).
        const int policy = n->opt.wgb_policy;
        int bn = bn0, ppc = kPixPerChunk, chunks = chunks0;
        if (policy) {
            if ((long long)nkb * (s.Cout / bn) * chunks < 2048) bn = 32;
            while ((long long)nkb * (s.Cout / bn) * chunks < 1024 && ppc > 256) { ppc /= 2; chunks = (int)((M + ppc - 1) / ppc); }
            XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)chunks * (K + 1) * s.Cout * sizeof(float)));
        }
        const WgradGrid gdb{nkb / nk, s.Cout / bn, chunks, xcd_remap(n)};
        const dim3 gridb(gdb.launch_blocks());
#define WGB_CASE(KS_, BN_, NK_) do { if (st.x16) hipLaunchKernelGGL((k_conv_wgrad_bf16<KS_, BN_, NK_, __bf16>), gridb, dim3(64 * NK_), 0, n->stream, (const __bf16*)X, dZ, (float*)(*n->slab_sel).p, s, ppc, gdb); \
                                     else hipLaunchKernelGGL((k_conv_wgrad_bf16<KS_, BN_, NK_>), gridb, dim3(64 * NK_), 0, n->stream, X, dZ, (float*)(*n->slab_sel).p, s, ppc, gdb); } while (0)
        if (st.y16 || (st.x16 && ks != 1)) return fail(n, -3, kStoreGap);
#define WGB_NK(KS_, BN_) do { if (nk == 4) WGB_CASE(KS_, BN_, 4); else if (nk == 3) WGB_CASE(KS_, BN_, 3); else if (nk == 2) WGB_CASE(KS_, BN_, 2); else WGB_CASE(KS_, BN_, 1); } while (0)
#define WGB_BN(KS_) do { if (bn == 64) WGB_NK(KS_, 64); else WGB_NK(KS_, 32); } while (0)
        if (dry_note(n, "  wgrad %s %dx%dx%d->%d: k_conv_wgrad_bf16<%d, %d, %d>, %d chunks", ks == 3 ? "conv3x3" : "dense", s.H, s.W, s.Cin, s.Cout, ks, bn, nk, chunks)) { *chunks_out = chunks; return 0; }
        if (ks == 3) WGB_BN(3); else WGB_BN(1);
#undef WGB_BN
#undef WGB_NK
#undef WGB_CASE
        XTRY(n, hipGetLastError());
        *chunks_out = chunks;
        return 0;
    }
    // (as in the bf16 branch above: below two waves per SIMD the column blocks are 32 wide -- CIFAR net's 2048 -> 256 at B = 512: 256 -> 512
    // workgroups, step 0.419 -> 0.417 ms; MNIST shape B = 256: 0.165 -> 0.1625 ms)
    if (st.x16 || st.y16) return fail(n, -3, kStoreGap);
    const int f32_policy = n->opt.wgf_policy;
    const int bnf = (f32_policy && !smallc && 4LL * (K / 32) * (s.Cout / bn0) * chunks < 2048) ? 32 : bn0;
    const WgradGrid gd{smallc ? 1 : K / 32, s.Cout / bnf, chunks, xcd_remap(n)};
    const dim3 grid(gd.launch_blocks());
#define WG_CASE(KS_, SM_, BN_) hipLaunchKernelGGL((k_conv_wgrad<KS_, SM_, BN_>), grid, dim3(kThreads), 0, n->stream, X, dZ, (float*)(*n->slab_sel).p, s, kPixPerChunk, gd)
#define WG_BN(KS_, SM_) do { if (bnf == 64) WG_CASE(KS_, SM_, 64); else WG_CASE(KS_, SM_, 32); } while (0)
    if (dry_note(n, "  wgrad %s %dx%dx%d->%d: k_conv_wgrad<%d, %s, %d>, %d chunks", ks == 3 ? "conv3x3" : "dense", s.H, s.W, s.Cin, s.Cout, ks, smallc ? "gather" : "tile", bnf, chunks)) { *chunks_out = chunks; return 0; }
    if (ks == 3) { if (smallc) WG_BN(3, true); else WG_BN(3, false); }
    else { if (smallc) WG_BN(1, true); else WG_BN(1, false); }
#undef WG_BN
#undef WG_CASE
    XTRY(n, hipGetLastError());
    *chunks_out = chunks;
    return 0;
}

float* P(rcn_hipx_net* n, long long off) { return (float*)n->params.p + off; }

int ensure_batch(rcn_hipx_net* n, int B) {
    if (B < 1 || B > n->max_batch) return fail(n, -1, "batch size out of range for this net (max_batch)");
    return 0;
}

// `to` waits for everything enqueued on `from` so far (an event from the net's pool; inside a capture this is a graph dependency)
int stream_after(rcn_hipx_net* n, hipStream_t from, hipStream_t to) {
    if (n->ev_next == n->events.size()) {
        hipEvent_t e = nullptr;
        XTRY(n, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        n->events.push_back(e);
    }
    hipEvent_t ev = n->events[n->ev_next++];
    XTRY(n, hipEventRecord(ev, from));
    XTRY(n, hipStreamWaitEvent(to, ev, 0));
    return 0;
}

bool head_fusable(const rcn_hipx_net* n);

// bf16 mode: a layer's prepared bf16 operand copy (nullptr: none -- launch_conv then makes its own)
const __bf16* wb_of(const rcn_hipx_net* n, long long off) {
    return (n->precision == RCN_HIPX_BF16 && n->wb16.p && off >= 0) ? (const __bf16*)n->wb16.p + off : (const __bf16*)nullptr;
}

// bf16 mode, once per step / forward call: the transposed bf16 copies of every layer's weights -- [Cout][Kp] for the forward pass from W,
// [Cin][Kp'] for the input gradient from the flipped copy (which the update kernel keeps current) -- in ONE launch.  The first layer
// (fp32 kernels in either mode) has none.
int prep_bf16_weights(rcn_hipx_net* n) {
    if (n->precision != RCN_HIPX_BF16) return 0;
    if (dry_note(n, "  bf16 operand copies of every layer's weights, both orientations: k_prep_all_bf16 (one launch)")) return 0;
    if (!n->wb16.p) {
        long long off = 0;
        int q = 0;
        long long blocks = 0;
        std::vector<std::tuple<size_t, int, long long>> plan;      // (layer, orientation, offset)
        for (size_t i = 1; i < n->L.size(); ++i) {
            Layer& l = n->L[i];
            if (l.kind == RCN_HIPX_MAXPOOL2) continue;
            const int ks = l.kind == RCN_HIPX_CONV3X3_RELU ? 3 : 1;
            const long long cin = l.kind == RCN_HIPX_CONV3X3_RELU ? l.Cin : l.K;
            const long long Kf = (long long)ks * ks * cin, Kpf = (Kf + 31) / 32 * 32;                 // forward: [CoutP][Kpf]
            const long long Kb = (long long)ks * ks * l.CoutP, Kpb = (Kb + 31) / 32 * 32;              // input gradient: [cin][Kpb]
            l.wbf_off = off; off += (long long)l.CoutP * Kpf;
            l.wbb_off = off; off += cin * Kpb;
            off = (off + 63) / 64 * 64;
            q += 2;
        }
        if (q > kMaxPrepJobs) { for (Layer& l : n->L) l.wbf_off = l.wbb_off = -1; return 0; }       // more layers than one launch takes: per-layer copies as before
        XTRY(n, n->wb16.ensure((size_t)(off > 0 ? off : 1) * sizeof(__bf16)));
        n->prep.njobs = 0;
        for (size_t i = 1; i < n->L.size(); ++i) {
            Layer& l = n->L[i];
            if (l.kind == RCN_HIPX_MAXPOOL2) continue;
            const int ks = l.kind == RCN_HIPX_CONV3X3_RELU ? 3 : 1;
            const long long cin = l.kind == RCN_HIPX_CONV3X3_RELU ? l.Cin : l.K;
            const long long Kf = (long long)ks * ks * cin, Kpf = (Kf + 31) / 32 * 32, Kb = (long long)ks * ks * l.CoutP, Kpb = (Kb + 31) / 32 * 32;
            PrepJob& a = n->prep.j[n->prep.njobs++];
            a = PrepJob{(const float*)P(n, l.w_off), (__bf16*)n->wb16.p + l.wbf_off, (int)Kf, l.CoutP, (int)Kpf, (int)blocks};
            blocks += prep_job_blocks(l.CoutP, (int)Kpf);
            PrepJob& b = n->prep.j[n->prep.njobs++];
            b = PrepJob{(const float*)n->wt.p + l.w_off, (__bf16*)n->wb16.p + l.wbb_off, (int)Kb, (int)cin, (int)Kpb, (int)blocks};
            blocks += prep_job_blocks((int)cin, (int)Kpb);
        }
        n->prep_blocks = blocks;
    }
    if (n->prep.njobs) {
        hipLaunchKernelGGL(k_prep_all_bf16, dim3((unsigned)n->prep_blocks), dim3(256), 0, n->stream, n->prep);
        XTRY(n, hipGetLastError());
    }
    return 0;
}

// forward for batch B; returns pointer to logits (padded rows of CoutP)
int forward(rcn_hipx_net* n, const float* x, int B, size_t n_layers = (size_t)-1) {
    const float* cur = x;
    bool cur16 = false;                                  // `cur` is a bf16 tensor (RCN_HIPX_BF16_STORED)
    if (n_layers > n->L.size()) n_layers = n->L.size();
    for (size_t i = 0; i < n_layers; ++i) {
        Layer& l = n->L[i];
        if (l.kind == RCN_HIPX_MAXPOOL2) {
            if (n->store16) return fail(n, -3, kStoreGap);          // (a pool that no convolution kernel could fuse)
            const long long tot = (long long)B * l.oH * l.oW * (l.Cin / 4);
            if (dry_note(n, "  pool %dx%dx%d: k_pool_fwd", l.H, l.W, l.Cin)) { cur = (const float*)l.out.p; continue; }
            hipLaunchKernelGGL(k_pool_fwd, dim3(grid1d(tot, 256)), dim3(256), 0, n->stream, cur, (float*)l.out.p, (uint8_t*)l.idx.p, B, l.H, l.W, l.Cin);
            XTRY(n, hipGetLastError());
        } else if (l.kind == RCN_HIPX_CONV3X3_RELU) {
            const ConvShape cs{B, l.H, l.W, l.Cin, l.CoutP};
            if (l.pool_follows && conv_pool_fusable(n, cs)) {
                // the pool that follows runs in this kernel's epilogue: only the pooled map (and its arg-max image) is written
                Layer& pl = n->L[i + 1];
                RTRY(launch_conv(n, cur, P(n, l.w_off), P(n, l.b_off), (float*)pl.out.p, cs, 3, 4, (uint8_t*)pl.idx.p, nullptr, false, wb_of(n, l.wbf_off), Store{cur16, n->store16}));
                cur = (const float*)pl.out.p;
                cur16 = n->store16;
                ++i;
                continue;
            }
            RTRY(launch_conv(n, cur, P(n, l.w_off), P(n, l.b_off), (float*)l.out.p, cs, 3, 2, nullptr, nullptr, false, wb_of(n, l.wbf_off), Store{cur16, n->store16}));
            cur = (const float*)l.out.p;
            cur16 = n->store16;
            continue;
        } else {
            // (the logits layer of a head that training runs as k_head_f32 is fp32 here too: what bf16 mode rounds is a matter of shape)
            RTRY(launch_conv(n, cur, P(n, l.w_off), P(n, l.b_off), (float*)l.out.p, ConvShape{B, 1, 1, l.K, l.CoutP}, 1, l.kind == RCN_HIPX_DENSE_RELU ? 2 : 1, nullptr, nullptr,
                             i + 1 == n->L.size() && head_fusable(n), wb_of(n, l.wbf_off), Store{cur16, false}));
        }
        cur = (const float*)l.out.p;
        cur16 = false;
    }
    return 0;
}

// The slab of partial [W | b] tiles of layer i (chunks x (K + 1) x Cout, as the weight-gradient kernels leave it in the layer's slab)
// is queued for the step's ONE reduction launch (run_reduce_jobs): summed in chunk order, and either applied (p <- p - lr g, the flipped
// copy of the weights kept current) or written to grad.  [W | b] is contiguous (b_off == w_off + K * CoutP): one job finishes both.
int reduce_slab(rcn_hipx_net* n, size_t i, int chunks, int ks, const ConvShape& s, float lr, float* grad, bool apply) {
    (void)lr;
    Layer& l = n->L[i];
    if (n->jobs.njobs >= kMaxReduceJobs) return fail(n, -3, "too many layers with parameters for one reduction launch");
    ReduceJob& jb = n->jobs.j[n->jobs.njobs];
    jb.p = P(n, l.w_off);
    jb.grad = grad ? grad + l.w_off : (float*)nullptr;
    jb.slab = (const float*)l.slab.p;
    jb.flip = FlipSpec{(apply && i > 0) ? (float*)n->wt.p + l.w_off : (float*)nullptr, ks * ks, s.Cin, s.Cout};
    jb.n = ((long long)s.Cin * ks * ks + 1) * s.Cout;
    if (jb.n % 4 != 0 || l.w_off % 4 != 0) return fail(n, -3, "internal: a layer's [W | b] is not a whole number of 16-byte pieces");
    jb.chunks = chunks;
    int prev_end = 0;
    if (n->jobs.njobs) { const ReduceJob& pj = n->jobs.j[n->jobs.njobs - 1]; prev_end = pj.first_block + (int)((pj.n + reduce_job_elems(pj.chunks) - 1) / reduce_job_elems(pj.chunks)); }
    jb.first_block = prev_end;
    ++n->jobs.njobs;
    return 0;
}

int run_reduce_jobs(rcn_hipx_net* n, float lr, bool apply) {
    if (!n->jobs.njobs) return 0;
    const ReduceJob& last = n->jobs.j[n->jobs.njobs - 1];
    const long long blocks = last.first_block + (last.n + reduce_job_elems(last.chunks) - 1) / reduce_job_elems(last.chunks);
    n->jobs.lr = lr; n->jobs.apply = apply ? 1 : 0;
    if (dry_note(n, "  update: k_reduce_all, %d layers' slabs in one launch, %lld workgroups", n->jobs.njobs, blocks)) { n->jobs.njobs = 0; return 0; }
    hipLaunchKernelGGL(k_reduce_all, dim3((unsigned)blocks), dim3(kReduceThreads), 0, n->stream, n->jobs);
    XTRY(n, hipGetLastError());
    n->jobs.njobs = 0;
    return 0;
}

// backward from dlogits (already in L.back().dout); apply: update parameters with lr, else write gradients to grad (padded layout).
// backward_layers handles the layers hi .. lo (downwards) and queues their slab reductions; the state that travels from layer to layer
// lives in n->bw, so the walk can stop after a bucket of layers (rcn_hipx_gradients_bucket_dev) and go on later.
int backward_layers(rcn_hipx_net* n, const float* x, int B, float lr, float* grad, bool apply, int hi, int lo, bool allow_overlap) {
    std::vector<char>& gated = n->bw.gated;
    std::vector<PooledGrad>& pooled = n->bw.pooled;
    hipStream_t const main_s = n->stream;
    struct Restore { rcn_hipx_net* n; hipStream_t s; ~Restore() { n->stream = s; } } restore{n, main_s};   // launch_* enqueue on n->stream: it is switched below
    const bool ov = allow_overlap && n->overlap && n->side;
    bool& side_busy = n->bw.side_busy;
    for (int i = hi; i >= lo; --i) {
        Layer& l = n->L[i];
        const float* in = i == 0 ? x : (const float*)n->L[i - 1].out.p;
        float* din = i == 0 ? nullptr : n->dry ? reinterpret_cast<float*>(sizeof(float)) : (float*)n->L[i - 1].dout.p;     // (dry run: no buffers; non-null = "has an input gradient")
        if (l.kind == RCN_HIPX_MAXPOOL2) {
            // When the LDS-tiled kernels run both consumers of the convolution's dZ (its weight gradient, and its input gradient if
            // it has one), they rebuild dZ from (dP, P, arg-max) at pooled resolution while staging: no k_pool_bwd, no full-size dZ.
            const int fuse_on = n->opt.fuse_pool_bwd;
            const Layer& cl = n->L[i - 1];
            const ConvShape cs{B, cl.H, cl.W, cl.Cin, cl.CoutP};
            const bool fusable = fuse_on && wgrad_halo_runs(n, cs, 3) && (i - 1 == 0 || conv_halo_runs(n, ConvShape{B, cl.H, cl.W, cl.CoutP, cl.Cin}));
            if (fusable) {
                pooled[i - 1] = PooledGrad{(const float*)l.dout.p, (const float*)l.out.p, (const uint8_t*)l.idx.p};
                if (dry_note(n, "  pool-bwd %dx%dx%d: none (the convolution's gradient kernels unpool while staging)", l.H, l.W, l.Cin))
                    pooled[i - 1].dP = reinterpret_cast<const float*>(sizeof(float));      // (dry run: no buffers; any non-null marks "pooled")
                continue;
            }
            // gradient wrt the pool INPUT, with the preceding conv's ReLU mask folded in (pooled value > 0)
            const long long tot = (long long)B * l.oH * l.oW * (l.Cin / 4);
            if (dry_note(n, "  pool-bwd %dx%dx%d: k_pool_bwd", l.H, l.W, l.Cin)) continue;
            if (n->store16) hipLaunchKernelGGL(k_pool_bwd<__bf16>, dim3(grid1d(tot, 256)), dim3(256), 0, n->stream, (const __bf16*)l.dout.p, (const __bf16*)l.out.p, (const uint8_t*)l.idx.p,
                                               (__bf16*)din, B, l.H, l.W, l.Cin);
            else hipLaunchKernelGGL(k_pool_bwd<float>, dim3(grid1d(tot, 256)), dim3(256), 0, n->stream, (const float*)l.dout.p, (const float*)l.out.p, (const uint8_t*)l.idx.p,
                                    din, B, l.H, l.W, l.Cin);
            XTRY(n, hipGetLastError());
            continue;
        }
        const bool conv = l.kind == RCN_HIPX_CONV3X3_RELU;
        const int ks = conv ? 3 : 1;
        const ConvShape s = conv ? ConvShape{B, l.H, l.W, l.Cin, l.CoutP} : ConvShape{B, 1, 1, l.K, l.CoutP};
        const long long M = (long long)s.N * s.H * s.W;
        // dZ: gradient wrt the pre-activation
        const float* dZ = (const float*)l.dout.p;
        if (l.kind != RCN_HIPX_DENSE && !l.pool_follows && !gated[i]) {
            if (stage16(n, i)) return fail(n, -3, kStoreGap);
            XTRY(n, scratch_ensure(n, n->dz, (size_t)M * l.CoutP * sizeof(float)));
            if (!dry_note(n, "  relu-bwd: k_relu_bwd")) {
                hipLaunchKernelGGL(k_relu_bwd, dim3(grid1d(M * l.CoutP, 256)), dim3(256), 0, n->stream, (const float*)l.dout.p, (const float*)l.out.p, (float*)n->dz.p, M * l.CoutP);
                XTRY(n, hipGetLastError());
            }
            dZ = (const float*)n->dz.p;
        }
        // The weight gradient of this layer goes to the side stream: it needs dZ (ready on the main stream here) and the layer's
        // input.  Not when dZ sits in the shared scratch buffer, which the main stream reuses for the next layer.
        const bool on_side = ov && dZ != (const float*)n->dz.p && (n->overlap != 2 || !conv);     // 2: only the (latency-bound) dense layers
        if (on_side) { RTRY(stream_after(n, main_s, n->side)); side_busy = true; }
        // dgrad first (needs the weights BEFORE this step's update): dX = conv(dZ, flip(W)^T)
        if (din) {
            // the tap-flipped transposed weights are kept current by every kernel that writes a weight (FlipSpec, refresh_flipped)
            const float* wt = (const float*)n->wt.p + l.w_off;
            // the layer below is a ReLU layer feeding this one directly (no pool in between): gate the gradient with its output in
            // this kernel's epilogue, so that layer finds its dZ ready instead of running a k_relu_bwd pass over the tensor
            const Layer& below = n->L[i - 1];
            const bool gate = below.kind == RCN_HIPX_CONV3X3_RELU || below.kind == RCN_HIPX_DENSE_RELU;
            RTRY(launch_conv(n, dZ, wt, gate ? (const float*)below.out.p : nullptr, din, ConvShape{s.N, s.H, s.W, s.Cout, s.Cin}, ks, gate ? 3 : 0,
                             nullptr, pooled[i].dP ? &pooled[i] : nullptr, false, wb_of(n, l.wbb_off), Store{stage16(n, i), stage16(n, i - 1)}));
            gated[i - 1] = gate;
        }
        int chunks = 0;
        if (on_side) n->stream = n->side;
        n->slab_sel = &l.slab;
        RTRY(launch_wgrad(n, in, dZ, s, ks, &chunks, pooled[i].dP ? &pooled[i] : nullptr, Store{stage16(n, i - 1), stage16(n, i)}));
        RTRY(reduce_slab(n, (size_t)i, chunks, ks, s, lr, grad, apply));
        n->stream = main_s;
    }
    return 0;
}

void backward_reset(rcn_hipx_net* n, int first, bool first_gated) {
    n->bw.gated.assign(n->L.size(), 0);
    if (first >= 0 && first_gated) n->bw.gated[first] = 1;
    n->bw.pooled.assign(n->L.size(), PooledGrad{nullptr, nullptr, nullptr});
    n->bw.side_busy = false;
    n->ev_next = 0;
}

int backward(rcn_hipx_net* n, const float* x, int B, float lr, float* grad, bool apply, int first = -1, bool first_gated = false) {
    backward_reset(n, first, first_gated);
    RTRY(backward_layers(n, x, B, lr, grad, apply, first >= 0 ? first : (int)n->L.size() - 1, 0, true));
    if (n->bw.side_busy) RTRY(stream_after(n, n->side, n->stream));   // join: the step's next kernels (and an end of capture) find everything on the main stream
    return run_reduce_jobs(n, lr, apply);                             // every layer's slab in one launch; no weight was written before this point
}

// [partial sums of the loss, one per workgroup][counter of finished workgroups: zero between launches]
int ensure_loss_buf(rcn_hipx_net* n, unsigned** counter) {
    const void* before = n->loss_part.p;
    XTRY(n, scratch_ensure(n, n->loss_part, ((size_t)n->max_batch / 8 + 2) * sizeof(float)));
    *counter = (unsigned*)n->loss_part.p + (n->max_batch / 8 + 1);
    if (n->loss_part.p != before) XTRY(n, hipMemsetAsync(n->loss_part.p, 0, n->loss_part.cap, n->stream));
    return 0;
}

int loss_and_dlogits(rcn_hipx_net* n, const int32_t* labels, int B, float* loss_dev, bool want_grad) {
    Layer& l = n->L.back();
    const int blocks = (B + 7) / 8;                       // eight samples per workgroup
    if (dry_note(n, "  loss: k_softmax_ce, %d workgroups", blocks)) return 0;
    unsigned* counter = nullptr;
    RTRY(ensure_loss_buf(n, &counter));
    hipLaunchKernelGGL(k_softmax_ce, dim3(blocks), dim3(256), 0, n->stream, (const float*)l.out.p, labels, B, n->classes, l.CoutP, want_grad ? (float*)l.dout.p : (float*)nullptr,
                       (float*)n->loss_part.p, counter, 1.0f / (float)B, loss_dev);
    XTRY(n, hipGetLastError());
    return 0;
}

// Does the classifier head run as one launch (k_head_f32, fp32 arithmetic in either precision mode)?  Logits layer of at most 32 classes
// on a ReLU dense layer of at most 256 units.
bool head_fusable(const rcn_hipx_net* n) {
    const int on = n->opt.head;
    // (fp32: not in GEMM tiling mode, which keeps every layer on the implicit-GEMM kernels; bf16: by shape alone -- what the mode rounds
    // must not depend on a tiling switch)
    if (!on || (n->precision == RCN_HIPX_FP32 && n->tiling == RCN_HIPX_TILING_GEMM) || n->L.size() < 2) return false;
    const Layer& l = n->L.back();
    const Layer& b = n->L[n->L.size() - 2];
    return l.kind == RCN_HIPX_DENSE && l.CoutP == 32 && l.K % 32 == 0 && l.K <= 256 && b.kind == RCN_HIPX_DENSE_RELU && b.CoutP == l.K;
}

// logits, loss, d logits, gradient into the hidden layer below (gated by its ReLU) and the logits layer's weight-gradient partials
int launch_head(rcn_hipx_net* n, const int32_t* labels, int B, float* loss_dev, int* chunks_out) {
    Layer& l = n->L.back();
    Layer& b = n->L[n->L.size() - 2];
    const int F = l.K, blocks = (B + 31) / 32;
    if (dry_note(n, "  head %d -> %d classes (logits, softmax + cross-entropy, gradient into the hidden layer, weight-gradient partials): k_head_f32, %d workgroups", F, n->classes, blocks)) {
        *chunks_out = blocks;
        return 0;
    }
    XTRY(n, scratch_ensure(n, (*n->slab_sel), (size_t)blocks * (F + 1) * 32 * sizeof(float)));
    unsigned* counter = nullptr;
    RTRY(ensure_loss_buf(n, &counter));
    const size_t lds = ((size_t)32 * (F + 1) + (size_t)F * 32 + (size_t)32 * (F + 32) + 4 * 1024 + 1024 + 32 * 33) * sizeof(float);
    // more than 64 KB of dynamic LDS has to be asked for (at most 127 KB here: F <= 256)
    static const hipError_t attr = hipFuncSetAttribute((const void*)k_head_f32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    XTRY(n, attr);
    hipLaunchKernelGGL((k_head_f32<true>), dim3(blocks), dim3(kThreads), lds, n->stream, (const float*)b.out.p, (const float*)P(n, l.w_off), (const float*)n->wt.p + l.w_off,
                       (const float*)P(n, l.b_off), labels, B, F, n->classes, (float*)l.out.p, (float*)b.dout.p, (float*)(*n->slab_sel).p, (float*)n->loss_part.p, counter,
                       1.0f / (float)B, loss_dev);
    XTRY(n, hipGetLastError());
    *chunks_out = blocks;
    return 0;
}

// the tap-flipped transposed copy of every layer's weights that has an input gradient, from the current parameters
int refresh_flipped(rcn_hipx_net* n) {
    for (size_t i = 1; i < n->L.size(); ++i) {
        const Layer& l = n->L[i];
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        const int ks = l.kind == RCN_HIPX_CONV3X3_RELU ? 3 : 1;
        const int cin = l.kind == RCN_HIPX_CONV3X3_RELU ? l.Cin : l.K;
        const long long wn = (long long)l.K * l.CoutP;
        hipLaunchKernelGGL(k_flip_weights, dim3(grid1d(wn, 256)), dim3(256), 0, n->stream, (const float*)P(n, l.w_off), (float*)n->wt.p + l.w_off, ks, cin, l.CoutP);
    }
    XTRY(n, hipGetLastError());
    return 0;
}

int backward(rcn_hipx_net* n, const float* x, int B, float lr, float* grad, bool apply, int first, bool first_gated);
int forward(rcn_hipx_net* n, const float* x, int B, size_t n_layers);

// forward + loss + backward of one batch: parameters updated in place (apply) or gradients written to grad (padded layout)
int step_core(rcn_hipx_net* n, const float* x, const int32_t* labels, int B, float lr, float* grad, bool apply, float* loss_dev) {
    n->jobs.njobs = 0;
    RTRY(prep_bf16_weights(n));
    if (head_fusable(n)) {
        const int last = (int)n->L.size() - 1;
        RTRY(forward(n, x, B, (size_t)last));
        int chunks = 0;
        n->slab_sel = &n->L[last].slab;
        RTRY(launch_head(n, labels, B, loss_dev, &chunks));
        RTRY(reduce_slab(n, (size_t)last, chunks, 1, ConvShape{B, 1, 1, n->L[last].K, n->L[last].CoutP}, lr, grad, apply));
        return backward(n, x, B, lr, grad, apply, last - 1, true);
    }
    RTRY(forward(n, x, B, (size_t)-1));
    RTRY(loss_and_dlogits(n, labels, B, loss_dev, true));
    return backward(n, x, B, lr, grad, apply, -1, false);
}

void drop_graphs(rcn_hipx_net* n) { for (auto& kv : n->graphs) (void)hipGraphExecDestroy(kv.second); n->graphs.clear(); }

// ---- gradient buckets: the data-parallel step with its all-reduce overlapped with the backward pass -------------------------------------
// The layers with parameters, walked from the last to the first (the order the backward pass finishes them), are cut into buckets of at
// least min_bytes of gradient; the padded flat layout is in layer order, so a bucket is ONE contiguous slice.  Returns the bucket count.
int bucket_layout(rcn_hipx_net* n, long long min_bytes) {
    auto& bw = n->bw;
    bw.lo.clear(); bw.off.clear(); bw.len.clear();
    long long acc = 0, end = n->n_pad;
    int first_param = -1;
    for (int i = 0; i < (int)n->L.size(); ++i)
        if (n->L[i].kind != RCN_HIPX_MAXPOOL2) { first_param = i; break; }
    for (int i = (int)n->L.size() - 1; i >= 0; --i) {
        const Layer& l = n->L[i];
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        acc = (end - l.w_off) * (long long)sizeof(float);
        if (acc >= min_bytes || i == first_param) {
            bw.lo.push_back(i); bw.off.push_back(l.w_off); bw.len.push_back(end - l.w_off);
            end = l.w_off;
        }
    }
    // (the layers left over at the bottom are less than a bucket: they join the one above them)
    if (bw.lo.size() >= 2 && bw.len.back() * (long long)sizeof(float) < min_bytes) {
        const size_t m = bw.lo.size() - 1;
        bw.lo[m - 1] = bw.lo[m]; bw.off[m - 1] = bw.off[m]; bw.len[m - 1] += bw.len[m];
        bw.lo.pop_back(); bw.off.pop_back(); bw.len.pop_back();
    }
    return (int)bw.lo.size();
}

// forward, loss (and, where the classifier head is the fused one, its share of the backward pass): everything in front of bucket 0
int grad_begin(rcn_hipx_net* n, const float* x, const int32_t* labels, int B, float* grad, float* loss_dev, long long min_bytes) {
    n->jobs.njobs = 0;
    const int nb = bucket_layout(n, min_bytes);
    n->bw.x = x; n->bw.B = B; n->bw.grad = grad; n->bw.taken = 0;
    RTRY(prep_bf16_weights(n));
    if (head_fusable(n)) {
        const int last = (int)n->L.size() - 1;
        RTRY(forward(n, x, B, (size_t)last));
        int chunks = 0;
        n->slab_sel = &n->L[last].slab;
        RTRY(launch_head(n, labels, B, loss_dev, &chunks));
        RTRY(reduce_slab(n, (size_t)last, chunks, 1, ConvShape{B, 1, 1, n->L[last].K, n->L[last].CoutP}, 0.f, grad, false));
        backward_reset(n, last - 1, true);
        n->bw.next = last - 1;
    } else {
        RTRY(forward(n, x, B, (size_t)-1));
        RTRY(loss_and_dlogits(n, labels, B, loss_dev, true));
        backward_reset(n, -1, false);
        n->bw.next = (int)n->L.size() - 1;
    }
    return nb;
}

// the backward pass through bucket k's layers and the reduction of their slabs: grad[off, off + len) is final on the net's stream
int grad_bucket(rcn_hipx_net* n, int k, long long* off, long long* len) {
    auto& bw = n->bw;
    if (k < 0 || k >= (int)bw.lo.size()) return fail(n, -1, "gradients_bucket: no such bucket (rcn_hipx_gradients_begin_dev returns their number)");
    if (k != bw.taken) return fail(n, -6, "gradients_bucket: buckets are taken in order, 0 .. n - 1, after rcn_hipx_gradients_begin_dev");
    const int lo = k + 1 == (int)bw.lo.size() ? 0 : bw.lo[k];            // (the last bucket also walks whatever lies below its first layer with parameters: nothing)
    const int hi = bw.next;                                               // (hi < lo: the fused head has already handled this bucket's only layer)
    (void)dry_note(n, " bucket %d: layers %d .. %d", k, hi, lo);
    if (hi >= lo) RTRY(backward_layers(n, bw.x, bw.B, 0.f, bw.grad, false, hi, lo, false));
    RTRY(run_reduce_jobs(n, 0.f, false));
    bw.next = hi < lo - 1 ? hi : lo - 1;
    bw.taken = k + 1;
    if (off) *off = bw.off[k];
    if (len) *len = bw.len[k];
    (void)dry_note(n, " bucket %d done: grad[%lld, +%lld) = %.2f MB is final -- its all-reduce may start while the layers below run", k, bw.off[k], bw.len[k], bw.len[k] * 4.0 / 1e6);
    return 0;
}

}  // namespace

namespace {
// the layer table of a net (shapes, padded sizes, parameter offsets) from its description; no device involved
int describe_layers(rcn_hipx_net* n, int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers) {
    int H = in_h, W = in_w, C = in_c;
    bool flat = false;
    for (int i = 0; i < n_layers; ++i) {
        Layer l{};
        l.kind = layers[i].kind; l.H = H; l.W = W; l.Cin = C;
        if (l.kind == RCN_HIPX_CONV3X3_RELU) {
            if (flat) return fail(n, -2, "convolution after a dense layer");
            if (layers[i].out < 32 || layers[i].out % 32) return fail(n, -3, "conv output channels must be a positive multiple of 32");
            if (!(9 * C <= 32) && C % 32) return fail(n, -3, "conv input channels must be a multiple of 32 (or 9*Cin <= 32 for the first layer)");
            l.Cout = l.CoutP = layers[i].out; l.oH = H; l.oW = W; l.K = 9 * C;
            C = l.Cout;
        } else if (l.kind == RCN_HIPX_MAXPOOL2) {
            if (flat || i == 0 || n->L.back().kind != RCN_HIPX_CONV3X3_RELU) return fail(n, -2, "max-pool must follow a convolution");
            if (H % 2 || W % 2) return fail(n, -3, "max-pool needs even height and width");
            l.Cout = l.CoutP = C; l.oH = H / 2; l.oW = W / 2; l.K = 0;
            n->L.back().pool_follows = true;
            H = l.oH; W = l.oW;
        } else if (l.kind == RCN_HIPX_DENSE_RELU || l.kind == RCN_HIPX_DENSE) {
            const long long feat = flat ? C : (long long)H * W * C;
            if (feat % 32 || feat > 0x7fffffff) return fail(n, -3, "dense input features must be a multiple of 32");
            if (layers[i].out < 1) return fail(n, -1, "dense units must be positive");
            if (l.kind == RCN_HIPX_DENSE_RELU && layers[i].out % 32) return fail(n, -3, "hidden dense units must be a multiple of 32");
            if (l.kind == RCN_HIPX_DENSE && i != n_layers - 1) return fail(n, -2, "the logits layer must be last");
            l.K = (int)feat; l.H = 1; l.W = 1; l.Cin = (int)feat; l.Cout = layers[i].out; l.CoutP = (l.Cout + 31) / 32 * 32; l.oH = l.oW = 1;
            C = l.Cout; flat = true; H = W = 1;
        } else return fail(n, -1, "unknown layer kind");
        if (l.kind != RCN_HIPX_MAXPOOL2) {
            l.w_off = n->n_pad; n->n_pad += (long long)l.K * l.CoutP; l.b_off = n->n_pad; n->n_pad += l.CoutP;
            l.lw_off = n->n_log; n->n_log += (long long)l.K * l.Cout; l.lb_off = n->n_log; n->n_log += l.Cout;
        }
        n->L.push_back(l);
    }
    if (n->L.back().kind != RCN_HIPX_DENSE) return fail(n, -2, "the last layer must be RCN_HIPX_DENSE (logits)");
    n->classes = n->L.back().Cout;
    return 0;
}

// the layer descriptions of `from` without their buffers (host-only dry-run nets)
void copy_layer_table(rcn_hipx_net& to, const rcn_hipx_net& from) {
    for (const Layer& l : from.L) {
        Layer c;
        c.kind = l.kind; c.H = l.H; c.W = l.W; c.Cin = l.Cin; c.oH = l.oH; c.oW = l.oW; c.Cout = l.Cout; c.CoutP = l.CoutP; c.K = l.K;
        c.w_off = l.w_off; c.b_off = l.b_off; c.lw_off = l.lw_off; c.lb_off = l.lb_off; c.pool_follows = l.pool_follows;
        to.L.push_back(c);
    }
    to.n_pad = from.n_pad; to.n_log = from.n_log;
}

const char* precision_name(int precision, bool store16) { return precision != RCN_HIPX_BF16 ? "fp32 operands" : store16 ? "bf16 operands, the convolutional stage's tensors stored as bf16" : "bf16 operands"; }

}  // namespace

extern "C" {

int rcn_hipx_create(int device, int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int max_batch, void* stream, rcn_hipx_net** out) {
    if (!out || !layers || n_layers < 1 || in_h < 1 || in_w < 1 || in_c < 1 || max_batch < 1) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -5;
    rcn_hipx_net* n = new (std::nothrow) rcn_hipx_net();
    if (!n) return -7;
    *out = n;
    n->device = device; n->in_h = in_h; n->in_w = in_w; n->in_c = in_c; n->max_batch = max_batch;
    n->tiling = halo_f32_default();
    seed_options(n->opt);                             // the environment seeds the defaults, once, here
    RTRY(describe_layers(n, in_h, in_w, in_c, layers, n_layers));
    Dev g(device);
    if (stream) { n->stream = (hipStream_t)stream; } else { XTRY(n, hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking)); n->own_stream = true; }
    { const char* e = std::getenv("RCN_HIPX_OVERLAP"); n->overlap = e ? std::atoi(e) : 0; }
    XTRY(n, hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking));
    XTRY(n, n->params.ensure((size_t)n->n_pad * sizeof(float)));
    XTRY(n, hipMemsetAsync(n->params.p, 0, (size_t)n->n_pad * sizeof(float), n->stream));
    XTRY(n, n->wt.ensure((size_t)n->n_pad * sizeof(float)));
    XTRY(n, hipMemsetAsync(n->wt.p, 0, (size_t)n->n_pad * sizeof(float), n->stream));
    for (Layer& l : n->L) {
        const size_t elems = (size_t)max_batch * l.oH * l.oW * l.CoutP;
        XTRY(n, l.out.ensure(elems * sizeof(float)));
        XTRY(n, l.dout.ensure(elems * sizeof(float)));
        if (l.kind == RCN_HIPX_MAXPOOL2) XTRY(n, l.idx.ensure(elems));
    }
    XTRY(n, hipStreamSynchronize(n->stream));
    return 0;
}

void rcn_hipx_destroy(rcn_hipx_net* n) {
    if (!n) return;
    {
        Dev g(n->device);
        if (n->stream) (void)hipStreamSynchronize(n->stream);
        drop_graphs(n);
        for (Layer& l : n->L) { l.out.release(); l.idx.release(); l.dout.release(); l.slab.release(); }
        for (Buf* b : {&n->params, &n->wt, &n->wb16, &n->dz, &n->loss_part, &n->grad_tmp, &n->dlogits, &n->skbuf, &n->wb}) b->release();
        if (n->side) { (void)hipStreamSynchronize(n->side); (void)hipStreamDestroy(n->side); }
        for (hipEvent_t e : n->events) (void)hipEventDestroy(e);
        if (n->own_stream && n->stream) (void)hipStreamDestroy(n->stream);
    }
    delete n;
}

const char* rcn_hipx_last_error(const rcn_hipx_net* n) { return n ? n->err.c_str() : "null net"; }
int rcn_hipx_synchronize(rcn_hipx_net* n) { if (!n) return -1; Dev g(n->device); XTRY(n, hipStreamSynchronize(n->stream)); return 0; }
int rcn_hipx_param_count(const rcn_hipx_net* n, int64_t* logical, int64_t* padded) { if (!n) return -1; if (logical) *logical = n->n_log; if (padded) *padded = n->n_pad; return 0; }
int rcn_hipx_classes(const rcn_hipx_net* n) { return n ? n->classes : -1; }

int rcn_hipx_set_precision(rcn_hipx_net* n, int mode) {
    if (!n) return -1;
    if (mode != RCN_HIPX_FP32 && mode != RCN_HIPX_BF16 && mode != RCN_HIPX_BF16_STORED)
        return fail(n, -1, "set_precision: mode must be RCN_HIPX_FP32, RCN_HIPX_BF16 or RCN_HIPX_BF16_STORED");
    const int prec = mode == RCN_HIPX_FP32 ? RCN_HIPX_FP32 : RCN_HIPX_BF16;
    const bool st16 = mode == RCN_HIPX_BF16_STORED;
    if (st16) {
        // does every layer of this net run on kernels that take bf16 tensors?  Asked of the plan (the same walk as the step), before anything changes.
        rcn_hipx_net probe;
        probe.in_h = n->in_h; probe.in_w = n->in_w; probe.in_c = n->in_c; probe.max_batch = n->max_batch; probe.classes = n->classes;
        probe.precision = prec; probe.store16 = true; probe.tiling = n->tiling; probe.overlap = 0; probe.dry = true; probe.opt = n->opt;
        copy_layer_table(probe, *n);
        const int ps = step_core(&probe, nullptr, nullptr, n->max_batch, 0.f, nullptr, true, nullptr);
        if (ps != 0) return fail(n, ps, probe.err);
    }
    Dev g(n->device);
    if (prec != n->precision || st16 != n->store16) { XTRY(n, hipStreamSynchronize(n->stream)); drop_graphs(n); }
    n->precision = prec;
    n->store16 = st16;
    return 0;
}

int rcn_hipx_set_tiling(rcn_hipx_net* n, int mode) {
    if (!n) return -1;
    if (mode != RCN_HIPX_TILING_GEMM && mode != RCN_HIPX_TILING_AUTO && mode != RCN_HIPX_TILING_LDS) return fail(n, -1, "set_tiling: unknown mode");
    Dev g(n->device);
    if (mode != n->tiling) { XTRY(n, hipStreamSynchronize(n->stream)); drop_graphs(n); }
    n->tiling = mode;
    return 0;
}

int rcn_hipx_set_overlap(rcn_hipx_net* n, int on) {
    if (!n) return -1;
    Dev g(n->device);
    if (on != n->overlap) { XTRY(n, hipStreamSynchronize(n->stream)); drop_graphs(n); }
    n->overlap = on;
    return 0;
}

int rcn_hipx_set_option(rcn_hipx_net* n, const char* name, int value) {
    if (!n || !name) return -1;
    for (const XOptDesc& d : kXOptTable)
        if (std::strcmp(d.name, name) == 0) {
            if (value < d.lo || value > d.hi) return fail(n, -1, std::string("set_option: ") + name + " must be in " + std::to_string(d.lo) + ".." + std::to_string(d.hi));
            if (n->opt.*(d.field) == value) return 0;
            Dev g(n->device);
            XTRY(n, hipStreamSynchronize(n->stream));
            drop_graphs(n);                             // captured graphs bake in the kernels chosen
            n->opt.*(d.field) = value;
            return 0;
        }
    return fail(n, -1, std::string("set_option: unknown option '") + name + "'");
}

int rcn_hipx_get_option(const rcn_hipx_net* n, const char* name, int* value) {
    if (!n || !name || !value) return -1;
    for (const XOptDesc& d : kXOptTable)
        if (std::strcmp(d.name, name) == 0) { *value = n->opt.*(d.field); return 0; }
    return -1;
}

int rcn_hipx_set_params(rcn_hipx_net* n, const float* flat) {
    if (!n || !flat) return -1;
    Dev g(n->device);
    std::vector<float> pad((size_t)n->n_pad, 0.f);
    for (const Layer& l : n->L) {
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        for (int k = 0; k < l.K; ++k) std::memcpy(&pad[l.w_off + (long long)k * l.CoutP], &flat[l.lw_off + (long long)k * l.Cout], sizeof(float) * l.Cout);
        std::memcpy(&pad[l.b_off], &flat[l.lb_off], sizeof(float) * l.Cout);
    }
    XTRY(n, hipMemcpyAsync(n->params.p, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice, n->stream));
    RTRY(refresh_flipped(n));
    XTRY(n, hipStreamSynchronize(n->stream));
    return 0;
}

static int unpad(rcn_hipx_net* n, const float* dev, float* flat) {
    std::vector<float> pad((size_t)n->n_pad);
    XTRY(n, hipMemcpyAsync(pad.data(), dev, pad.size() * sizeof(float), hipMemcpyDeviceToHost, n->stream));
    XTRY(n, hipStreamSynchronize(n->stream));
    for (const Layer& l : n->L) {
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        for (int k = 0; k < l.K; ++k) std::memcpy(&flat[l.lw_off + (long long)k * l.Cout], &pad[l.w_off + (long long)k * l.CoutP], sizeof(float) * l.Cout);
        std::memcpy(&flat[l.lb_off], &pad[l.b_off], sizeof(float) * l.Cout);
    }
    return 0;
}

int rcn_hipx_get_params(rcn_hipx_net* n, float* flat) { if (!n || !flat) return -1; Dev g(n->device); return unpad(n, (const float*)n->params.p, flat); }
int rcn_hipx_unpad_host(rcn_hipx_net* n, const float* padded_dev, float* logical_host) { if (!n || !padded_dev || !logical_host) return -1; Dev g(n->device); return unpad(n, padded_dev, logical_host); }

int rcn_hipx_init_params(rcn_hipx_net* n, uint64_t seed) {
    if (!n) return -1;
    std::mt19937_64 gen(seed ? seed : std::random_device{}());
    std::vector<float> flat((size_t)n->n_log, 0.f);
    for (const Layer& l : n->L) {
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        std::normal_distribution<float> nd(0.f, std::sqrt(2.0f / (float)l.K));
        for (long long i = 0; i < (long long)l.K * l.Cout; ++i) flat[l.lw_off + i] = nd(gen);
    }
    return rcn_hipx_set_params(n, flat.data());
}

int rcn_hipx_forward_dev(rcn_hipx_net* n, const float* x, int B, float* logits) {
    if (!n || !x || !logits) return -1;
    RTRY(ensure_batch(n, B));
    Dev g(n->device);
    RTRY(prep_bf16_weights(n));
    RTRY(forward(n, x, B));
    const Layer& l = n->L.back();
    XTRY(n, hipMemcpy2DAsync(logits, (size_t)n->classes * sizeof(float), l.out.p, (size_t)l.CoutP * sizeof(float), (size_t)n->classes * sizeof(float), (size_t)B,
                             hipMemcpyDeviceToDevice, n->stream));
    return 0;
}

int rcn_hipx_train_step_dev(rcn_hipx_net* n, const float* x, const int32_t* labels, int B, float lr, float* loss_dev) {
    if (!n || !x || !labels) return -1;
    RTRY(ensure_batch(n, B));
    Dev g(n->device);
    const Key key{x, labels, B, lr, loss_dev};
    auto it = n->graphs.find(key);
    if (it == n->graphs.end()) {
        // one eager step first: sizes every scratch buffer outside capture (hipMalloc is illegal while capturing)
        RTRY(step_core(n, x, labels, B, lr, nullptr, true, loss_dev));
        hipGraph_t graph = nullptr;
        XTRY(n, hipStreamBeginCapture(n->stream, hipStreamCaptureModeThreadLocal));
        const int st = step_core(n, x, labels, B, lr, nullptr, true, loss_dev);
        hipError_t e = hipStreamEndCapture(n->stream, &graph);
        if (st != 0) { if (graph) (void)hipGraphDestroy(graph); return st; }
        XTRY(n, e);
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        XTRY(n, e);
        if (n->graphs.size() >= 8) drop_graphs(n);
        n->graphs.emplace(key, exec);
        return 0;                                       // the eager step above WAS this call's step
    }
    XTRY(n, hipGraphLaunch(it->second, n->stream));
    return 0;
}

int rcn_hipx_gradients_dev(rcn_hipx_net* n, const float* x, const int32_t* labels, int B, float* grad, float* loss_dev) {
    if (!n || !x || !labels || !grad) return -1;
    RTRY(ensure_batch(n, B));
    Dev g(n->device);
    XTRY(n, hipMemsetAsync(grad, 0, (size_t)n->n_pad * sizeof(float), n->stream));
    return step_core(n, x, labels, B, 0.f, grad, false, loss_dev);
}

int rcn_hipx_gradients_begin_dev(rcn_hipx_net* n, const float* x, const int32_t* labels, int B, float* grad, float* loss_dev, int64_t min_bucket_bytes, int* n_buckets) {
    if (!n || !x || !labels || !grad || !n_buckets || min_bucket_bytes < 0) return -1;
    RTRY(ensure_batch(n, B));
    Dev g(n->device);
    XTRY(n, hipMemsetAsync(grad, 0, (size_t)n->n_pad * sizeof(float), n->stream));
    const int nb = grad_begin(n, x, labels, B, grad, loss_dev, (long long)min_bucket_bytes);
    if (nb < 0) return nb;
    *n_buckets = nb;
    return 0;
}

int rcn_hipx_gradients_bucket_dev(rcn_hipx_net* n, int k, int64_t* off, int64_t* len) {
    if (!n) return -1;
    Dev g(n->device);
    long long o = 0, l = 0;
    RTRY(grad_bucket(n, k, &o, &l));
    if (off) *off = o;
    if (len) *len = l;
    return 0;
}

__global__ void k_axpy(float* __restrict__ p, const float* __restrict__ g, float scale, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = p[i] - scale * g[i];
}

int rcn_hipx_apply_dev(rcn_hipx_net* n, const float* grad, float scale) {
    if (!n || !grad) return -1;
    Dev g(n->device);
    hipLaunchKernelGGL(k_axpy, dim3(grid1d(n->n_pad, 256)), dim3(256), 0, n->stream, (float*)n->params.p, grad, scale, n->n_pad);
    XTRY(n, hipGetLastError());
    return refresh_flipped(n);
}

int rcn_hipx_step_flops(const rcn_hipx_net* n, int B, double* flops) {
    if (!n || !flops) return -1;
    double f = 0;
    for (size_t i = 0; i < n->L.size(); ++i) {
        const Layer& l = n->L[i];
        if (l.kind == RCN_HIPX_MAXPOOL2) continue;
        const double macs = (double)B * l.oH * l.oW * (double)l.K * l.Cout * (l.kind == RCN_HIPX_CONV3X3_RELU ? 1.0 : 1.0);
        f += 2.0 * macs * (i == 0 ? 2.0 : 3.0);         // forward + wgrad (+ dgrad except for the first layer)
    }
    *flops = f;
    return 0;
}

int rcn_hipx_plan(int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int batch, int precision, int tiling, char* out, int cap) {
    if (!layers || n_layers < 1 || in_h < 1 || in_w < 1 || in_c < 1 || batch < 1 || !out || cap < 1) return -1;
    if ((precision != RCN_HIPX_FP32 && precision != RCN_HIPX_BF16 && precision != RCN_HIPX_BF16_STORED) || tiling < RCN_HIPX_TILING_GEMM || tiling > RCN_HIPX_TILING_LDS) return -1;
    rcn_hipx_net net;                                   // host-only: no device, no stream, no buffers
    net.in_h = in_h; net.in_w = in_w; net.in_c = in_c; net.max_batch = batch;
    net.store16 = precision == RCN_HIPX_BF16_STORED;
    if (net.store16) precision = RCN_HIPX_BF16;
    net.precision = precision; net.tiling = tiling; net.overlap = 0; net.dry = true;
    seed_options(net.opt);                              // as a net created now would be (rcn_hipx_plan_net: an existing net's own options)
    int st = describe_layers(&net, in_h, in_w, in_c, layers, n_layers);
    if (st == 0) {
        net.plan = "forward + loss + backward of one batch of " + std::to_string(batch) + " (" + precision_name(precision, net.store16) + "), launch by launch:\n";
        st = step_core(&net, nullptr, nullptr, batch, 0.f, nullptr, true, nullptr);
    }
    const std::string& text = st == 0 ? net.plan : net.err;
    std::snprintf(out, (size_t)cap, "%s", text.c_str());
    return st;
}

// The bucketed gradient step of a data-parallel rank, launch by launch and bucket by bucket (no GPU needed): forward + loss, then for every
// bucket the backward pass of its layers, the ONE reduction launch of their slabs, and the slice of the flat gradient that is final there.
int rcn_hipx_plan_buckets(int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int batch, int precision, int tiling, int64_t min_bucket_bytes,
                          char* out, int cap) {
    if (!layers || n_layers < 1 || in_h < 1 || in_w < 1 || in_c < 1 || batch < 1 || !out || cap < 1 || min_bucket_bytes < 0) return -1;
    if ((precision != RCN_HIPX_FP32 && precision != RCN_HIPX_BF16 && precision != RCN_HIPX_BF16_STORED) || tiling < RCN_HIPX_TILING_GEMM || tiling > RCN_HIPX_TILING_LDS) return -1;
    rcn_hipx_net net;
    net.in_h = in_h; net.in_w = in_w; net.in_c = in_c; net.max_batch = batch;
    net.store16 = precision == RCN_HIPX_BF16_STORED;
    if (net.store16) precision = RCN_HIPX_BF16;
    net.precision = precision; net.tiling = tiling; net.overlap = 0; net.dry = true;
    seed_options(net.opt);
    int st = describe_layers(&net, in_h, in_w, in_c, layers, n_layers);
    if (st == 0) {
        net.plan = "gradients of one batch of " + std::to_string(batch) + " in buckets of at least " + std::to_string((long long)min_bucket_bytes) +
                   " bytes (data-parallel step: a bucket's all-reduce overlaps the backward pass below it):\n";
        const int nb = grad_begin(&net, nullptr, nullptr, batch, reinterpret_cast<float*>(sizeof(float)), nullptr, (long long)min_bucket_bytes);
        st = nb < 0 ? nb : 0;
        for (int k = 0; st == 0 && k < nb; ++k) st = grad_bucket(&net, k, nullptr, nullptr);
    }
    const std::string& text = st == 0 ? net.plan : net.err;
    std::snprintf(out, (size_t)cap, "%s", text.c_str());
    return st;
}

// the same walk for an EXISTING net, with that net's own precision, tiling and options: the plan and the step agree by construction
int rcn_hipx_plan_net(const rcn_hipx_net* n, int batch, char* out, int cap) {
    if (!n || batch < 1 || batch > n->max_batch || !out || cap < 1) return -1;
    rcn_hipx_net net;
    net.in_h = n->in_h; net.in_w = n->in_w; net.in_c = n->in_c; net.max_batch = batch; net.classes = n->classes;
    net.precision = n->precision; net.store16 = n->store16; net.tiling = n->tiling; net.overlap = 0; net.dry = true;
    net.opt = n->opt;
    copy_layer_table(net, *n);
    net.plan = "forward + loss + backward of one batch of " + std::to_string(batch) + " (" + precision_name(net.precision, net.store16) + "), launch by launch:\n";
    const int st = step_core(&net, nullptr, nullptr, batch, 0.f, nullptr, true, nullptr);
    const std::string& text = st == 0 ? net.plan : net.err;
    std::snprintf(out, (size_t)cap, "%s", text.c_str());
    return st;
}

}  // extern "C"
