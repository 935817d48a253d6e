// rcn_hip_api.hip -- the C ABI of include/rcn_hip.h on top of the gfx950 kernels.
// Host logic here mirrors the bookkeeping of RCN::{new, load_weights_and_bias, train_batch, classify}
// (rcn/src/rcn.rs); all arithmetic on sample data happens in the HIP kernels -- there is no CPU fallback.
#include "../../include/rcn_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "dense.hpp"
#include "dense_pipe.hpp"
#include "dense_wide.hpp"
#include "dense_p2.hpp"
#include "features.hpp"
#include "ops.hpp"
#include "dp_rccl.hpp"
#include "serve.hpp"
#include "dp_p2p.hpp"
#include "dense_p2_dp.hpp"
#include "dense_xcd.hpp"
// Parked experiments (a resident kernel per epoch segment, one launch per step, both step kernels as roles of one kernel object):
// correct, measured, slower than or equal to the default two-kernel pipeline.  They are compiled only into
// librcn_hip_exp.so (-DRCN_HIP_EXPERIMENTS; mercer_research_amd/build.py: build_experiments), which their tests and the stamp tools
// load; the shipping library does not carry them.
#ifdef RCN_HIP_EXPERIMENTS
#include "dense_p2_persist.hpp"
#include "dense_p2_step.hpp"
#endif
#include <atomic>
#include <chrono>

using namespace rcn;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct EpochKey {
    const void* X; const void* Y; const void* perm; size_t B; size_t nb; double eta; const void* loss; size_t j0 = 0;
    bool operator<(const EpochKey& o) const {
        return std::tie(X, Y, perm, B, nb, eta, loss, j0) < std::tie(o.X, o.Y, o.perm, o.B, o.nb, o.eta, o.loss, o.j0);
    }
};

}  // namespace

// Per-context options (rcn_hip_set_option / rcn_hip_get_option).  The environment variable of the same meaning only SEEDS the default
// when the context is created; two contexts of one process can differ, and nothing reads the environment afterwards.
struct CtxOptions {
    long long xcd = 1;                  // "xcd"               RCN_HIP_XCD            0: dense path 0 (auto) never selects the resident one-XCD kernel
    long long xcd_select = 0;           // "xcd_select"        RCN_HIP_XCD_SELECT     which blocks are the workers: blockIdx.x % 8 == this (0..7)
    long long xcd_gather = 0;           // "xcd_gather"        RCN_HIP_XCD_GATHER     1: rows fetched by the resident kernel itself (B = 256; measured slower)
    long long xcd_timeout_ticks = 20000000;   // "xcd_timeout_ticks"            bound of every wait inside the resident kernel, 100 MHz ticks (0.2 s)
    long long xcd_exact_lds = 0;        // "xcd_exact_lds"                            1: the resident kernel asks for exactly the LDS it uses (two workers may share a CU:
                                        //                                            what lets two contexts' kernels be resident on ONE device); 0: at least half a CU's
    long long xcd_fault_launch = 0;     // "xcd_fault_launch"                         test hook: the n-th resident launch of the context (1-based) loses a worker
    long long xcd_auto_fallback = 1;    // "xcd_auto_fallback"                        1: an expired wait of the single-GPU resident kernel re-runs the segment on the two-kernel pipeline
    long long dp_p2p = 1;               // "dp_p2p"            RCN_HIP_DP_P2P         0: no peer exchange (ncclAllReduce), 1: when world > 1, 2: also at world 1
    long long dp_fused = 1;             // "dp_fused"          RCN_HIP_DP_FUSED       0: the exchange never runs inside a step kernel
    long long dp_timeout_ticks = 100000000;   // "dp_timeout_ticks" RCN_HIP_DP_TIMEOUT_TICKS  bound of a peer wait, 100 MHz ticks (1 s)
    long long dp_cached_buf = 0;        // "dp_cached_buf"     RCN_HIP_DP_CACHED_BUF  1: exported buffers in ordinary (cached) device memory (A/B measurements)
    long long dp_graph = 1;             // "dp_graph"          RCN_HIP_DP_GRAPH       0: the three-kernel data-parallel step is enqueued eagerly
    long long feat_waves = 1;           // "feat_waves"        RCN_HIP_FEAT_WAVES     2: two waves per picture in k_features_cpcp (measured neutral)
    long long no_fragimg = 0;           // "no_fragimg"        RCN_HIP_NO_FRAGIMG     1: k_p2_b gathers its tail parameters itself
    long long exact_div_only = 0;       // "exact_div_only"    RCN_HIP_EXACT_DIV_ONLY 1: the f32 standardisation always divides
    long long pack_segment_bytes = (long long)64 << 20;   // "pack_segment_bytes" RCN_HIP_PACK_SEGMENT_BYTES  one half of the epoch image
};

namespace {
struct OptDesc { const char* name; const char* env; long long CtxOptions::*field; long long lo, hi; };
const OptDesc kOptTable[] = {
    {"xcd", "RCN_HIP_XCD", &CtxOptions::xcd, 0, 1},
    {"xcd_select", "RCN_HIP_XCD_SELECT", &CtxOptions::xcd_select, 0, 7},
    {"xcd_gather", "RCN_HIP_XCD_GATHER", &CtxOptions::xcd_gather, 0, 1},
    {"xcd_timeout_ticks", "RCN_HIP_XCD_TIMEOUT_TICKS", &CtxOptions::xcd_timeout_ticks, 1, 1LL << 40},
    {"xcd_exact_lds", "RCN_HIP_XCD_EXACT_LDS", &CtxOptions::xcd_exact_lds, 0, 1},
    {"xcd_auto_fallback", "RCN_HIP_XCD_AUTO_FALLBACK", &CtxOptions::xcd_auto_fallback, 0, 1},
    {"xcd_fault_launch", "RCN_HIP_XCD_FAULT_LAUNCH", &CtxOptions::xcd_fault_launch, 0, 0x7fffffff},
    {"dp_p2p", "RCN_HIP_DP_P2P", &CtxOptions::dp_p2p, 0, 2},
    {"dp_fused", "RCN_HIP_DP_FUSED", &CtxOptions::dp_fused, 0, 1},
    {"dp_timeout_ticks", "RCN_HIP_DP_TIMEOUT_TICKS", &CtxOptions::dp_timeout_ticks, 1, 1LL << 40},
    {"dp_cached_buf", "RCN_HIP_DP_CACHED_BUF", &CtxOptions::dp_cached_buf, 0, 1},
    {"dp_graph", "RCN_HIP_DP_GRAPH", &CtxOptions::dp_graph, 0, 1},
    {"feat_waves", "RCN_HIP_FEAT_WAVES", &CtxOptions::feat_waves, 1, 2},
    {"no_fragimg", "RCN_HIP_NO_FRAGIMG", &CtxOptions::no_fragimg, 0, 1},
    {"exact_div_only", "RCN_HIP_EXACT_DIV_ONLY", &CtxOptions::exact_div_only, 0, 1},
    {"pack_segment_bytes", "RCN_HIP_PACK_SEGMENT_BYTES", &CtxOptions::pack_segment_bytes, 1, 1LL << 40},
};
}  // namespace

struct rcn_hip_ctx {
    int device = 0;
    int dtype = RCN_HIP_F32;
    CtxOptions opt;
    std::string dp_fault;                   // RCN_HIP_DP_FAULT as it was when the context was created (fault injection for the admission tests)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    FeatDesc fd{};
    NetDesc nd{};
    int n_conv = 0;
    bool params_set = false;
    std::string dense_err;                  // non-empty: every dense call panics in the reference (see rcn_hip_create)
    double mean = 1.0, sd = 1.0;            // scale_set initial value (1,1): rcn.rs:71
    int feat_kernel = 0;                    // 0 auto, 1 always the generic k_features (tests compare the two)
    float fd_mean = 0.f, fd_sd = 0.f, fd_rcp = 0.f;   // last (mean, sd) checked by standardise_fast_is_exact; fd_rcp = 0: divide
    bool fd_checked = false;
    DevBuf pll;                             // persistent epoch kernel: tagged-word exchange buffers (dense_p2_persist.hpp)
    size_t pll_B = 0;
    unsigned ptag = 0;                      // last tag handed out; monotonic for the life of the context
    unsigned* perr_dev = nullptr;           // sticky timeout word of the persistent kernel, and its pinned mirror
    unsigned* perr_host = nullptr;
    int dense_path = 0;                     // 0 auto, 1 sample-tile kernels (dense.hpp), 2 feature-sliced pipeline (dense_pipe.hpp)
    DevBuf slab, xpack, ypack, p2buf;
    DevBuf fragimg;                         // tail parameters as k_p2_b's operand fragments (dense.hpp: p2_frag_scatter), f32 pipeline only
    bool frag_on = false;                   // set while enqueue_pipe_steps<float> runs: its k_p2_a / k_p2_b launches use the image
    DevBuf stepx;                           // one-launch step (dense_p2_step.hpp): flags of the sample groups, then the tag word
    size_t stepx_B = 0;
    size_t packed_B = 0, packed_nb = 0;     // what the epoch image currently holds (k_pack_epoch)
    size_t epoch_B = 0, epoch_nb = 0, epoch_seg = 0;   // rcn_hip_epoch_begin*_dev: the image holds batches 0..epoch_nb of a begun epoch
                                                       // (epoch_nb = 0: none; any other call that re-packs the image ends it)
    void* pin_host = nullptr;               // small pinned, device-mapped staging block for the serving path (classify)
    void* pin_dev = nullptr;
    DevBuf params, acts, deltas, loss_part, grad, xstage, ystage, ostage, scratch0, scratch1, scratch2, redpart, misc;
    std::map<EpochKey, hipGraphExec_t> graphs, dp_graphs, img_graphs, step_graphs;
    ncclComm_t comm = nullptr;              // data-parallel group (rcn_hip_dp_init); one rank per context
    int dp_rank = 0, dp_world = 1;
    struct P2P {                            // peer-read all-reduce over xGMI (dp_p2p.hpp)
        bool exported = false, attached = false, on = false;
        bool fused = false;                 // the exchange may run inside the gradient kernel (passed its own known-answer vote)
        bool push = false;                  // the pushed reduce-scatter + all-gather of the resident kernel passed its vote (dp_push.hpp)
        size_t push_off = 0;                // byte offset of its region in every rank's exported buffer
        void* local_buf = nullptr;          // [2][stride] values + [2][stride] tagged words + the pushed exchange's rows, uncached device memory
        size_t local_bytes = 0;
        bool local_uncached = false;
        unsigned* local_flags = nullptr;    // [kP2PMaxWorld], uncached device memory
        void* peer_buf[rcn::kP2PMaxWorld] = {};
        unsigned* peer_flags[rcn::kP2PMaxWorld] = {};
        size_t stride = 0;                  // values per slot (>= P+1, multiple of 4)
        unsigned seq = 0;                   // step sequence number, identical on every rank
        unsigned* err_dev = nullptr;        // sticky: 1 + rank that timed out; err_dev[16] = sequence base read by replayed graphs
        unsigned* err_host = nullptr;       // pinned copy, refreshed after every epoch call
        DevBuf raw, mism;
    } p2p;
    DevBuf xcdbuf;                          // one-XCD resident epoch kernel (dense_xcd.hpp): slab, deltas, fragment image, flags
    size_t xcd_B = 0;
    unsigned xcd_tag = 0;                   // last step tag handed out; monotonic for the life of the buffer
    unsigned* xerr_host = nullptr;          // its sticky error word: pinned host memory the kernel writes straight into (no copy-back in the stream)
    unsigned* xerr_dev = nullptr;
    struct { const float* X = nullptr; const float* Y = nullptr; const int32_t* perm = nullptr; size_t B = 0, nb = 0; } xg;   // the last call the resident
                                            // kernel ran in its gather form (rcn_hip_time_kernels_dev times that form over the same rows)
    bool xcd_dp_used = false;               // the resident kernel ran data-parallel steps since dp_init: dp_finalize clears its timeout word
    bool xcd_stepped_down = false;          // a wait of the resident kernel expired and the library stepped this context down to the two-kernel pipeline
    unsigned xcd_launch_id = 0;             // id of the newest resident launch enqueued (the kernel reports the newest COMPLETE one in xerr_host[1])
    unsigned long long xcd_launches = 0;    // resident launches of this context so far (option "xcd_fault_launch" counts them)
    int fallbacks_taken = 0;                // rcn_hip_fallbacks_taken
    bool replaying = false;                 // the redo log is being replayed: nothing is logged
    // Redo log of the training calls whose resident launches have not been seen complete yet (dense_xcd.hpp: parameters are written only
    // by a launch ALL of whose workers finished, and a launch that finds the error word set leaves at once -- so after a failure the
    // parameters are the state after launch xerr_host[1], and everything enqueued behind it can be re-run on the two-kernel pipeline).
    struct PermSource {                      // how a call's index rows came to be: a device shuffle (re-drawn from its seed) or a host upload (kept)
        int kind = 0;                        // 0 none (caller's buffer, taken as unchanged), 1 rcn_hip_shuffle_dev, 2 upload
        int32_t* buf = nullptr; size_t n = 0, passes = 0; uint64_t seed = 0;
        std::vector<int32_t> host;
    };
    struct XcdLaunchRec { unsigned id; size_t k0, n; };
    struct BeginRec {                        // the arguments of the newest rcn_hip_epoch_begin*_dev (the epoch image can be laid out again from them)
        const void* X = nullptr; const void* Y = nullptr; const int32_t* perm = nullptr; size_t B = 0, nb = 0; bool from_images = false; bool valid = false;
        PermSource src;
    };
    struct RedoRec {
        int kind = 0;                        // 0 train_epoch (X, Y, perm), 1 epoch_steps on the image of `begin`
        const void* X = nullptr; const void* Y = nullptr; const int32_t* perm = nullptr;
        size_t B = 0, nb = 0, j0 = 0; double eta = 0; void* loss_dev = nullptr; bool from_images = false;
        PermSource src;
        BeginRec begin;
        std::vector<XcdLaunchRec> launches;
    };
    std::vector<RedoRec> redo;
    std::vector<PermSource> perm_sources;    // newest source per index buffer
    BeginRec last_begin;
    unsigned* xerrd = nullptr;               // device copy of the resident kernel's sticky error word (outlives every workspace reset)
    int xcd_probe = 0;                      // 0 not probed, 1 the blocks with equal b % 8 share one XCD and the eight classes sit on eight XCDs, -1 not so
    struct ResidentSet {                     // rcn_hip_load_data: one of RCN::train's two data sets, kept in HBM (rcn.rs:134-137)
        DevBuf imgs, X, Y, perm, loss;
        size_t n = 0;
    } sets[2];
    std::map<const void*, size_t> lds_attr;   // kernels whose dynamic-LDS limit was already raised
    std::string err;
    size_t esz() const { return dtype == RCN_HIP_F64 ? 8 : 4; }
};

namespace {

int fail(rcn_hip_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RCN_HIP_ERR_OOM : RCN_HIP_ERR_HIP,                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                  \
    } while (0)

#define RCN_TRY(expr)                      \
    do {                                   \
        int s_ = (expr);                   \
        if (s_ != RCN_HIP_OK) return s_;   \
    } while (0)

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- shape logic shared by the operator API and the feature-stack validation --------------------------------
int conv_shape(int R, int C, int kr, int kc, int padding, int* oR, int* oC) {
    if (padding != RCN_HIP_PAD_NONE && padding != RCN_HIP_PAD_SAME) return RCN_HIP_ERR_INVALID_ARG;
    if (kr <= 0 || kc <= 0 || kr > R || kc > C) return RCN_HIP_ERR_SHAPE;          // kernel.rs:123-128
    if (padding == RCN_HIP_PAD_SAME) {
        if (kr % 2 == 0 || kc % 2 == 0) return RCN_HIP_ERR_SHAPE;                  // kernel.rs:131-135
        // the pad-copy loop (kernel.rs:154-158) reads self[(cy-1, cx-1)] up to cy = R+kr/2-1, cx = C+kc/2-1:
        // out of bounds (a panic) as soon as a half-width reaches 2
        const int cy_hi = R + kr / 2 - 1, cx_hi = C + kc / 2 - 1;
        if (cy_hi >= 1 && cx_hi >= 1 && (cy_hi - 1 >= R || cx_hi - 1 >= C)) return RCN_HIP_ERR_SHAPE;
        *oR = R; *oC = C;
    } else {
        *oR = R - kr + 1; *oC = C - kc + 1;
    }
    return RCN_HIP_OK;
}

int pool_shape(int R, int C, int padding, int* oR, int* oC) {
    if (padding != RCN_HIP_PAD_NONE && padding != RCN_HIP_PAD_SAME) return RCN_HIP_ERR_INVALID_ARG;
    if (R < 2 || C < 2) return RCN_HIP_ERR_SHAPE;                                  // kernel.rs:246-251
    if (padding == RCN_HIP_PAD_SAME) { *oR = (R + 1) / 2; *oC = (C + 1) / 2; }
    else { *oR = R / 2; *oC = C / 2; }
    return RCN_HIP_OK;
}

int build_feat_desc(rcn_hip_ctx* c, const rcn_hip_cfg* cfg) {
    FeatDesc& fd = c->fd;
    fd.H = cfg->in_h; fd.W = cfg->in_w; fd.n = cfg->n_convpool;
    long maps = 0;
    int R = fd.H, C = fd.W;
    long max_elems = (long)R * C;
    c->n_conv = 0;
    for (int i = 0; i < fd.n; ++i) {
        const int kind = cfg->convpool[i].kind, arg = cfg->convpool[i].arg;
        fd.kind[i] = kind; fd.arg[i] = arg;
        if (kind == RCN_HIP_LAYER_CONVOLVE2D) {
            if (arg != RCN_HIP_PAD_NONE && arg != RCN_HIP_PAD_SAME) return fail(c, RCN_HIP_ERR_INVALID_ARG, "Convolve2D: bad Padding");
            if (R < 3 || C < 3)                                                      // kernel.rs:199-201
                return fail(c, RCN_HIP_ERR_SHAPE, "convolve_2d_separated expects a matrix of at least 3x3");
            if (arg == RCN_HIP_PAD_NONE) { R -= 2; C -= 2; }
            maps = maps ? maps * 4 : 4;
            ++c->n_conv;
        } else if (kind == RCN_HIP_LAYER_POOL2D) {
            if (arg != RCN_HIP_POOL_AVERAGE && arg != RCN_HIP_POOL_MAX) return fail(c, RCN_HIP_ERR_INVALID_ARG, "Pool2D: bad Pooling");
            if (maps == 0) continue;                                                 // rcn.rs:343 on an empty feature_set
            if (R < 2 || C < 2) return fail(c, RCN_HIP_ERR_SHAPE, "pool_2d expects a matrix of at least 2x2");
            if (arg != RCN_HIP_POOL_MAX) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "Pooling::Average: Not implemented (kernel.rs:283)");
            R = (R + 1) / 2; C = (C + 1) / 2;
        } else {
            return fail(c, RCN_HIP_ERR_INVALID_ARG, "unknown RCNLayer kind");
        }
        if (maps * R * C > max_elems) max_elems = maps * R * C;
    }
    if (maps * (long)R * C > 0x7fffffffL) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "feature vector too long");
    fd.F = (int)(maps * R * C);
    fd.max_elems = (int)max_elems;
    return RCN_HIP_OK;
}

// load_weights_and_bias's fan-in: usize::pow(4,c) / usize::pow(2,p) * l with p += 2 per pool layer (rcn.rs:429-443)
long first_layer_fan_in(const rcn_hip_cfg* cfg, long l) {
    unsigned cc = 0, pp = 0;
    for (int i = 0; i < cfg->n_convpool; ++i) {
        if (cfg->convpool[i].kind == RCN_HIP_LAYER_CONVOLVE2D) cc += 1; else pp += 2;
    }
    unsigned long long num = 1, den = 1;
    for (unsigned i = 0; i < cc; ++i) num *= 4ULL;
    for (unsigned i = 0; i < pp; ++i) den *= 2ULL;
    return (long)(num / den * (unsigned long long)l);
}

int build_net_desc(rcn_hip_ctx* c, const rcn_hip_cfg* cfg) {
    NetDesc& nd = c->nd;
    nd.L = cfg->n_hidden + 1;                                                        // rcn.rs:426
    nd.dims[0] = c->fd.F;
    for (int i = 0; i < cfg->n_hidden; ++i) nd.dims[i + 1] = cfg->hidden[i];
    nd.dims[nd.L] = cfg->classes;
    long off = 0;
    nd.act_off[0] = 0; nd.act_off[1] = 0;
    nd.tile_start[0] = 0;
    for (int j = 0; j < nd.L; ++j) {
        if (nd.dims[j] <= 0 || nd.dims[j + 1] <= 0) return fail(c, RCN_HIP_ERR_SHAPE, "every dense layer needs at least one input and one output");
        nd.w_off[j] = (int)off;
        off += (long)nd.dims[j] * nd.dims[j + 1] + nd.dims[j + 1];
        if (off > 0x7fffffffL) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "more than 2^31 parameters");
        if (j + 1 <= nd.L && j + 2 <= kMaxLayers) nd.act_off[j + 2] = nd.act_off[j + 1] + nd.dims[j + 1];
        nd.tile_start[j + 1] = nd.tile_start[j] + (nd.dims[j] + 1 + 15) / 16;
    }
    nd.P = (int)off;
    return RCN_HIP_OK;
}

int sum_hidden_dims(const NetDesc& nd) { int s = 0; for (int j = 1; j <= nd.L; ++j) s += nd.dims[j]; return s; }

// ---- host <-> device dtype conversion ------------------------------------------------------------------------
int upload(rcn_hip_ctx* c, DevBuf& buf, const double* src, size_t count) {
    HIP_TRY(c, buf.ensure(count * c->esz()));
    if (count == 0) return RCN_HIP_OK;
    if (c->dtype == RCN_HIP_F64) {
        HIP_TRY(c, hipMemcpyAsync(buf.p, src, count * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        std::vector<float> tmp(count);
        for (size_t i = 0; i < count; ++i) tmp[i] = (float)src[i];
        HIP_TRY(c, hipMemcpyAsync(buf.p, tmp.data(), count * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return RCN_HIP_OK;
}

int download(rcn_hip_ctx* c, const void* dev, double* dst, size_t count) {
    if (count == 0) return RCN_HIP_OK;
    if (c->dtype == RCN_HIP_F64) {
        HIP_TRY(c, hipMemcpyAsync(dst, dev, count * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        std::vector<float> tmp(count);
        HIP_TRY(c, hipMemcpyAsync(tmp.data(), dev, count * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
    }
    return RCN_HIP_OK;
}

template <typename K>
int set_dyn_lds(rcn_hip_ctx* c, K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "layer sizes need more than 160 KiB of LDS per workgroup");
    const void* fn = reinterpret_cast<const void*>(kernel);
    if (bytes > 64 * 1024 && c->lds_attr[fn] < bytes) {
        HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        c->lds_attr[fn] = bytes;
    }
    return RCN_HIP_OK;
}

// ---- dense launches ---------------------------------------------------------------------------------------------
void drop_graphs(rcn_hip_ctx* c);
void drop_img_graphs(rcn_hip_ctx* c);

// Workspaces that captured hipGraphs point into: growing one moves it (DevBuf::ensure frees and reallocates), so every cached
// graph -- whichever call shape it was captured for -- would replay on freed memory.  A moved workspace drops them all; they are
// re-captured on demand.  (Found by running the benchmark with a warm-up shorter than the timed run.)
hipError_t ws_ensure(rcn_hip_ctx* c, DevBuf& b, size_t bytes) {
    // DevBuf::ensure frees the old block BEFORE it allocates the new one: the cached graphs must go first (they may still be
    // in flight on the stream -- drop_graphs drains it), and they must go on the out-of-memory path too, where b.p ends up null
    if (b.p && bytes > b.cap) {
        drop_graphs(c);
        if (&b == &c->xpack || &b == &c->ypack) c->epoch_nb = 0;      // a begun epoch's image goes with its buffer
    }
    return b.ensure(bytes);
}

int ensure_dense_ws(rcn_hip_ctx* c, size_t B) {
    const size_t sd = (size_t)sum_hidden_dims(c->nd);
    HIP_TRY(c, ws_ensure(c, c->acts, B * sd * c->esz()));
    HIP_TRY(c, ws_ensure(c, c->deltas, B * sd * c->esz()));
    HIP_TRY(c, ws_ensure(c, c->loss_part, ((B + kTileS - 1) / kTileS) * c->esz()));
    return RCN_HIP_OK;
}

template <typename T, bool TRAIN, bool VECX, bool STAGED>
int launch_fwd_v(rcn_hip_ctx* c, const void* x, const void* y, const int32_t* idx, size_t B, void* out) {
    const NetDesc& nd = c->nd;
    const int tiles = (int)((B + kTileS - 1) / kTileS);
    const size_t lds = dense_fwd_lds_elems(nd) * sizeof(T);
    RCN_TRY(set_dyn_lds(c, k_dense_fwd<T, TRAIN, VECX, STAGED>, lds));
    hipLaunchKernelGGL((k_dense_fwd<T, TRAIN, VECX, STAGED>), dim3(tiles), dim3(kDenseThreads), lds, c->stream, nd, (const T*)c->params.p,
                       (const T*)x, (const T*)y, idx, (int)B, TRAIN ? (T*)c->acts.p : (T*)nullptr, TRAIN ? (T*)c->deltas.p : (T*)nullptr,
                       TRAIN ? (T*)c->loss_part.p : (T*)nullptr, (T*)out);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// layer stacks too wide for k_dense_fwd's LDS image: layer by layer on global activations (dense_wide.hpp)
template <typename T>
int launch_fwd_wide(rcn_hip_ctx* c, bool train, const void* x, const void* y, const int32_t* idx, size_t B, void* out) {
    const NetDesc& nd = c->nd;
    T* acts = (T*)c->acts.p;
    T* deltas = (T*)c->deltas.p;
    auto layer = [&](int j) { return acts + B * (size_t)nd.act_off[j]; };          // activations of layer j >= 1: [B][d_j]
    auto grid = [](long long total) { long long g = (total + 255) / 256; return (unsigned)(g < 1 ? 1 : g > 8192 ? 8192 : g); };
    for (int j = 0; j < nd.L; ++j) {
        const T* ain = j == 0 ? (const T*)x : layer(j);
        T* aout = (!train && j + 1 == nd.L && out) ? (T*)out : layer(j + 1);
        hipLaunchKernelGGL((k_wide_forward<T>), dim3(grid((long long)B * nd.dims[j + 1])), dim3(256), 0, c->stream, nd, (const T*)c->params.p, j, ain,
                           (long long)nd.dims[j], j == 0 ? idx : (const int32_t*)nullptr, (int)B, aout);
    }
    if (train) {
        const int tiles = (int)((B + kTileS - 1) / kTileS);
        hipLaunchKernelGGL((k_wide_output_delta<T>), dim3(tiles), dim3(64), 0, c->stream, nd, (const T*)layer(nd.L), (const T*)y, idx, (int)B,
                           deltas + B * (size_t)nd.act_off[nd.L], (T*)c->loss_part.p);
        for (int j = nd.L - 1; j >= 1; --j)
            hipLaunchKernelGGL((k_wide_delta<T>), dim3(grid((long long)B * nd.dims[j])), dim3(256), 0, c->stream, nd, (const T*)c->params.p, j,
                               (const T*)(deltas + B * (size_t)nd.act_off[j + 1]), (const T*)layer(j), (int)B, deltas + B * (size_t)nd.act_off[j]);
    }
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

template <typename T>
int launch_fwd(rcn_hip_ctx* c, bool train, const void* x, const void* y, const int32_t* idx, size_t B, void* out) {
    if (dense_is_wide(c->nd, sizeof(T))) return launch_fwd_wide<T>(c, train, x, y, idx, B, out);
    const bool vec = dense_vec_rows(c->nd, sizeof(T)) && ((uintptr_t)x % 16 == 0);
    const bool st = dense_tail_staged(c->nd);
#define RCN_FWD(TR, V, S) return launch_fwd_v<T, TR, V, S>(c, x, y, idx, B, out)
    if (train) { if (vec) { if (st) RCN_FWD(true, true, true); else RCN_FWD(true, true, false); } else { if (st) RCN_FWD(true, false, true); else RCN_FWD(true, false, false); } }
    else       { if (vec) { if (st) RCN_FWD(false, true, true); else RCN_FWD(false, true, false); } else { if (st) RCN_FWD(false, false, true); else RCN_FWD(false, false, false); } }
#undef RCN_FWD
}

template <typename T>
int launch_wgrad(rcn_hip_ctx* c, bool apply, const void* x, const int32_t* idx, size_t B, double scale, void* grad_out,
                 void* loss_out, double loss_scale) {
    const NetDesc& nd = c->nd;
    const int grid = nd.tile_start[nd.L];
    const size_t lds = dense_wgrad_lds_elems() * sizeof(T);
    const int tiles = (int)((B + kTileS - 1) / kTileS);
    if (apply) {
        hipLaunchKernelGGL((k_dense_wgrad<T, true>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (T*)c->params.p, (T*)nullptr,
                           (const T*)x, idx, (const T*)c->acts.p, (const T*)c->deltas.p, (int)B, (T)scale, (const T*)c->loss_part.p,
                           tiles, (T)loss_scale, (T*)loss_out);
    } else {
        hipLaunchKernelGGL((k_dense_wgrad<T, false>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (T*)c->params.p, (T*)grad_out,
                           (const T*)x, idx, (const T*)c->acts.p, (const T*)c->deltas.p, (int)B, (T)scale, (const T*)c->loss_part.p,
                           tiles, (T)loss_scale, (T*)loss_out);
    }
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

bool use_pipe(const rcn_hip_ctx* c, size_t B) {
    if (c->dense_path == 1) return false;
    if (!pipe_supported(c->nd)) return false;
    if (c->dense_path == 2) return true;
    return B <= 1024;                       // beyond that the slabs (G x B x d1) cost more HBM traffic than they save
}

int ensure_pipe_ws(rcn_hip_ctx* c, size_t B) {
    const size_t Bp = (B + 15) / 16 * 16;
    const size_t mp = p2_supported(c->nd, B) ? (size_t)kP2H : (size_t)pipe_mp(c->nd);     // slab row: the specialised kernels pad to 32
    HIP_TRY(c, ws_ensure(c, c->slab, (size_t)pipe_slices(c->nd) * Bp * mp * c->esz()));
    HIP_TRY(c, ws_ensure(c, c->loss_part, ((B + kPipeTs - 1) / kPipeTs) * c->esz()));
    if (p2_supported(c->nd, B)) HIP_TRY(c, ws_ensure(c, c->p2buf, B * (size_t)(2 * kP2H + kP2C) * c->esz()));
    if (p2_supported(c->nd, B)) HIP_TRY(c, ws_ensure(c, c->fragimg, (size_t)kP2BFrag * 64 * c->esz()));
    return RCN_HIP_OK;
}

// RCN_HIP_P2_ONE_OBJECT=1: both kernels of a pipelined step as roles of one kernel object (dense_p2.hpp: k_p2_ab).  Measured 2 % slower
// than two kernels (9.75 vs 9.55 us/step), so off by default: the cost of alternating is not the switch of kernel object.
#ifdef RCN_HIP_EXPERIMENTS
static bool p2_one_object() { static const int v = [] { const char* e = std::getenv("RCN_HIP_P2_ONE_OBJECT"); return e ? std::atoi(e) : 0; }(); return v != 0; }
#else
static constexpr bool p2_one_object() { return false; }
#endif

template <typename T>
int launch_pipe_a(rcn_hip_ctx* c, const void* xp, const void* xn, size_t B, double scale, void* loss_out, double loss_scale, bool do_update, bool do_fwd) {
    const NetDesc& nd = c->nd;
    const int G = pipe_slices(nd), grid = G + pipe_extra_wgs(nd);
    const size_t lds = pipe_a_lds_elems(nd) * sizeof(T);
    const int n_loss = (int)((B + kPipeTs - 1) / kPipeTs);
    if (p2_supported(nd, B)) {       // lean specialisation: one hidden layer <= 32, classes <= 16, B % 256 == 0
        T* a1 = (T*)c->p2buf.p; T* d1 = a1 + B * kP2H; T* d2 = d1 + B * kP2H;
#ifdef RCN_HIP_EXPERIMENTS
        if (p2_one_object())
            hipLaunchKernelGGL((k_p2_ab<T>), dim3(grid), dim3(kDenseThreads), p2_ab_lds_elems() * sizeof(T), c->stream, 1, nd, (T*)c->params.p, (const T*)xp,
                               (const T*)xn, (const T*)nullptr, (int)B, a1, d1, d2, (T)scale, (T*)c->slab.p, G, (T*)c->loss_part.p, n_loss, (T)loss_scale,
                               (T*)loss_out, do_update ? 1 : 0, do_fwd ? 1 : 0);
        else
#endif
            hipLaunchKernelGGL((k_p2_a<T>), dim3(grid), dim3(kDenseThreads), p2_a_lds_elems() * sizeof(T), c->stream, nd, (T*)c->params.p, (const T*)xp,
                               (const T*)xn, (int)B, (const T*)a1, (const T*)d1, (const T*)d2, (T)scale, (T*)c->slab.p, G, (const T*)c->loss_part.p, n_loss,
                               (T)loss_scale, (T*)loss_out, do_update ? 1 : 0, do_fwd ? 1 : 0, c->frag_on ? (T*)c->fragimg.p : (T*)nullptr);
        HIP_TRY(c, hipGetLastError());
        return RCN_HIP_OK;
    }
    RCN_TRY(set_dyn_lds(c, k_pipe_a<T>, lds));
    hipLaunchKernelGGL((k_pipe_a<T>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (T*)c->params.p, (const T*)xp, (const T*)xn, (int)B,
                       (const T*)c->acts.p, (const T*)c->deltas.p, (T)scale, (T*)c->slab.p, G, (const T*)c->loss_part.p, n_loss, (T)loss_scale,
                       (T*)loss_out, do_update ? 1 : 0, do_fwd ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

template <typename T>
int launch_pipe_b(rcn_hip_ctx* c, const void* ys, size_t B) {
    const NetDesc& nd = c->nd;
    if (p2_supported(nd, B)) {
        T* a1 = (T*)c->p2buf.p; T* d1 = a1 + B * kP2H; T* d2 = d1 + B * kP2H;
#ifdef RCN_HIP_EXPERIMENTS
        if (p2_one_object())
            hipLaunchKernelGGL((k_p2_ab<T>), dim3((unsigned)(B / kP2Ts)), dim3(kDenseThreads), p2_ab_lds_elems() * sizeof(T), c->stream, 0, nd, (T*)c->params.p,
                               (const T*)nullptr, (const T*)nullptr, (const T*)ys, (int)B, a1, d1, d2, (T)0, (T*)c->slab.p, pipe_slices(nd), (T*)c->loss_part.p, 0,
                               (T)0, (T*)nullptr, 0, 0);
        else
#endif
            hipLaunchKernelGGL((k_p2_b<T>), dim3((unsigned)(B / kP2Ts)), dim3(kP2BThreads), p2_b_lds_elems() * sizeof(T), c->stream, nd, (const T*)c->params.p,
                               (const T*)c->slab.p, pipe_slices(nd), (const T*)ys, (int)B, a1, d1, d2, (T*)c->loss_part.p,
                               c->frag_on ? (const T*)c->fragimg.p : (const T*)nullptr);
        HIP_TRY(c, hipGetLastError());
        return RCN_HIP_OK;
    }
    const size_t lds = pipe_b_lds_elems(nd) * sizeof(T);
    RCN_TRY(set_dyn_lds(c, k_pipe_b<T>, lds));
    hipLaunchKernelGGL((k_pipe_b<T>), dim3((unsigned)((B + kPipeTs - 1) / kPipeTs)), dim3(kPipeBThreads), lds, c->stream, nd, (const T*)c->params.p,
                       (const T*)c->slab.p, pipe_slices(nd), (const T*)ys, (int)B, (T*)c->acts.p, (T*)c->deltas.p, (T*)c->loss_part.p);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// The epoch image is kept to two segments of at most ~64 MB each so that it stays resident in the 256 MB Infinity Cache
// next to the source set however many batches one call covers (a 411 MB image made the per-step kernels ~15 % slower).
size_t pack_segment(const rcn_hip_ctx* c, size_t B) {
    const size_t per_batch = (size_t)pipe_slices(c->nd) * B * 16 * c->esz();
    const size_t seg = (size_t)c->opt.pack_segment_bytes / (per_batch ? per_batch : 1);
    return seg ? seg : 1;
}

// batches [j0, j0+n) of the call -> slice-major image (k_pack_epoch) in half `half` of the context-owned scratch
template <typename T>
int launch_pack(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t j0, size_t n, int half, size_t seg) {
    const NetDesc& nd = c->nd;
    const int G = pipe_slices(nd), F = nd.dims[0], Cc = nd.dims[nd.L];
    if (n > 65535) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "train_epoch: segment of more than 65535 batches");
    const bool vec = dense_vec_rows(nd, sizeof(T)) && ((uintptr_t)X % 16 == 0);
    // identity order: batch j is rows [jB, (j+1)B) -> shift the base pointers; shuffled: shift the index pointer
    const T* Xb = perm ? (const T*)X : (const T*)X + j0 * B * (size_t)F;
    const T* Yb = perm ? (const T*)Y : (const T*)Y + j0 * B * (size_t)Cc;
    const int32_t* pb = perm ? perm + j0 * B : nullptr;
    T* xs = (T*)c->xpack.p + (size_t)half * seg * G * B * 16;
    T* ys = (T*)c->ypack.p + (size_t)half * seg * B * Cc;
    if (vec) hipLaunchKernelGGL((k_pack_epoch<T, true>), dim3(pack_grid_x(G), (unsigned)n), dim3(256), 0, c->stream, Xb, Yb, pb, (int)B, F, Cc, G, xs, ys);
    else hipLaunchKernelGGL((k_pack_epoch<T, false>), dim3(pack_grid_x(G), (unsigned)n), dim3(256), 0, c->stream, Xb, Yb, pb, (int)B, F, Cc, G, xs, ys);
    HIP_TRY(c, hipGetLastError());
    if (half == 0) { c->packed_B = B; c->packed_nb = n; }
    c->epoch_nb = 0;                        // whatever epoch rcn_hip_epoch_begin_dev had laid out is overwritten
    return RCN_HIP_OK;
}

// Grid of a one-wave-per-workgroup kernel that loops over its work: exactly as many workgroups as the device holds at
// once (CUs x resident workgroups per CU for that kernel's LDS footprint), so every image loop runs in a single pass --
// a grid larger than that queues the excess behind the first pass and the tail runs on a part-empty chip.
template <typename Kern>
static int resident_grid(rcn_hip_ctx* c, Kern kern, size_t work, int block = 64) {
    static std::map<std::pair<int, const void*>, int> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(c->device, (const void*)kern);
    auto it = cache.find(key);
    if (it == cache.end()) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, block, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1) cus = 256;
        it = cache.emplace(key, per_cu * cus).first;
    }
    return (int)(work < (size_t)it->second ? work : (size_t)it->second);
}

// f32 standardisation in the specialised feature kernels: the reciprocal to use, or 0 when only a true division is
// bit-exact for the scale in force (features.hpp: standardise_fast_is_exact, checked once per (mean, sd))
static float fast_standardise_rcp(rcn_hip_ctx* c) {
    if (c->opt.exact_div_only) return 0.f;
    const float m = (float)c->mean, sd = (float)c->sd;
    if (!c->fd_checked || std::memcmp(&m, &c->fd_mean, 4) != 0 || std::memcmp(&sd, &c->fd_sd, 4) != 0) {
        c->fd_mean = m; c->fd_sd = sd; c->fd_rcp = 0.f;
        float y = 0.f;
        if (standardise_fast_is_exact(m, sd, Cpcp<28, 28>::VMAX, &y)) c->fd_rcp = y;
        c->fd_checked = true;
    }
    return c->fd_rcp;
}

// the same image straight from u8 pictures: features + standardise + slice-major packing in one kernel (features.hpp)
template <typename T>
int launch_feat_pack(rcn_hip_ctx* c, const uint8_t* imgs, const void* Y, const int32_t* perm, size_t B, size_t j0, size_t n, int half, size_t seg) {
    const NetDesc& nd = c->nd;
    const int G = pipe_slices(nd), Cc = nd.dims[nd.L], HW = c->fd.H * c->fd.W;
    const uint8_t* ib = perm ? imgs : imgs + j0 * B * (size_t)HW;
    const T* Yb = perm ? (const T*)Y : (const T*)Y + j0 * B * (size_t)Cc;
    const int32_t* pb = perm ? perm + j0 * B : nullptr;
    T* xs = (T*)c->xpack.p + (size_t)half * seg * G * B * 16;
    T* ys = (T*)c->ypack.p + (size_t)half * seg * B * Cc;
    const size_t total = n * B;
    float rcp = 0.f;
    if constexpr (std::is_same<T, float>::value) rcp = fast_standardise_rcp(c);
    const int grid = rcp != 0.f ? resident_grid(c, k_features_cpcp_packed<28, 28, T, true>, total) : resident_grid(c, k_features_cpcp_packed<28, 28, T, false>, total);
    if (rcp != 0.f)
        hipLaunchKernelGGL((k_features_cpcp_packed<28, 28, T, true>), dim3(grid), dim3(64), 0, c->stream, ib, Yb, pb, (int)B, (int)n, G, Cc, (T)c->mean,
                           (T)c->sd, (T)rcp, xs, ys);
    else
        hipLaunchKernelGGL((k_features_cpcp_packed<28, 28, T, false>), dim3(grid), dim3(64), 0, c->stream, ib, Yb, pb, (int)B, (int)n, G, Cc, (T)c->mean,
                           (T)c->sd, (T)0, xs, ys);
    HIP_TRY(c, hipGetLastError());
    if (half == 0) { c->packed_B = B; c->packed_nb = n; }
    c->epoch_nb = 0;
    return RCN_HIP_OK;
}

int ensure_pack_ws(rcn_hip_ctx* c, size_t B, size_t nb) {
    const size_t seg = pack_segment(c, B), cap = nb <= seg ? nb : 2 * seg;
    HIP_TRY(c, ws_ensure(c, c->xpack, cap * (size_t)pipe_slices(c->nd) * B * 16 * c->esz()));
    HIP_TRY(c, ws_ensure(c, c->ypack, cap * B * (size_t)c->nd.dims[c->nd.L] * c->esz()));
    return RCN_HIP_OK;
}

// nb consecutive train_batch steps through the feature-sliced pipeline: pack, A(F0) B0 A(U0,F1) B1 ... A(U_{nb-1}),
// re-packing the next segment (into the other half of the image) just before the step that first needs it.
// perm (nullable) holds nb*B sample indices; without it batch j is rows [jB, (j+1)B) of X / Y.
// prepacked (rcn_hip_epoch_steps_dev): the image already holds the begun epoch (laid out with segment length pre_seg); the call
// runs its batches j0 .. j0+nb and packs nothing.
template <typename T>
int enqueue_pipe_steps(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev,
                       bool from_images = false, bool prepacked = false, size_t j0 = 0, size_t pre_seg = 0) {
    const size_t G = pipe_slices(c->nd), Cc = c->nd.dims[c->nd.L], es = c->esz();
    const double scale = eta / (double)B, loss_scale = 1.0 / (2.0 * (double)B);
    const size_t seg = prepacked ? pre_seg : (nb <= pack_segment(c, B) ? nb : pack_segment(c, B));
    auto slot = [&](size_t j) { j += j0; return ((j / seg) % 2) * seg + j % seg; };
    auto xb = [&](size_t j) { return (const void*)((const char*)c->xpack.p + slot(j) * G * B * 16 * es); };
    auto yb = [&](size_t j) { return (const void*)((const char*)c->ypack.p + slot(j) * B * Cc * es); };
    // from_images: X is the resident u8 picture set; features, standardisation and packing are one kernel per segment
    auto pack = [&](size_t j0) {
        const size_t n = nb - j0 < seg ? nb - j0 : seg;
        return from_images ? launch_feat_pack<T>(c, (const uint8_t*)X, Y, perm, B, j0, n, (int)((j0 / seg) % 2), seg)
                           : launch_pack<T>(c, X, Y, perm, B, j0, n, (int)((j0 / seg) % 2), seg);
    };
    // f32, default shape class: the tail parameters also live as an image of k_p2_b's operand fragments, built here from the
    // parameter vector and kept current by k_p2_a's tail tiles for the rest of this call
    struct FragGuard { rcn_hip_ctx* c; ~FragGuard() { c->frag_on = false; } } frag_guard{c};
    if constexpr (std::is_same<T, float>::value) {
        if (p2_supported(c->nd, B) && !p2_one_object() && !c->opt.no_fragimg && c->fragimg.p) {
            hipLaunchKernelGGL(k_p2_fragimg, dim3(1), dim3(512), 0, c->stream, c->nd, (const float*)c->params.p, (float*)c->fragimg.p);
            HIP_TRY(c, hipGetLastError());
            c->frag_on = true;
        }
    }
    if (!prepacked) RCN_TRY(pack(0));
    RCN_TRY(launch_pipe_a<T>(c, xb(0), xb(0), B, scale, nullptr, loss_scale, false, true));
    for (size_t j = 0; j < nb; ++j) {
        RCN_TRY(launch_pipe_b<T>(c, yb(j), B));
        const bool more = j + 1 < nb;
        if (!prepacked && more && (j + 1) % seg == 0) RCN_TRY(pack(j + 1));
        void* lj = loss_dev ? (char*)loss_dev + j * es : nullptr;
        RCN_TRY(launch_pipe_a<T>(c, xb(j), more ? xb(j + 1) : xb(j), B, scale, lj, loss_scale, true, more));
    }
    return RCN_HIP_OK;
}

#ifdef RCN_HIP_EXPERIMENTS
constexpr long long kPersistTimeoutTicks = 5000000LL;        // 50 ms of the 100 MHz wall clock per wait

// ---- one launch per step (dense_p2_step.hpp): A(F0) S0 S1 ... S_{nb-1}, S_j = sample groups of batch j + feature slices
// (update from batch j, partials of batch j+1) + tail tiles in ONE kernel; the last node advances the tag word so the
// captured graph can be replayed.
bool use_step(const rcn_hip_ctx* c, size_t B) {
    if (c->dtype != RCN_HIP_F32 || !step_supported(c->nd, B)) return false;
    if (c->dense_path == 4) return true;
    if (c->dense_path != 0) return false;
    static const int auto_on = [] { const char* e = std::getenv("RCN_HIP_STEP_KERNEL"); return e ? std::atoi(e) : 0; }();
    return auto_on != 0;
}

int ensure_step_ws(rcn_hip_ctx* c, size_t B) {
    const size_t NS = B / kP2Ts, bytes = (NS * kStepFlagStride + 64) * sizeof(unsigned) + B * kP2H * sizeof(pw_t);
    if (!c->perr_dev) {
        HIP_TRY(c, hipMalloc((void**)&c->perr_dev, 256));
        HIP_TRY(c, hipHostMalloc((void**)&c->perr_host, 64, hipHostMallocDefault));
        *c->perr_host = 0;
        HIP_TRY(c, hipMemsetAsync(c->perr_dev, 0, 256, c->stream));
    }
    if (*c->perr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "train_epoch: a wait inside the one-launch step timed out in an earlier call; the parameters are no longer "
                                        "consistent.  rcn_hip_set_dense_path(ctx, 2) avoids this kernel");
    if (c->stepx_B != B || c->stepx.cap < bytes) {
        HIP_TRY(c, ws_ensure(c, c->stepx, bytes));
        HIP_TRY(c, hipMemsetAsync(c->stepx.p, 0, c->stepx.cap, c->stream));       // flags 0, tag word 0: the first tag is 1
        drop_graphs(c);                                                            // cached graphs count on the tag word's history
        c->stepx_B = B;
    }
    return RCN_HIP_OK;
}

int enqueue_step_epoch(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool from_images) {
    using T = float;
    const NetDesc& nd = c->nd;
    const size_t G = pipe_slices(nd), Cc = nd.dims[nd.L], es = sizeof(T), NS = B / kP2Ts;
    const double scale = eta / (double)B, loss_scale = 1.0 / (2.0 * (double)B);
    const size_t seg = nb <= pack_segment(c, B) ? nb : pack_segment(c, B);
    auto slot = [&](size_t j) { return ((j / seg) % 2) * seg + j % seg; };
    auto xb = [&](size_t j) { return (const T*)((const char*)c->xpack.p + slot(j) * G * B * 16 * es); };
    auto yb = [&](size_t j) { return (const T*)((const char*)c->ypack.p + slot(j) * B * Cc * es); };
    auto pack = [&](size_t j0) {
        const size_t n = nb - j0 < seg ? nb - j0 : seg;
        return from_images ? launch_feat_pack<T>(c, (const uint8_t*)X, Y, perm, B, j0, n, (int)((j0 / seg) % 2), seg)
                           : launch_pack<T>(c, X, Y, perm, B, j0, n, (int)((j0 / seg) % 2), seg);
    };
    StepBufs sb;
    sb.slab = (T*)c->slab.p;
    sb.a1 = (T*)c->p2buf.p; sb.d1 = sb.a1 + B * kP2H; sb.d2 = sb.d1 + B * kP2H;
    sb.loss = (T*)c->loss_part.p;
    sb.oflag = (unsigned*)c->stepx.p;
    unsigned* tagw = sb.oflag + NS * kStepFlagStride + 32;
    sb.tag = tagw;
    sb.d1w = (pw_t*)(sb.oflag + NS * kStepFlagStride + 64);
    static const int first_look = [] { const char* e = std::getenv("RCN_HIP_STEP_FIRST_LOOK"); return e ? std::atoi(e) : 320; }();   // 100 MHz ticks
    const int grid = step_grid(nd, B);
    RCN_TRY(pack(0));
    RCN_TRY(launch_pipe_a<T>(c, xb(0), xb(0), B, scale, nullptr, loss_scale, false, true));
    for (size_t j = 0; j < nb; ++j) {
        const bool more = j + 1 < nb;
        if (more && (j + 1) % seg == 0) RCN_TRY(pack(j + 1));
        T* lj = loss_dev ? (T*)loss_dev + j : nullptr;
        hipLaunchKernelGGL(k_p2_step, dim3(grid), dim3(kPersistThreads), 0, c->stream, nd, (T*)c->params.p, xb(j), more ? xb(j + 1) : xb(j), yb(j), (int)B,
                           (int)G, (T)scale, (T)loss_scale, lj, sb, (unsigned)(j + 1), more ? 1 : 0, c->perr_dev, kPersistTimeoutTicks, first_look);
        HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_add_u32, dim3(1), dim3(1), 0, c->stream, tagw, (unsigned)(nb + 1));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->perr_host, c->perr_dev, 4, hipMemcpyDeviceToHost, c->stream));     // looked at by the next call
    return RCN_HIP_OK;
}

// ---- one resident kernel per epoch segment (dense_p2_persist.hpp) ------------------------------------------------
bool use_persist(const rcn_hip_ctx* c, size_t B) {
    if (c->dtype != RCN_HIP_F32 || !persist_supported(c->nd, B)) return false;
    if (c->dense_path == 3) return true;
    if (c->dense_path != 0) return false;
    static const int auto_on = [] { const char* e = std::getenv("RCN_HIP_PERSIST"); return e ? std::atoi(e) : 0; }();
    return auto_on != 0;
}


int enqueue_persist_epoch(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev) {
    const NetDesc& nd = c->nd;
    const size_t G = pipe_slices(nd), Cc = nd.dims[nd.L];
    if (!c->perr_dev) {
        HIP_TRY(c, hipMalloc((void**)&c->perr_dev, 256));
        HIP_TRY(c, hipHostMalloc((void**)&c->perr_host, 64, hipHostMallocDefault));
        *c->perr_host = 0;
        HIP_TRY(c, hipMemsetAsync(c->perr_dev, 0, 256, c->stream));
    }
    if (*c->perr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "train_epoch: a resident epoch kernel timed out in an earlier call (its workgroups were not all on the GPU at "
                                        "once -- is the device shared?); the parameters are no longer consistent.  rcn_hip_set_dense_path(ctx, 2) avoids this kernel");
    const size_t bytes = persist_bytes(B, G);
    if (c->pll_B != B || c->pll.cap < bytes) {
        HIP_TRY(c, c->pll.ensure(bytes));
        HIP_TRY(c, hipMemsetAsync(c->pll.p, 0, c->pll.cap, c->stream));      // flag / tag 0 never matches (tags start at 1)
        c->pll_B = B;
    }
    const size_t NS = B / kP2Ts;
    PersistBufs pb;
    float* f = (float*)c->pll.p;
    pb.slab = f; f += 2 * NS * G * kP2Ts * kP2H;
    pb.d1 = f;   f += 2 * B * kP2H;
    pb.a1 = f;   f += 2 * B * kP2H;
    pb.d2 = f;   f += 2 * B * kP2C;
    pb.loss = f; f += 2 * NS;
    unsigned* u = (unsigned*)f;
    pb.sflag = u; u += 64;
    pb.oflag = u; u += NS;
    pb.tail = (pw_t*)(((uintptr_t)u + 63) & ~(uintptr_t)63);
    const size_t seg = nb <= pack_segment(c, B) ? nb : pack_segment(c, B);
    const float scale = (float)(eta / (double)B), loss_scale = (float)(1.0 / (2.0 * (double)B));
    const int grid = persist_grid(nd, B);
    for (size_t j0 = 0; j0 < nb; j0 += seg) {
        const size_t n = nb - j0 < seg ? nb - j0 : seg;
        const int half = (int)((j0 / seg) % 2);
        RCN_TRY(launch_pack<float>(c, X, Y, perm, B, j0, n, half, seg));
        const float* xs = (const float*)c->xpack.p + (size_t)half * seg * G * B * 16;
        const float* ys = (const float*)c->ypack.p + (size_t)half * seg * B * Cc;
        hipLaunchKernelGGL(k_p2_epoch, dim3(grid), dim3(kPersistThreads), 0, c->stream, nd, (float*)c->params.p, xs, ys, (int)B, (int)n, (int)G, scale,
                           loss_scale, loss_dev ? (float*)loss_dev + j0 : (float*)nullptr, pb, c->ptag, c->perr_dev, kPersistTimeoutTicks);
        HIP_TRY(c, hipGetLastError());
        c->ptag += (unsigned)n + 2;
    }
    HIP_TRY(c, hipMemcpyAsync(c->perr_host, c->perr_dev, 4, hipMemcpyDeviceToHost, c->stream));     // looked at by the next call
    return RCN_HIP_OK;
}

#else
static bool use_step(const rcn_hip_ctx*, size_t) { return false; }
static bool use_persist(const rcn_hip_ctx*, size_t) { return false; }
#endif

// ---- the resident one-XCD epoch kernel (dense_xcd.hpp) ------------------------------------------------------------------
constexpr size_t kXcdProbeLds = xcd_lds_floats(256) * sizeof(float);

// Are the 32 blocks with blockIdx.x % 8 == 0 of a 256-block launch with this LDS footprint on ONE XCD, and every other block
// elsewhere?  Asked once per context, synchronously, before the resident kernel is ever selected (the kernel checks again itself).
int xcd_probe(rcn_hip_ctx* c) {
    if (c->xcd_probe != 0) return RCN_HIP_OK;
    c->xcd_probe = -1;
    const size_t lds = kXcdProbeLds;
    RCN_TRY(set_dyn_lds(c, k_xcd_probe, lds));
    DevBuf out;
    HIP_TRY(c, out.ensure(8 * kXcdWorkers * sizeof(unsigned)));
    std::vector<unsigned> host(8 * kXcdWorkers);
    int good = 0;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_TRY(c, hipMemsetAsync(out.p, 0, host.size() * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(k_xcd_probe, dim3(8 * kXcdWorkers), dim3(kXcdThreads), lds, c->stream, (unsigned*)out.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(host.data(), out.p, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bool ok = true;
        for (size_t b = 0; b < host.size(); ++b) {                       // block b sits on the XCD of block b % 8, and those eight differ
            if (!(host[b] & 0x100u)) ok = false;
            for (size_t k = 0; k < 8; ++k)
                if ((b % 8 == k) != ((host[b] & 0xfu) == (host[k] & 0xfu))) ok = false;
        }
        good += ok ? 1 : 0;
    }
    out.release();
    if (good == 3) c->xcd_probe = 1;
    return RCN_HIP_OK;
}

bool use_xcd(rcn_hip_ctx* c, size_t B) {
    if (c->dtype != RCN_HIP_F32 || !xcd_supported(c->nd, B)) return false;
    if (c->dense_path != 0 && c->dense_path != 5) return false;
    if (c->xcd_stepped_down) return false;                              // (rcn_hip_set_dense_path(ctx, 5) arms it again)
    if (c->dense_path == 0 && c->opt.xcd == 0) return false;
    if (c->xcd_probe == 0 && xcd_probe(c) != RCN_HIP_OK) return false;
    return c->xcd_probe == 1;
}

int xcd_heal(rcn_hip_ctx* c);

// the data-parallel step runs on the resident kernel: the in-kernel exchange was admitted, one hidden layer, a shard of 32 / 64 / 128 / 256
bool dp_on_xcd(rcn_hip_ctx* c, size_t B) {
    return c->p2p.on && c->p2p.push && c->dtype == RCN_HIP_F32 && c->nd.L == 2 && (size_t)xcd_bt(B) == B && use_xcd(c, B);
}

int ensure_xcd_ws(rcn_hip_ctx* c, size_t B) {
    if (!c->xerr_host) {
        HIP_TRY(c, hipHostMalloc((void**)&c->xerr_host, 64, hipHostMallocMapped));
        c->xerr_host[0] = 0;                    // [0] the sticky error word, [1] id of the newest launch all of whose workers finished
        c->xerr_host[1] = 0;
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->xerr_dev, c->xerr_host, 0));
        HIP_TRY(c, hipMalloc((void**)&c->xerrd, 256));
        HIP_TRY(c, hipMemsetAsync(c->xerrd, 0, 256, c->stream));
    }
    if (*c->xerr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, *c->xerr_host == 2 ? "train_epoch: the resident kernel's workgroups did not share one XCD in an earlier call; nothing was "
                                                             "updated by it.  rcn_hip_set_dense_path(ctx, 2) selects the two-kernel pipeline"
                                                           : "train_epoch: a bounded wait inside the resident kernel expired in an earlier call (is the device shared?); "
                                                             "that call's segment was not applied.  rcn_hip_set_dense_path(ctx, 2) selects the two-kernel pipeline");
    const size_t BT = (size_t)xcd_bt(B);
    const size_t bytes = xcd_buf_bytes(c->nd, BT);
    if (c->xcd_B != BT || c->xcdbuf.cap < bytes) {
        HIP_TRY(c, c->xcdbuf.ensure(bytes));
        HIP_TRY(c, hipMemsetAsync(c->xcdbuf.p, 0, c->xcdbuf.cap, c->stream));      // flags 0: tags start at 1; error word 0
        c->xcd_B = BT;
        c->xcd_tag = 0;
    }
    return RCN_HIP_OK;
}

P2PDesc p2p_desc(const rcn_hip_ctx* c);
static long long p2p_timeout_ticks(const rcn_hip_ctx* c);

// the workspace of one batch instantiation, carved out of c->xcdbuf
static XcdBufs xcd_bufs(rcn_hip_ctx* c, size_t BT) {
    const size_t NS = BT / kP2Ts, NA = (size_t)xcd_na(c->nd);
    XcdBufs xb;
    float* f = (float*)c->xcdbuf.p;
    xb.slab = f; f += NS * NA * kP2Ts * kP2H;
    xb.d1 = f;   f += BT * kP2H;
    xb.a1 = f;   f += BT * kP2H;
    xb.d2 = f;   f += BT * kP2C;
    xb.a2 = f;   f += BT * kP2C;
    xb.d3 = f;   f += BT * kP2C;
    xb.loss = f; f += NS;
    xb.fragimg = f; f += (size_t)kP3BFrag * 64;
    unsigned* u = (unsigned*)(((uintptr_t)f + 127) & ~(uintptr_t)127);
    xb.flagA = u; u += kXcdWorkers * kXcdFlagStride;
    xb.flagB = u; u += kXcdWorkers * kXcdFlagStride;
    xb.xcc = u;   u += kXcdWorkers * kXcdFlagStride;
    xb.flagD = u; u += kXcdWorkers * kXcdFlagStride;
    xb.flagT = u;
    xb.errd = c->xerrd;
    xb.done = c->xerr_dev + 1;
    return xb;
}

// One launch of the instantiation for batch BT.  The kernel asks for at least half a CU's LDS plus one byte so that no two of its
// workers share a CU (option "xcd_exact_lds" = 1: exactly what it uses -- two contexts' resident kernels can then be on one device).
template <int BT, bool FULL>
int xcd_launch_bt(rcn_hip_ctx* c, const float* xs, const float* ys, size_t B, size_t nb, float scale, float loss_scale, float* loss_dev, bool dp,
                  const int32_t* gperm, bool gather, const XcdBufs& xb, unsigned tag0, unsigned launch_id) {
    const NetDesc& nd = c->nd;
    size_t lds = xcd_lds_floats(BT) * sizeof(float);
    if (!c->opt.xcd_exact_lds && lds < 81 * 1024) lds = 81 * 1024;
    const long long to = c->opt.xcd_timeout_ticks;
    const int xsel = (int)c->opt.xcd_select;
#define RCN_XCD_LAUNCH(KERN, TO, DPARG)                                                                                                                   \
    do {                                                                                                                                                  \
        RCN_TRY(set_dyn_lds(c, KERN, lds));                                                                                                               \
        hipLaunchKernelGGL(KERN, dim3(8 * kXcdWorkers), dim3(kXcdThreads), lds, c->stream, nd, (float*)c->params.p, xs, ys, (int)B, (int)nb,              \
                           pipe_slices(nd), scale, loss_scale, loss_dev, xb, tag0, c->xerr_dev, TO, DPARG, xsel, (const int*)gperm, launch_id);           \
    } while (0)
    if (dp) {
        // (the data-parallel form exists for whole instantiation sizes: a shard of 32 / 64 / 128 / 256 samples per rank)
        if constexpr (FULL) {
            RCN_XCD_LAUNCH((k_xcd_epoch<BT, true, true>), to + 2 * p2p_timeout_ticks(c),
                           (XcdDpOn{PushDesc{p2p_desc(c), c->p2p.stride, c->p2p.push_off}, c->p2p.seq + 1, p2p_timeout_ticks(c)}));
            c->p2p.seq += (unsigned)nb;
            c->xcd_dp_used = true;
        } else return fail(c, RCN_HIP_ERR_UNSUPPORTED, "the resident kernel's data-parallel form needs a shard of 32, 64, 128 or 256 samples");
    } else if (nd.L == 3) {
        RCN_XCD_LAUNCH((k_xcd_epoch<BT, FULL, false, true>), to, XcdDpOff{});
    } else if (gather) {
        if constexpr (BT == 256 && FULL) RCN_XCD_LAUNCH((k_xcd_epoch<256, true, false, false, true>), to, XcdDpOff{});
        else return fail(c, RCN_HIP_ERR_UNSUPPORTED, "the gather form of the resident kernel exists for batch 256 only");
    } else {
        RCN_XCD_LAUNCH((k_xcd_epoch<BT, FULL, false>), to, XcdDpOff{});
    }
#undef RCN_XCD_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// nb consecutive steps over batches whose packed images are contiguous from xs / ys (one segment of the epoch image).
// dp: the data-parallel step -- B is this rank's shard, the update uses the global batch length, gradients meet inside the kernel.
// gather: xs / ys are the caller's X[rows][F] / Y[rows][C] as stored and gperm the order of their rows (NULL: stored order) -- the kernel
// fetches every batch's rows itself, a step ahead; else they are the packed epoch image (k_pack_epoch) and gperm is unused.
// *id_out (nullable): the launch's id, which the kernel reports in `done` once all of its workers have finished.
int enqueue_xcd_steps(rcn_hip_ctx* c, const float* xs, const float* ys, size_t B, size_t nb, double eta, float* loss_dev, bool dp = false,
                      const int32_t* gperm = nullptr, bool gather = false, unsigned* id_out = nullptr) {
    const int BT = xcd_bt(B);
    const XcdBufs xb = xcd_bufs(c, (size_t)BT);
    // (the tail parameters as the sample groups' operand fragments -- xb.fragimg -- are written by the kernel's own tail tiles: at its
    // start from the parameter vector, then after every update; its pads are the zeros the buffer was created with)
    const unsigned tag0 = c->xcd_tag + 1;
    // launch ids are unique in the process (the placement vote of a launch accepts only answers that carry its id: dense_xcd.hpp) and
    // increase along a context's stream (the redo journal compares them with the id the kernel reports complete)
    static std::atomic<unsigned> g_launch{0};
    unsigned id = (g_launch.fetch_add(1) + 1u) & 0x7fffffffu;
    if (id == 0) id = (g_launch.fetch_add(1) + 1u) & 0x7fffffffu;
    c->xcd_launch_id = id;
    c->xcd_launches += 1;
    const bool faulty = c->opt.xcd_fault_launch != 0 && (long long)c->xcd_launches == c->opt.xcd_fault_launch;
    const unsigned id_arg = id | (faulty ? 0x80000000u : 0u);
    const double Bg = (double)B * (dp ? (double)c->dp_world : 1.0);          // the global batch.len() of rcn.rs:214
    const float scale = (float)(eta / Bg), loss_scale = (float)(1.0 / (2.0 * Bg));
    int st;
    const bool full = (size_t)BT == B;
#define RCN_XCD_BT(N)                                                                                                                        \
    st = full ? xcd_launch_bt<N, true>(c, xs, ys, B, nb, scale, loss_scale, loss_dev, dp, gperm, gather, xb, tag0, id_arg)                  \
              : xcd_launch_bt<N, false>(c, xs, ys, B, nb, scale, loss_scale, loss_dev, dp, gperm, gather, xb, tag0, id_arg)
    switch (BT) {
    case 32:  RCN_XCD_BT(32); break;
    case 64:  RCN_XCD_BT(64); break;
    case 128: RCN_XCD_BT(128); break;
    default:  RCN_XCD_BT(256); break;
    }
#undef RCN_XCD_BT
    RCN_TRY(st);
    c->xcd_tag += (unsigned)nb;
    if (id_out) *id_out = id;
    return RCN_HIP_OK;
}

// The gather form of the resident kernel (rows fetched by the workers themselves; f32 feature vectors whose rows are whole 16-byte
// chunks) is OFF unless RCN_HIP_XCD_GATHER=1.  Measured on MI355X (bench workload): the kernel's step takes 7.4 us this way against
// 6.45 us on the packed image -- a wave's loads retire in order, so every wait for a slab or flag word that follows the prefetch also
// waits for 256 scattered 128-byte reads, where the packed image gives it one 32 KB run -- and k_pack_epoch's gather costs only
// 0.39 us per step amortised: 7.39 vs 6.80 us per step in the bench's steady state.
constexpr size_t kXcdMaxStepsPerLaunch = 1u << 20;
static bool xcd_gather(const rcn_hip_ctx* c) {
    return c->opt.xcd_gather != 0 && c->nd.dims[0] % 4 == 0 && c->nd.L == 2;
}

// the newest source of the index rows `perm` points into (a shuffle or an upload the library performed), or none
static rcn_hip_ctx::PermSource perm_source_of(const rcn_hip_ctx* c, const int32_t* perm) {
    if (perm)
        for (auto it = c->perm_sources.rbegin(); it != c->perm_sources.rend(); ++it)
            if (perm >= it->buf && perm < it->buf + it->n * it->passes) return *it;
    return rcn_hip_ctx::PermSource{};
}
static void note_perm_source(rcn_hip_ctx* c, rcn_hip_ctx::PermSource&& src) {
    if (c->replaying) return;
    for (auto& e : c->perm_sources)
        if (e.buf == src.buf) { e = std::move(src); return; }
    if (c->perm_sources.size() >= 8) c->perm_sources.erase(c->perm_sources.begin());
    c->perm_sources.push_back(std::move(src));
}

// a whole call on the resident kernel: batches [j0, j0 + nb) of the call, packed segment by segment (or already packed)
int enqueue_xcd_epoch(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool from_images,
                      bool prepacked, size_t j0, size_t pre_seg, bool dp = false) {
    const size_t G = pipe_slices(c->nd), Cc = c->nd.dims[c->nd.L];
    // single-GPU calls are journalled until their launches have been seen complete (xcd_verify): what a failed launch did not apply is
    // re-run from here on the two-kernel pipeline
    rcn_hip_ctx::RedoRec* rec = nullptr;
    if (!dp && !c->replaying && c->opt.xcd_auto_fallback) {
        // (launches the kernel has already reported complete need no record any more: xerr_host[1] is pinned memory, read for free)
        if (c->xerr_host && c->xerr_host[0] == 0) {
            const unsigned done = c->xerr_host[1];
            size_t keep = 0;
            while (keep < c->redo.size() && (c->redo[keep].launches.empty() || (int)(c->redo[keep].launches.back().id - done) <= 0)) ++keep;
            if (keep) c->redo.erase(c->redo.begin(), c->redo.begin() + keep);
        }
        c->redo.emplace_back();
        rec = &c->redo.back();
        rec->kind = prepacked ? 1 : 0;
        rec->X = X; rec->Y = Y; rec->perm = perm; rec->B = B; rec->nb = nb; rec->j0 = j0; rec->eta = eta; rec->loss_dev = loss_dev; rec->from_images = from_images;
        if (!prepacked) rec->src = perm_source_of(c, perm);
        else rec->begin = c->last_begin;
    }
    auto note = [&](unsigned id, size_t k0, size_t n) { if (rec) rec->launches.push_back({id, k0, n}); };
    if (!prepacked && !from_images && !dp && xcd_gather(c)) {
        c->xg.X = (const float*)X; c->xg.Y = (const float*)Y; c->xg.perm = perm; c->xg.B = B; c->xg.nb = nb;
        // feature vectors as stored: no packed image at all -- ONE launch walks the whole call, every worker gathering its 128 bytes of
        // each row of the batch after next while it works on the current one (the bytes k_pack_epoch would read, write and hand back)
        for (size_t k = 0; k < nb;) {
            const size_t n = nb - k < kXcdMaxStepsPerLaunch ? nb - k : kXcdMaxStepsPerLaunch;
            unsigned id = 0;
            RCN_TRY(enqueue_xcd_steps(c, (const float*)X, (const float*)Y, B, n, eta, loss_dev ? (float*)loss_dev + k : nullptr, dp,
                                      perm ? perm + k * B : nullptr, true, &id));
            note(id, k, n);
            if (!perm) { X = (const float*)X + n * B * c->nd.dims[0]; Y = (const float*)Y + n * B * Cc; }
            k += n;
        }
        return RCN_HIP_OK;
    }
    const size_t seg = prepacked ? pre_seg : (nb <= pack_segment(c, B) ? nb : pack_segment(c, B));
    auto slot = [&](size_t j) { return ((j / seg) % 2) * seg + j % seg; };
    for (size_t j = prepacked ? j0 : 0, end = j + nb, k = 0; j < end;) {
        const size_t in_seg = seg - j % seg, n = end - j < in_seg ? end - j : in_seg;       // up to the end of this segment of the image
        if (!prepacked) {
            const int half = (int)((j / seg) % 2);
            RCN_TRY(from_images ? launch_feat_pack<float>(c, (const uint8_t*)X, Y, perm, B, j, n, half, seg) : launch_pack<float>(c, X, Y, perm, B, j, n, half, seg));
        }
        const float* xs = (const float*)c->xpack.p + slot(j) * G * B * 16;
        const float* ys = (const float*)c->ypack.p + slot(j) * B * Cc;
        unsigned id = 0;
        RCN_TRY(enqueue_xcd_steps(c, xs, ys, B, n, eta, loss_dev ? (float*)loss_dev + k : nullptr, dp, nullptr, false, &id));
        note(id, k, n);
        j += n; k += n;
    }
    return RCN_HIP_OK;                      // (the sticky error word lives in pinned host memory: current once the stream has drained)
}

// one train_batch (rcn.rs:176-223) on device-resident data; idx selects the batch's rows (or NULL)
int enqueue_train_step(rcn_hip_ctx* c, const void* x, const void* y, const int32_t* idx, size_t B, double eta, void* loss_dev) {
    const double scale = eta / (double)B;                       // rcn.rs:214: eta / batch.len() as f64
    const double loss_scale = 1.0 / (2.0 * (double)B);
    if (use_pipe(c, B)) {
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, 1));
        return c->dtype == RCN_HIP_F64 ? enqueue_pipe_steps<double>(c, x, y, idx, B, 1, eta, loss_dev)
                                       : enqueue_pipe_steps<float>(c, x, y, idx, B, 1, eta, loss_dev);
    }
    if (c->dtype == RCN_HIP_F64) {
        RCN_TRY(launch_fwd<double>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<double>(c, true, x, idx, B, scale, nullptr, loss_dev, loss_scale));
    } else {
        RCN_TRY(launch_fwd<float>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<float>(c, true, x, idx, B, scale, nullptr, loss_dev, loss_scale));
    }
    return RCN_HIP_OK;
}

int check_ctx(const rcn_hip_ctx* c) { return c ? RCN_HIP_OK : RCN_HIP_ERR_INVALID_ARG; }

// the resident kernel's sticky error word, both copies (a recovery action of the caller, the heal below, the end of a data-parallel group)
int xcd_clear_error(rcn_hip_ctx* c) {
    if (c->xerr_host) c->xerr_host[0] = 0;
    if (c->xerrd) HIP_TRY(c, hipMemsetAsync(c->xerrd, 0, 4, c->stream));
    return RCN_HIP_OK;
}

// Self-healing step-down of the single-GPU resident kernel.  Precondition: the stream is drained and the sticky word is set (a bounded
// wait expired -- typically a co-tenant holds CUs of the XCD, so the 32 workers were never resident together -- or the workers were
// not on one XCD).  Nothing a failed launch computed reached memory and every launch enqueued behind it left at once, so the
// parameter vector is the state after launch xerr_host[1]: the context steps down to the two-kernel pipeline for good, and every
// step the journal holds beyond that launch is re-run there, from the arguments its call was given (index rows the library itself
// shuffled or uploaded are re-created first; anything else the calls read is taken to be unchanged -- the contract of an
// asynchronous call whose inputs must stay untouched until a synchronise).  Reported through rcn_hip_fallbacks_taken, not as an error.
static int redo_perm(rcn_hip_ctx* c, const rcn_hip_ctx::PermSource& ps) {
    if (ps.kind == 1) return rcn_hip_shuffle_dev(c, ps.buf, ps.n, ps.passes, ps.seed);
    if (ps.kind == 2) {
        HIP_TRY(c, hipMemcpyAsync(ps.buf, ps.host.data(), ps.host.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return RCN_HIP_OK;
}
static int redo_begin(rcn_hip_ctx* c, const rcn_hip_ctx::BeginRec& b) {
    RCN_TRY(redo_perm(c, b.src));
    return b.from_images ? rcn_hip_epoch_begin_images_dev(c, (const uint8_t*)b.X, b.Y, b.perm, b.B, b.nb) : rcn_hip_epoch_begin_dev(c, b.X, b.Y, b.perm, b.B, b.nb);
}
int xcd_heal(rcn_hip_ctx* c) {
    const unsigned code = c->xerr_host[0], done = c->xerr_host[1];
    RCN_TRY(xcd_clear_error(c));
    c->xcd_stepped_down = true;
    c->fallbacks_taken += 1;
    std::vector<rcn_hip_ctx::RedoRec> redo;
    redo.swap(c->redo);
    c->replaying = true;
    struct Guard { rcn_hip_ctx* c; ~Guard() { c->replaying = false; } } guard{c};
    const size_t F = (size_t)c->nd.dims[0], Cc = (size_t)c->nd.dims[c->nd.L], es = c->esz(), HW = (size_t)c->fd.H * c->fd.W;
    const bool image_was_live = c->epoch_nb != 0;
    bool image_touched = false;
    for (const auto& r : redo) {
        size_t k0 = r.nb;                        // the first step of this call no complete launch covered
        for (const auto& l : r.launches)
            if ((int)(l.id - done) > 0) { k0 = l.k0; break; }
        if (k0 >= r.nb) continue;
        void* loss = r.loss_dev ? (char*)r.loss_dev + k0 * es : nullptr;
        if (r.kind == 0) {
            RCN_TRY(redo_perm(c, r.src));
            const int32_t* pm = r.perm ? r.perm + k0 * r.B : nullptr;
            const void* Y = r.perm ? r.Y : (const void*)((const char*)r.Y + k0 * r.B * Cc * es);
            if (r.from_images) {
                const uint8_t* X = r.perm ? (const uint8_t*)r.X : (const uint8_t*)r.X + k0 * r.B * HW;
                RCN_TRY(rcn_hip_train_epoch_images_dev(c, X, Y, pm, r.B, r.nb - k0, r.eta, loss));
            } else {
                const void* X = r.perm ? r.X : (const void*)((const char*)r.X + k0 * r.B * F * es);
                RCN_TRY(rcn_hip_train_epoch_dev(c, X, Y, pm, r.B, r.nb - k0, r.eta, loss));
            }
            image_touched = true;
        } else {
            if (!r.begin.valid) return fail(c, RCN_HIP_ERR_HIP, "the resident kernel failed and the epoch image its steps ran on cannot be laid out again; what it had not applied is lost");
            RCN_TRY(redo_begin(c, r.begin));
            RCN_TRY(rcn_hip_epoch_steps_dev(c, r.j0 + k0, r.nb - k0, r.eta, loss));
            image_touched = true;
        }
    }
    // the index buffers and the epoch image end as the caller's newest calls left them
    for (const auto& ps : c->perm_sources) RCN_TRY(redo_perm(c, ps));
    if (image_touched && image_was_live && c->last_begin.valid) RCN_TRY(redo_begin(c, c->last_begin));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)code;
    return RCN_HIP_OK;
}

// The in-kernel waits (peer exchange, resident / one-launch step kernels) are bounded: a wait that expires sets a sticky
// device word, every later kernel of that family drains, and the updates of the call are only partly applied.  The word is
// copied back asynchronously at the end of each epoch call; after a stream synchronise it is current.  Every entry point
// that tells the caller "the work is complete / here are the parameters" calls this behind its synchronise.
int sticky_errors(rcn_hip_ctx* c) {
    if (c->p2p.err_host && c->p2p.err_dev) {
        if (*c->p2p.err_host != 0)
            return fail(c, RCN_HIP_ERR_HIP, "data-parallel exchange: rank " + std::to_string(c->dp_rank) + " timed out waiting for peer data (sticky word " +
                                                std::to_string(*c->p2p.err_host) + "); the last call's updates are incomplete and the replicas are no longer in step");
    }
    if (c->perr_host && *c->perr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "a bounded wait inside the resident / one-launch step kernel expired; the last call's updates are incomplete");
    if (c->xerr_host && *c->xerr_host != 0) {
        if (!c->xcd_dp_used && c->opt.xcd_auto_fallback && !c->replaying) return xcd_heal(c);
        return fail(c, RCN_HIP_ERR_HIP, *c->xerr_host == 2 ? "the resident one-XCD kernel found its workgroups on different XCDs; what it had not applied is lost"
                                                           : (c->xcd_dp_used ? "a bounded wait inside the resident one-XCD kernel expired in a data-parallel step; the replicas are no longer in step"
                                                                             : "a bounded wait inside the resident one-XCD kernel expired; what it had not applied is lost"));
    }
    if (c->xerr_host) c->redo.clear();          // the stream is drained and nothing failed: every journalled launch is complete
    return RCN_HIP_OK;
}

// a failure of the resident kernel the host can already see: healed (or reported) before anything else is enqueued behind it
int xcd_entry_check(rcn_hip_ctx* c) {
    if (c->xerr_host && c->xerr_host[0] != 0 && !c->replaying) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return sticky_errors(c);
    }
    return RCN_HIP_OK;
}

int need_dense(rcn_hip_ctx* c) {
    if (!c->dense_err.empty()) return fail(c, RCN_HIP_ERR_SHAPE, c->dense_err);
    return RCN_HIP_OK;
}

int need_params(rcn_hip_ctx* c) {
    RCN_TRY(need_dense(c));
    if (!c->params_set) return fail(c, RCN_HIP_ERR_STATE, "parameters not set: call rcn_hip_set_params for every layer or rcn_hip_init_params first");
    return RCN_HIP_OK;
}

// ---- peer-read all-reduce plumbing (dp_p2p.hpp) ------------------------------------------------------------------
// Uncached (fine-grained) device memory is never handed back to the runtime while the process lives: it is parked here and reused by the
// next data-parallel group.  Measured in round 3: after a 3.9 MB hipDeviceMallocUncached block had been hipFree'd, the next context's
// ORDINARY hipMalloc allocations came back on that memory still behaving uncached -- plain stores no longer stayed in the XCD's L2 and
// the resident kernel's hand-offs (payload, drain, flag: dense_xcd.hpp) were read stale: deterministic wrong costs in the first step
// of a context created right after a data-parallel one, gone with RCN_HIP_DP_CACHED_BUF=1 and gone with this cache.
struct UncachedCache {
    std::mutex mu;
    std::vector<std::tuple<int, size_t, void*>> free_list;        // (device, bytes, pointer)
    hipError_t alloc(int device, size_t bytes, void** out) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); ++i)
                if (std::get<0>(free_list[i]) == device && std::get<1>(free_list[i]) == bytes) {
                    *out = std::get<2>(free_list[i]);
                    free_list.erase(free_list.begin() + i);
                    return hipSuccess;
                }
        }
        return hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached);
    }
    void park(int device, size_t bytes, void* p) {
        std::lock_guard<std::mutex> lk(mu);
        free_list.emplace_back(device, bytes, p);
    }
};
UncachedCache& uncached_cache() { static UncachedCache* u = new UncachedCache(); return *u; }      // (never destroyed: no hipFree at exit)
constexpr size_t kP2PFlagBytes = 4096;

void p2p_release(rcn_hip_ctx* c) {
    auto& q = c->p2p;
    if (!c->dp_graphs.empty() && c->stream) (void)hipStreamSynchronize(c->stream);     // a replay may still be in flight
    for (auto& kv : c->dp_graphs) (void)hipGraphExecDestroy(kv.second);      // they hold pointers into the buffers freed below
    c->dp_graphs.clear();
    for (int r = 0; r < kP2PMaxWorld; ++r) {
        if (q.attached && r != c->dp_rank) {
            if (q.peer_buf[r]) (void)hipIpcCloseMemHandle(q.peer_buf[r]);
            if (q.peer_flags[r]) (void)hipIpcCloseMemHandle(q.peer_flags[r]);
        }
        q.peer_buf[r] = nullptr;
        q.peer_flags[r] = nullptr;
    }
    if (q.local_buf) { if (q.local_uncached) uncached_cache().park(c->device, q.local_bytes, q.local_buf); else (void)hipFree(q.local_buf); }
    if (q.local_flags) uncached_cache().park(c->device, kP2PFlagBytes, q.local_flags);
    if (q.err_dev) (void)hipFree(q.err_dev);
    if (q.err_host) (void)hipHostFree(q.err_host);
    q.raw.release();
    q.mism.release();
    q = rcn_hip_ctx::P2P{};
}

constexpr size_t kP2PHandleBytes = 2 * sizeof(hipIpcMemHandle_t);       // [data buffer | flag array]

int p2p_export(rcn_hip_ctx* c, void* out) {
    RCN_TRY(need_dense(c));
    auto& q = c->p2p;
    if (q.exported) p2p_release(c);
    q.stride = (((size_t)c->nd.P + 1) + 3) & ~(size_t)3;
    // [2 plain slots | 2 slots of self-validating words, 2 * esz bytes per value | the pushed exchange's rows]  (dp_p2p.hpp / dense_p2_dp.hpp / dp_push.hpp)
    q.push_off = 6 * q.stride * c->esz();
    const size_t bytes = q.push_off + push_region_bytes(q.stride);
    // Uncached (fine-grained) device memory for everything a peer reads while a kernel of ours is still running: the words of
    // the in-kernel exchange must leave this GPU's L2 when they are stored, not when the kernel ends -- the allocation type RCCL
    // uses for its own low-latency buffers.  (Ordinary hipMalloc memory is only guaranteed visible to a peer at kernel
    // boundaries; two ranks sharing ONE GPU, the only multi-rank case the development box offers, share its L2 and cannot
    // tell the difference.)  RCN_HIP_DP_CACHED_BUF=1 restores hipMalloc for A/B measurements.
    q.local_bytes = bytes;
    q.local_uncached = !c->opt.dp_cached_buf;
    if (c->opt.dp_cached_buf) HIP_TRY(c, hipMalloc(&q.local_buf, bytes));
    else HIP_TRY(c, uncached_cache().alloc(c->device, bytes, &q.local_buf));
    HIP_TRY(c, uncached_cache().alloc(c->device, kP2PFlagBytes, (void**)&q.local_flags));
    HIP_TRY(c, hipMalloc((void**)&q.err_dev, 256));
    HIP_TRY(c, hipHostMalloc((void**)&q.err_host, 64, hipHostMallocDefault));
    *q.err_host = 0;
    HIP_TRY(c, hipMemset(q.local_buf, 0, bytes));
    HIP_TRY(c, hipMemset(q.local_flags, 0, kP2PFlagBytes));
    HIP_TRY(c, hipMemset(q.err_dev, 0, 256));
    HIP_TRY(c, hipDeviceSynchronize());
    hipIpcMemHandle_t h[2];
    HIP_TRY(c, hipIpcGetMemHandle(&h[0], q.local_buf));
    HIP_TRY(c, hipIpcGetMemHandle(&h[1], q.local_flags));
    std::memcpy(out, h, sizeof h);
    q.exported = true;
    return RCN_HIP_OK;
}

int p2p_attach(rcn_hip_ctx* c, const void* all, int rank, int world) {
    auto& q = c->p2p;
    if (!q.exported) return fail(c, RCN_HIP_ERR_STATE, "p2p_attach: export first");
    if (world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world) return fail(c, RCN_HIP_ERR_INVALID_ARG, "p2p_attach: world must be 1..8");
    c->dp_rank = rank;
    c->dp_world = world;
    q.attached = true;
    for (int r = 0; r < world; ++r) {
        if (r == rank) { q.peer_buf[r] = q.local_buf; q.peer_flags[r] = q.local_flags; continue; }
        hipIpcMemHandle_t h[2];
        std::memcpy(h, (const char*)all + (size_t)r * kP2PHandleBytes, sizeof h);
        HIP_TRY(c, hipIpcOpenMemHandle(&q.peer_buf[r], h[0], hipIpcMemLazyEnablePeerAccess));
        HIP_TRY(c, hipIpcOpenMemHandle((void**)&q.peer_flags[r], h[1], hipIpcMemLazyEnablePeerAccess));
    }
    return RCN_HIP_OK;
}

P2PDesc p2p_desc(const rcn_hip_ctx* c) {
    P2PDesc d{};
    d.world = c->dp_world;
    d.rank = c->dp_rank;
    for (int r = 0; r < kP2PMaxWorld; ++r) { d.buf[r] = c->p2p.peer_buf[r < d.world ? r : 0]; d.flags[r] = c->p2p.peer_flags[r < d.world ? r : 0]; }
    return d;
}

// 1 s of the 100 MHz wall clock; option "dp_timeout_ticks" overrides it (the tests force a tiny one to see the sticky error surface)
static long long p2p_timeout_ticks(const rcn_hip_ctx* c) { return c->opt.dp_timeout_ticks; }

// one all-reduce step on the context's stream; mode 0 applies the update, mode 1 writes the raw sums to p2p.raw
template <typename T>
int p2p_step(rcn_hip_ctx* c, int mode, double scale, void* loss_out, long long timeout) {
    auto& q = c->p2p;
    const unsigned seq = ++q.seq;
    const size_t words = q.stride / P2PWord<T>::per;
    int grid = (int)((words + kP2PThreads - 1) / kP2PThreads);
    if (grid > 96) grid = 96;
    hipLaunchKernelGGL((k_p2p_allreduce<T>), dim3(grid), dim3(kP2PThreads), 0, c->stream, p2p_desc(c), seq, q.stride, c->nd.P, (T*)c->params.p,
                       (T)scale, (T*)loss_out, (T*)q.raw.p, mode, q.err_dev, timeout);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// The data-parallel epoch on the feature-sliced pipeline (dense_p2_dp.hpp): per step k_p2_b, k_p2_dp_grad, k_p2_dp_apply --
// the exchange happens inside the third kernel, which also computes the next batch's partial z_1.
template <typename T>
int enqueue_pipe_steps_dp(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev,
                          bool in_graph, bool fused) {
    auto& q = c->p2p;
    const unsigned* seq_base = in_graph ? q.err_dev + 16 : nullptr;     // set by the caller before each replay
    const NetDesc& nd = c->nd;
    const size_t G = pipe_slices(nd), Cc = nd.dims[nd.L], es = c->esz();
    const double Bg = (double)B * (double)c->dp_world;
    const double scale = eta / Bg, loss_scale = 1.0 / (2.0 * Bg);
    const size_t seg = nb <= pack_segment(c, B) ? nb : pack_segment(c, B);
    auto slot = [&](size_t j) { return ((j / seg) % 2) * seg + j % seg; };
    auto xb = [&](size_t j) { return (const void*)((const char*)c->xpack.p + slot(j) * G * B * 16 * es); };
    auto yb = [&](size_t j) { return (const void*)((const char*)c->ypack.p + slot(j) * B * Cc * es); };
    auto pack = [&](size_t j0) { return launch_pack<T>(c, X, Y, perm, B, j0, (nb - j0 < seg ? nb - j0 : seg), (int)((j0 / seg) % 2), seg); };
    const int grid = (int)G + pipe_extra_wgs(nd);
    const size_t lds = p2_a_lds_elems() * sizeof(T);
    const int n_loss = (int)((B + kPipeTs - 1) / kPipeTs);
    T* a1 = (T*)c->p2buf.p; T* d1 = a1 + B * kP2H; T* d2 = d1 + B * kP2H;
    // the in-kernel exchange form keeps k_p2_b's operand image current too (its tail tiles apply the summed gradient themselves)
    struct FragGuard { rcn_hip_ctx* c; ~FragGuard() { c->frag_on = false; } } frag_guard{c};
    if constexpr (std::is_same<T, float>::value) {
        if (fused && !p2_one_object() && !c->opt.no_fragimg && c->fragimg.p) {
            hipLaunchKernelGGL(k_p2_fragimg, dim3(1), dim3(512), 0, c->stream, c->nd, (const float*)c->params.p, (float*)c->fragimg.p);
            HIP_TRY(c, hipGetLastError());
            c->frag_on = true;
        }
    }
    RCN_TRY(pack(0));
    RCN_TRY(launch_pipe_a<T>(c, xb(0), xb(0), B, 0.0, nullptr, 0.0, false, true));      // partial z_1 of the first batch, current W_0
    for (size_t j = 0; j < nb; ++j) {
        RCN_TRY(launch_pipe_b<T>(c, yb(j), B));
        const bool more = j + 1 < nb;
        if (more && (j + 1) % seg == 0) RCN_TRY(pack(j + 1));
        const unsigned seq = in_graph ? (unsigned)(j + 1) : ++q.seq;       // offset from the base, or the number itself
        T* lj = loss_dev ? (T*)loss_dev + j : nullptr;
        if (fused) {
            hipLaunchKernelGGL((k_p2_dp_fused<T>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (T*)c->params.p, (const T*)xb(j),
                               (const T*)(more ? xb(j + 1) : xb(j)), (int)B, (const T*)a1, (const T*)d1, (const T*)d2, (T)scale, (T*)c->slab.p, (int)G,
                               (const T*)c->loss_part.p, n_loss, (T)loss_scale, lj, more ? 1 : 0, p2p_desc(c), seq_base, seq, q.stride, q.err_dev,
                               p2p_timeout_ticks(c), (T*)c->grad.p, c->frag_on ? (T*)c->fragimg.p : (T*)nullptr);
            HIP_TRY(c, hipGetLastError());
            continue;
        }
        hipLaunchKernelGGL((k_p2_dp_grad<T>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (const T*)xb(j), (int)B, (const T*)a1, (const T*)d1,
                           (const T*)d2, (T*)q.local_buf, q.stride, seq_base, seq, (int)G, (const T*)c->loss_part.p, n_loss, (T)loss_scale);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL((k_p2_dp_apply<T>), dim3(grid), dim3(kDenseThreads), lds, c->stream, nd, (T*)c->params.p, (const T*)(more ? xb(j + 1) : xb(j)),
                           (int)B, (T)scale, (T*)c->slab.p, (int)G, lj, more ? 1 : 0, p2p_desc(c), seq_base, seq, q.stride, q.err_dev, p2p_timeout_ticks(c));
        HIP_TRY(c, hipGetLastError());
    }
    return RCN_HIP_OK;
}

// `iters` exchanges of a known integer pattern; counts wrong sums and reads the timeout word.  Collective.
int p2p_selftest(rcn_hip_ctx* c, int iters, unsigned* mismatches, unsigned* err) {
    auto& q = c->p2p;
    if (!q.attached) return fail(c, RCN_HIP_ERR_STATE, "p2p_selftest: not attached");
    const size_t es = c->esz();
    HIP_TRY(c, q.raw.ensure(q.stride * es));
    HIP_TRY(c, q.mism.ensure(64));
    HIP_TRY(c, hipMemsetAsync(q.mism.p, 0, 64, c->stream));
    for (int it = 0; it < iters; ++it) {
        const unsigned seq = q.seq + 1;
        char* slot = (char*)q.local_buf + (size_t)(seq & 1u) * q.stride * es;
        const long long to = it == 0 ? 10 * p2p_timeout_ticks(c) : p2p_timeout_ticks(c);      // the first exchange absorbs start-up skew
        if (c->dtype == RCN_HIP_F64) {
            hipLaunchKernelGGL((k_p2p_fill<double>), dim3(48), dim3(256), 0, c->stream, (double*)slot, q.stride, c->dp_rank, seq);
            RCN_TRY(p2p_step<double>(c, 1, 0.0, nullptr, to));
            hipLaunchKernelGGL((k_p2p_check<double>), dim3(48), dim3(256), 0, c->stream, (const double*)q.raw.p, q.stride, c->dp_world, seq, (unsigned*)q.mism.p);
        } else {
            hipLaunchKernelGGL((k_p2p_fill<float>), dim3(48), dim3(256), 0, c->stream, (float*)slot, q.stride, c->dp_rank, seq);
            RCN_TRY(p2p_step<float>(c, 1, 0.0, nullptr, to));
            hipLaunchKernelGGL((k_p2p_check<float>), dim3(48), dim3(256), 0, c->stream, (const float*)q.raw.p, q.stride, c->dp_world, seq, (unsigned*)q.mism.p);
        }
        HIP_TRY(c, hipGetLastError());
    }
    unsigned host[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&host[0], q.mism.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&host[1], q.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // The slots keep the last patterns: peers may still be reading them, and nothing depends on their contents -- the
    // gradient kernels overwrite [0, P] every step and the reduce ignores the padding beyond P.
    *mismatches = host[0];
    *err = host[1];
    return RCN_HIP_OK;
}

// the same for the in-kernel exchange of k_p2_dp_fused (self-validating tagged words).  Collective.
int p2p_selftest_fused(rcn_hip_ctx* c, int iters, unsigned* mismatches, unsigned* err) {
    auto& q = c->p2p;
    if (!q.attached) return fail(c, RCN_HIP_ERR_STATE, "p2p_selftest: not attached");
    HIP_TRY(c, q.mism.ensure(64));
    HIP_TRY(c, hipMemsetAsync(q.mism.p, 0, 64, c->stream));
    const int wgs = (int)((q.stride + 255) / 256);
    for (int it = 0; it < iters; ++it) {
        const unsigned seq = ++q.seq;
        if (c->dtype == RCN_HIP_F64)
            hipLaunchKernelGGL((k_p2p_ll_selftest<double>), dim3(wgs), dim3(256), 0, c->stream, p2p_desc(c), seq, q.stride, q.err_dev, p2p_timeout_ticks(c), (unsigned*)q.mism.p);
        else
            hipLaunchKernelGGL((k_p2p_ll_selftest<float>), dim3(wgs), dim3(256), 0, c->stream, p2p_desc(c), seq, q.stride, q.err_dev, p2p_timeout_ticks(c), (unsigned*)q.mism.p);
        HIP_TRY(c, hipGetLastError());
    }
    unsigned host[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&host[0], q.mism.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&host[1], q.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mismatches = host[0];
    *err = host[1];
    return RCN_HIP_OK;
}

// the same for the pushed reduce-scatter + all-gather of the resident kernel's data-parallel form (dp_push.hpp; f32).  Collective.
int p2p_selftest_push(rcn_hip_ctx* c, int iters, unsigned* mismatches, unsigned* err) {
    auto& q = c->p2p;
    if (!q.attached) return fail(c, RCN_HIP_ERR_STATE, "p2p_selftest: not attached");
    HIP_TRY(c, q.mism.ensure(64));
    HIP_TRY(c, hipMemsetAsync(q.mism.p, 0, 64, c->stream));
    const int wgs = (int)((q.stride + 255) / 256);
    for (int it = 0; it < iters; ++it) {
        const unsigned seq = ++q.seq;
        hipLaunchKernelGGL(k_push_selftest, dim3(wgs), dim3(256), 0, c->stream, PushDesc{p2p_desc(c), q.stride, q.push_off}, seq, q.err_dev,
                           it == 0 ? 10 * p2p_timeout_ticks(c) : p2p_timeout_ticks(c), (unsigned*)q.mism.p);
        HIP_TRY(c, hipGetLastError());
    }
    unsigned host[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&host[0], q.mism.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&host[1], q.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mismatches = host[0];
    *err = host[1];
    return RCN_HIP_OK;
}

void drop_img_graphs(rcn_hip_ctx* c) {
    if (c->img_graphs.empty()) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->img_graphs) (void)hipGraphExecDestroy(kv.second);
    c->img_graphs.clear();
}

void drop_graphs(rcn_hip_ctx* c) {
    // a replay may still be running on the stream: destroying its executable under it is a use-after-free
    if (c->stream && !(c->graphs.empty() && c->dp_graphs.empty() && c->img_graphs.empty() && c->step_graphs.empty())) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->step_graphs) (void)hipGraphExecDestroy(kv.second);
    c->step_graphs.clear();
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    for (auto& kv : c->dp_graphs) (void)hipGraphExecDestroy(kv.second);
    c->dp_graphs.clear();
    for (auto& kv : c->img_graphs) (void)hipGraphExecDestroy(kv.second);
    c->img_graphs.clear();
}

}  // namespace

// =====================================================================================================================
extern "C" {

int rcn_hip_abi_version(void) { return RCN_HIP_ABI_VERSION; }

const char* rcn_hip_status_string(int s) {
    switch (s) {
    case RCN_HIP_OK: return "ok";
    case RCN_HIP_ERR_INVALID_ARG: return "invalid argument";
    case RCN_HIP_ERR_SHAPE: return "shape error (the reference panics here)";
    case RCN_HIP_ERR_UNSUPPORTED: return "unsupported (not implemented in the reference, or over a size limit)";
    case RCN_HIP_ERR_HIP: return "HIP runtime error";
    case RCN_HIP_ERR_NO_DEVICE: return "no usable HIP device";
    case RCN_HIP_ERR_STATE: return "call-order error";
    case RCN_HIP_ERR_OOM: return "out of device memory";
    default: return "unknown status";
    }
}

int rcn_hip_create(const rcn_hip_cfg* cfg, rcn_hip_ctx** out) {
    if (!cfg || !out) return RCN_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->struct_size != sizeof(rcn_hip_cfg)) return RCN_HIP_ERR_INVALID_ARG;
    if (cfg->dtype != RCN_HIP_F32 && cfg->dtype != RCN_HIP_F64) return RCN_HIP_ERR_INVALID_ARG;
    if (cfg->n_convpool < 0 || cfg->n_convpool > kMaxConvPool || (cfg->n_convpool > 0 && !cfg->convpool)) return RCN_HIP_ERR_INVALID_ARG;
    if (cfg->n_hidden < 1 || cfg->n_hidden + 1 > kMaxLayers || !cfg->hidden) return RCN_HIP_ERR_INVALID_ARG;   // rcn.rs:444 indexes feedforward_cfg[0]
    if (cfg->classes < 1 || cfg->in_h < 1 || cfg->in_w < 1) return RCN_HIP_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return RCN_HIP_ERR_NO_DEVICE;

    rcn_hip_ctx* c = new (std::nothrow) rcn_hip_ctx();
    if (!c) return RCN_HIP_ERR_OOM;
    *out = c;                                   // handed back even on failure so the caller can read last_error
    c->device = cfg->device;
    c->dtype = cfg->dtype;
    if (const char* e = std::getenv("RCN_HIP_DP_FAULT")) c->dp_fault = e;
    for (const OptDesc& od : kOptTable) {             // the environment seeds the defaults, once, here
        const char* e = std::getenv(od.env);
        if (!e || !*e) continue;
        const long long v = std::atoll(e);
        if (v >= od.lo && v <= od.hi) c->opt.*(od.field) = v;
    }
    RCN_TRY(build_feat_desc(c, cfg));
    // RCN::new itself never fails; a stack whose dense part cannot run in the reference is remembered and
    // reported by the dense entry points (the reference panics inside train, not inside new).
    const long fan = first_layer_fan_in(cfg, c->fd.F);
    if (c->fd.F <= 0)
        c->dense_err = "the conv/pool stack yields an empty feature vector (no Convolve2D layer)";
    else if (fan != c->fd.F)
        c->dense_err = "first-layer fan-in 4^c/2^p*l (rcn.rs:443) = " + std::to_string(fan) + " differs from the flattened feature length " +
                       std::to_string(c->fd.F) + ": the reference panics in gemv (rcn.rs:287)";
    if (c->fd.F <= 0) c->fd.F = 0;
    {
        const int savedF = c->fd.F;
        if (savedF == 0) c->fd.F = 1;           // keep the dense bookkeeping well-formed; dense calls are refused anyway
        const int st = build_net_desc(c, cfg);
        c->fd.F = savedF;
        if (st != RCN_HIP_OK) return st;
    }
    DevGuard g(c->device);
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIP_TRY(c, c->params.ensure((size_t)c->nd.P * c->esz()));
    HIP_TRY(c, hipMemsetAsync(c->params.p, 0, (size_t)c->nd.P * c->esz(), c->stream));
    // (layer stacks whose tile image exceeds LDS run layer by layer on global activations: dense_wide.hpp)
    return RCN_HIP_OK;            // (feature maps that do not fit LDS are staged in global memory: k_features' `spill`)
}

void rcn_hip_destroy(rcn_hip_ctx* c) {
    if (!c) return;
    {
        DevGuard g(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        drop_graphs(c);
        if (c->comm) { (void)rcn::Rccl::get().CommDestroy(c->comm); c->comm = nullptr; }
        p2p_release(c);
        for (DevBuf* b : {&c->slab, &c->xpack, &c->ypack, &c->p2buf, &c->params, &c->acts, &c->deltas, &c->loss_part, &c->grad, &c->xstage, &c->ystage, &c->ostage, &c->scratch0,
                          &c->scratch1, &c->scratch2, &c->redpart, &c->misc})
            b->release();
        c->xcdbuf.release();
        if (c->xerr_host) (void)hipHostFree(c->xerr_host);
        if (c->xerrd) (void)hipFree(c->xerrd);
        for (auto& rs : c->sets) { rs.imgs.release(); rs.X.release(); rs.Y.release(); rs.perm.release(); rs.loss.release(); }
        if (c->pin_host) (void)hipHostFree(c->pin_host);
        c->pll.release();
        if (c->perr_dev) (void)hipFree(c->perr_dev);
        if (c->perr_host) (void)hipHostFree(c->perr_host);
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char* rcn_hip_last_error(const rcn_hip_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rcn_hip_set_stream(rcn_hip_ctx* c, void* s) {
    RCN_TRY(check_ctx(c));
    DevGuard g(c->device);
    drop_graphs(c);
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
    else { HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    return RCN_HIP_OK;
}

int rcn_hip_set_feature_kernel(rcn_hip_ctx* c, int mode) {
    RCN_TRY(check_ctx(c));
    if (mode < 0 || mode > 1) return fail(c, RCN_HIP_ERR_INVALID_ARG, "set_feature_kernel: mode must be 0 or 1");
    c->feat_kernel = mode;
    return RCN_HIP_OK;
}

int rcn_hip_set_dense_path(rcn_hip_ctx* c, int mode) {
    RCN_TRY(check_ctx(c));
    if (mode < 0 || mode > 5)
        return fail(c, RCN_HIP_ERR_INVALID_ARG, "set_dense_path: mode must be 0 (auto), 1 (sample-tile), 2 (feature-sliced, two kernels per step), 5 (feature-sliced, resident "
                                                "one-XCD kernel) -- or 3 / 4, parked experiments of librcn_hip_exp.so");
#ifndef RCN_HIP_EXPERIMENTS
    if (mode == 3 || mode == 4) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "set_dense_path: modes 3 and 4 are parked experiments, compiled only into librcn_hip_exp.so (RCN_HIP_EXPERIMENTS)");
#endif
    if (mode >= 2 && !pipe_supported(c->nd)) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "feature-sliced path needs >= 2 dense layers whose tail fits LDS");
    DevGuard g(c->device);
    if (mode == 5) {
        if (c->dtype != RCN_HIP_F32 || !xcd_supported(c->nd, 256))
            return fail(c, RCN_HIP_ERR_UNSUPPORTED, "set_dense_path(5): the resident one-XCD kernel covers the f32 context, one hidden layer <= 32 (or two: <= 32, <= 16), classes <= 16, "
                                                      "at most 29 feature-slice pairs, batches of 1..256");
        RCN_TRY(xcd_probe(c));
        if (c->xcd_probe != 1)
            return fail(c, RCN_HIP_ERR_UNSUPPORTED, "set_dense_path(5): on this device the blocks with blockIdx.x % 8 == 0 do not share one XCD; the resident kernel "
                                                      "cannot be used");
    }
    drop_graphs(c);
    if (mode != 0 && mode != 5 && c->xerr_host && c->xerr_host[0] != 0 && !c->xcd_dp_used && !c->opt.xcd_auto_fallback) {
        // the recovery the error message names: leaving the resident kernel clears its sticky word (what it had not applied stays lost)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        RCN_TRY(xcd_clear_error(c));
        c->redo.clear();
    }
    if (mode == 5) c->xcd_stepped_down = false;
    c->dense_path = mode;
    return RCN_HIP_OK;
}

int rcn_hip_set_option(rcn_hip_ctx* c, const char* name, int64_t value) {
    RCN_TRY(check_ctx(c));
    if (!name) return fail(c, RCN_HIP_ERR_INVALID_ARG, "set_option: NULL name");
    for (const OptDesc& od : kOptTable)
        if (std::strcmp(od.name, name) == 0) {
            if (value < od.lo || value > od.hi)
                return fail(c, RCN_HIP_ERR_INVALID_ARG, std::string("set_option: ") + name + " must be in " + std::to_string(od.lo) + ".." + std::to_string(od.hi));
            if (c->opt.*(od.field) == (long long)value) return RCN_HIP_OK;
            DevGuard g(c->device);
            drop_graphs(c);                        // captured graphs bake in launch shapes, time-outs and the image's segment length
            if (od.field == &CtxOptions::pack_segment_bytes) c->epoch_nb = 0;
            c->opt.*(od.field) = (long long)value;
            return RCN_HIP_OK;
        }
    return fail(c, RCN_HIP_ERR_INVALID_ARG, std::string("set_option: unknown option '") + name + "'");
}

int rcn_hip_get_option(const rcn_hip_ctx* c, const char* name, int64_t* value) {
    if (!c || !name || !value) return RCN_HIP_ERR_INVALID_ARG;
    for (const OptDesc& od : kOptTable)
        if (std::strcmp(od.name, name) == 0) { *value = (int64_t)(c->opt.*(od.field)); return RCN_HIP_OK; }
    return RCN_HIP_ERR_INVALID_ARG;
}

int rcn_hip_fallbacks_taken(const rcn_hip_ctx* c) { return c ? c->fallbacks_taken : 0; }

int rcn_hip_synchronize(rcn_hip_ctx* c) {
    RCN_TRY(check_ctx(c));
    DevGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return sticky_errors(c);
}

int rcn_hip_feature_len(const rcn_hip_ctx* c, int64_t* out) {
    if (!c || !out) return RCN_HIP_ERR_INVALID_ARG;
    *out = c->fd.F;
    return RCN_HIP_OK;
}
int rcn_hip_num_layers(const rcn_hip_ctx* c) { return c ? c->nd.L : RCN_HIP_ERR_INVALID_ARG; }
int rcn_hip_layer_dims(const rcn_hip_ctx* c, int layer, int32_t* rows, int32_t* cols) {
    if (!c || !rows || !cols || layer < 0 || layer >= c->nd.L) return RCN_HIP_ERR_INVALID_ARG;
    *rows = c->nd.dims[layer + 1]; *cols = c->nd.dims[layer];
    return RCN_HIP_OK;
}
int rcn_hip_param_count(const rcn_hip_ctx* c, int64_t* out) {
    if (!c || !out) return RCN_HIP_ERR_INVALID_ARG;
    *out = c->nd.P;
    return RCN_HIP_OK;
}

// ---------------------------------------------------------------- parameters
int rcn_hip_set_params(rcn_hip_ctx* c, int layer, const double* W, const double* b) {
    RCN_TRY(check_ctx(c));
    RCN_TRY(need_dense(c));
    if (!W || !b || layer < 0 || layer >= c->nd.L) return fail(c, RCN_HIP_ERR_INVALID_ARG, "set_params: bad layer or NULL pointer");
    DevGuard g(c->device);
    const size_t rows = c->nd.dims[layer + 1], cols = c->nd.dims[layer];
    std::vector<double> flat(rows * cols + rows);
    std::memcpy(flat.data(), W, rows * cols * 8);
    std::memcpy(flat.data() + rows * cols, b, rows * 8);
    DevBuf tmp;
    int st = upload(c, tmp, flat.data(), flat.size());
    if (st == RCN_HIP_OK) {
        hipError_t e = hipMemcpyAsync((char*)c->params.p + (size_t)c->nd.w_off[layer] * c->esz(), tmp.p, flat.size() * c->esz(),
                                      hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(c, RCN_HIP_ERR_HIP, hipGetErrorString(e));
    }
    tmp.release();
    if (st == RCN_HIP_OK) {
        c->params_set = true;
        // a recovery action: whatever an earlier resident launch failed to apply is moot now (the stream was drained above)
        if (c->xerr_host && c->xerr_host[0] != 0 && !c->xcd_dp_used) { RCN_TRY(xcd_clear_error(c)); c->xcd_stepped_down = true; }
        c->redo.clear();
    }
    return st;
}

int rcn_hip_get_params(rcn_hip_ctx* c, int layer, double* W, double* b) {
    RCN_TRY(check_ctx(c));
    if (!W || !b || layer < 0 || layer >= c->nd.L) return fail(c, RCN_HIP_ERR_INVALID_ARG, "get_params: bad layer or NULL pointer");
    DevGuard g(c->device);
    const size_t rows = c->nd.dims[layer + 1], cols = c->nd.dims[layer];
    std::vector<double> flat(rows * cols + rows);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    RCN_TRY(sticky_errors(c));                 // parameters of a timed-out call are not handed out as good (a single-GPU resident failure is healed here)
    RCN_TRY(download(c, (char*)c->params.p + (size_t)c->nd.w_off[layer] * c->esz(), flat.data(), flat.size()));
    std::memcpy(W, flat.data(), rows * cols * 8);
    std::memcpy(b, flat.data() + rows * cols, rows * 8);
    return RCN_HIP_OK;
}

int rcn_hip_init_params(rcn_hip_ctx* c, uint64_t seed) {
    RCN_TRY(check_ctx(c));
    RCN_TRY(need_dense(c));
    // get_weight_matrix / get_bias_vector: StandardNormal samples, column-major fill order (rcn.rs:500-523)
    std::mt19937_64 gen(seed ? seed : std::random_device{}());
    std::normal_distribution<double> nrm(0.0, 1.0);
    for (int l = 0; l < c->nd.L; ++l) {
        const size_t rows = c->nd.dims[l + 1], cols = c->nd.dims[l];
        std::vector<double> W(rows * cols), b(rows);
        for (auto& v : W) v = nrm(gen);
        for (auto& v : b) v = nrm(gen);
        RCN_TRY(rcn_hip_set_params(c, l, W.data(), b.data()));
    }
    return RCN_HIP_OK;
}

int rcn_hip_params_dev(rcn_hip_ctx* c, void** p, int64_t* count) {
    if (!c || !p || !count) return RCN_HIP_ERR_INVALID_ARG;
    *p = c->params.p; *count = c->nd.P;
    c->params_set = true;      // the caller may fill the buffer directly (e.g. a DP broadcast)
    if ((c->p2p.err_host && *c->p2p.err_host != 0) || (c->perr_host && *c->perr_host != 0) || (c->xerr_host && *c->xerr_host != 0)) return sticky_errors(c);   // no sync here: last known state
    return RCN_HIP_OK;
}

// ---------------------------------------------------------------- operator API
int rcn_hip_conv_out_shape(int R, int C, int kr, int kc, int padding, int* oR, int* oC) {
    if (!oR || !oC) return RCN_HIP_ERR_INVALID_ARG;
    return conv_shape(R, C, kr, kc, padding, oR, oC);
}
int rcn_hip_pool_out_shape(int R, int C, int padding, int* oR, int* oC) {
    if (!oR || !oC) return RCN_HIP_ERR_INVALID_ARG;
    return pool_shape(R, C, padding, oR, oC);
}

static int grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int rcn_hip_convolve_2d(rcn_hip_ctx* c, const double* m, int n, int R, int C, const double* k, int kr, int kc, int padding, double* out) {
    RCN_TRY(check_ctx(c));
    if (!m || !k || !out || n < 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "convolve_2d: NULL pointer");
    int oR, oC;
    int st = conv_shape(R, C, kr, kc, padding, &oR, &oC);
    if (st != RCN_HIP_OK) return fail(c, st, "convolve_2d expects 'self.shape() >= kernel_shape() > 0' and odd kernels of half-width < 2 under Padding::Same (kernel.rs:123-135,156)");
    if (n == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    const size_t in_b = (size_t)n * R * C * 8, k_b = (size_t)kr * kc * 8, out_b = (size_t)n * oR * oC * 8;
    HIP_TRY(c, c->scratch0.ensure(in_b)); HIP_TRY(c, c->scratch1.ensure(k_b)); HIP_TRY(c, c->scratch2.ensure(out_b));
    HIP_TRY(c, hipMemcpyAsync(c->scratch0.p, m, in_b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->scratch1.p, k, k_b, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_convolve_2d_f64, dim3(grid_for((size_t)n * oR * oC, 256)), dim3(256), 0, c->stream, (const double*)c->scratch0.p, n, R, C,
                       (const double*)c->scratch1.p, kr, kc, padding == RCN_HIP_PAD_SAME ? 1 : 0, oR, oC, (double*)c->scratch2.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->scratch2.p, out_b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_convolve_2d_separated(rcn_hip_ctx* c, const double* m, int n, int R, int C, int op, int padding, double* out) {
    RCN_TRY(check_ctx(c));
    if (!m || !out || n < 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "convolve_2d_separated: NULL pointer");
    if (op < 0 || op > 3) return fail(c, RCN_HIP_ERR_INVALID_ARG, "convolve_2d_separated: bad SeparableOperator");
    if (R < 3 || C < 3) return fail(c, RCN_HIP_ERR_SHAPE, "convolve_2d_separated expects a matrix of at least 3x3 (kernel.rs:199-201)");
    // sobel_separated (kernel.rs:47-52): (3x1 column kernel, 1x3 row kernel)
    static const double cols[4][3] = {{1, 0, -1}, {-1, 0, 1}, {1, 2, 1}, {1, 2, 1}};      // Top, Bottom, Left, Right
    static const double rows[4][3] = {{1, 2, 1}, {1, 2, 1}, {1, 0, -1}, {-1, 0, 1}};
    int r1, c1, r2, c2;
    int st = conv_shape(R, C, 3, 1, padding, &r1, &c1);
    if (st == RCN_HIP_OK) st = conv_shape(r1, c1, 1, 3, padding, &r2, &c2);
    if (st != RCN_HIP_OK) return fail(c, st, "convolve_2d_separated: bad shape / padding");
    if (n == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    const size_t in_b = (size_t)n * R * C * 8, t_b = (size_t)n * r1 * c1 * 8, out_b = (size_t)n * r2 * c2 * 8;
    HIP_TRY(c, c->scratch0.ensure(in_b > out_b ? in_b : out_b)); HIP_TRY(c, c->scratch1.ensure(64)); HIP_TRY(c, c->scratch2.ensure(t_b));
    HIP_TRY(c, hipMemcpyAsync(c->scratch0.p, m, in_b, hipMemcpyHostToDevice, c->stream));
    double kk[6];
    std::memcpy(kk, cols[op], 24); std::memcpy(kk + 3, rows[op], 24);
    HIP_TRY(c, hipMemcpyAsync(c->scratch1.p, kk, 48, hipMemcpyHostToDevice, c->stream));
    const int same = padding == RCN_HIP_PAD_SAME ? 1 : 0;
    // column pass (3x1), row pass (1x3), ReLU -- kernel.rs:204-206
    hipLaunchKernelGGL(k_convolve_2d_f64, dim3(grid_for((size_t)n * r1 * c1, 256)), dim3(256), 0, c->stream, (const double*)c->scratch0.p, n, R, C,
                       (const double*)c->scratch1.p, 3, 1, same, r1, c1, (double*)c->scratch2.p);
    hipLaunchKernelGGL(k_convolve_2d_f64, dim3(grid_for((size_t)n * r2 * c2, 256)), dim3(256), 0, c->stream, (const double*)c->scratch2.p, n, r1, c1,
                       (const double*)c->scratch1.p + 3, 1, 3, same, r2, c2, (double*)c->scratch0.p);
    hipLaunchKernelGGL(k_relu_f64, dim3(grid_for((size_t)n * r2 * c2, 256)), dim3(256), 0, c->stream, (const double*)c->scratch0.p, (size_t)n * r2 * c2,
                       (double*)c->scratch0.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->scratch0.p, out_b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_relu(rcn_hip_ctx* c, const double* m, size_t count, double* out) {
    RCN_TRY(check_ctx(c));
    if ((!m || !out) && count) return fail(c, RCN_HIP_ERR_INVALID_ARG, "relu: NULL pointer");
    if (count == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    HIP_TRY(c, c->scratch0.ensure(count * 8));
    HIP_TRY(c, hipMemcpyAsync(c->scratch0.p, m, count * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_relu_f64, dim3(grid_for(count, 256)), dim3(256), 0, c->stream, (const double*)c->scratch0.p, count, (double*)c->scratch0.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->scratch0.p, count * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_pool_2d(rcn_hip_ctx* c, const double* m, int n, int R, int C, int padding, int pooling, double* out) {
    RCN_TRY(check_ctx(c));
    if (!m || !out || n < 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "pool_2d: NULL pointer");
    if (pooling != RCN_HIP_POOL_AVERAGE && pooling != RCN_HIP_POOL_MAX) return fail(c, RCN_HIP_ERR_INVALID_ARG, "pool_2d: bad Pooling");
    int oR, oC;
    int st = pool_shape(R, C, padding, &oR, &oC);
    if (st != RCN_HIP_OK) return fail(c, st, "stride_2d expected a matrix with dimensions greater than (2, 2) (kernel.rs:246-251)");
    if (pooling != RCN_HIP_POOL_MAX) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "Pooling::Average: Not implemented (kernel.rs:283-285)");
    if (n == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    const size_t in_b = (size_t)n * R * C * 8, out_b = (size_t)n * oR * oC * 8;
    HIP_TRY(c, c->scratch0.ensure(in_b)); HIP_TRY(c, c->scratch2.ensure(out_b));
    HIP_TRY(c, hipMemcpyAsync(c->scratch0.p, m, in_b, hipMemcpyHostToDevice, c->stream));
    // Padding::None truncates odd tails: oR = R/2 so rows/cols >= 2*oR are simply never visited
    hipLaunchKernelGGL(k_pool_2d_f64, dim3(grid_for((size_t)n * oR * oC, 256)), dim3(256), 0, c->stream, (const double*)c->scratch0.p, n, R, C, oR, oC,
                       (double*)c->scratch2.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->scratch2.p, out_b, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

// the default stack conv(Same),pool(Max),conv(Same),pool(Max) on 28x28 input (rcn/src/main.rs:53-59) has specialised kernels
static bool feat_is_cpcp28(const rcn_hip_ctx* c) {
    const FeatDesc& fd = c->fd;
    return c->feat_kernel != 1 && fd.n == 4 && fd.H == 28 && fd.W == 28 && fd.kind[0] == 0 && fd.arg[0] == RCN_HIP_PAD_SAME && fd.kind[1] == 1 &&
           fd.kind[2] == 0 && fd.arg[2] == RCN_HIP_PAD_SAME && fd.kind[3] == 1;
}

// ---------------------------------------------------------------- feature pipeline
int rcn_hip_features_dev(rcn_hip_ctx* c, const uint8_t* imgs, size_t n, void* out, int standardize) {
    RCN_TRY(check_ctx(c));
    if ((!imgs || !out) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "features: NULL pointer");
    if (n == 0 || c->fd.F == 0) return RCN_HIP_OK;          // an empty feature_set flattens to an empty vector (rcn.rs:350)
    if (n > 0x7fffffffULL) return fail(c, RCN_HIP_ERR_INVALID_ARG, "features: too many images in one call");
    DevGuard g(c->device);
    // the default stack on MNIST-shaped input has its own kernel (features.hpp: k_features_cpcp)
    if (feat_is_cpcp28(c) && ((uintptr_t)imgs & 3) == 0) {
        const float rcp = c->dtype == RCN_HIP_F32 && standardize ? fast_standardise_rcp(c) : 0.f;
        // RCN_HIP_FEAT_WAVES=2: two waves per picture (twice the waves per CU on the same LDS).  Measured neutral (144.8 vs 145.8 us per
        // 131 072 pictures): the kernel is not short of waves to hide latency behind, it is short of issue slots -- kept for the record.
        const int two_waves = (int)c->opt.feat_waves;
#define RCN_CPCP(TT, STD, FAST, RCPV)                                                                                              \
    do {                                                                                                                          \
        if (two_waves == 2) {                                                                                                     \
            auto kern = k_features_cpcp<28, 28, TT, STD, FAST, 128>;                                                              \
            hipLaunchKernelGGL(kern, dim3(resident_grid(c, kern, n, 128)), dim3(128), 0, c->stream, imgs, (int)n, (TT*)out,       \
                               (TT)c->mean, (TT)c->sd, (TT)(RCPV));                                                               \
        } else {                                                                                                                  \
            auto kern = k_features_cpcp<28, 28, TT, STD, FAST, 64>;                                                               \
            hipLaunchKernelGGL(kern, dim3(resident_grid(c, kern, n)), dim3(64), 0, c->stream, imgs, (int)n, (TT*)out, (TT)c->mean, \
                               (TT)c->sd, (TT)(RCPV));                                                                            \
        }                                                                                                                         \
    } while (0)
        if (c->dtype == RCN_HIP_F64) {
            if (standardize) RCN_CPCP(double, true, false, 0); else RCN_CPCP(double, false, false, 0);
        } else if (!standardize) RCN_CPCP(float, false, false, 0);
        else if (rcp != 0.f) RCN_CPCP(float, true, true, rcp);
        else RCN_CPCP(float, true, false, 0);
#undef RCN_CPCP
        HIP_TRY(c, hipGetLastError());
        return RCN_HIP_OK;
    }
    const bool wide = c->n_conv > 5;             // |v| <= 255*8^n stays below 2^24 only up to 5 conv layers
    size_t lds = 2 * (size_t)c->fd.max_elems * (wide ? 8 : 4);
    int grid = (int)(n < 4096 ? n : 4096);
    void* spill = nullptr;
    if (lds > 160 * 1024) {
        // the maps of one image do not fit LDS: ping-pong buffers in global memory, one pair per workgroup
        if (grid > 512) grid = 512;
        HIP_TRY(c, c->scratch1.ensure((size_t)grid * lds));
        spill = c->scratch1.p;
        lds = 0;
    }
#define LAUNCH_FEAT(TC, TO)                                                                                                   \
    do {                                                                                                                      \
        RCN_TRY(set_dyn_lds(c, k_features<TC, TO>, lds));                                                                     \
        hipLaunchKernelGGL((k_features<TC, TO>), dim3(grid), dim3(kFeatThreads), lds, c->stream, c->fd, imgs, (int)n, (TO*)out, \
                           standardize, (TO)c->mean, (TO)c->sd, (TC*)spill);                                                  \
    } while (0)
    if (c->dtype == RCN_HIP_F64) { if (wide) LAUNCH_FEAT(double, double); else LAUNCH_FEAT(float, double); }
    else { if (wide) LAUNCH_FEAT(double, float); else LAUNCH_FEAT(float, float); }
#undef LAUNCH_FEAT
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

int rcn_hip_features(rcn_hip_ctx* c, const uint8_t* imgs, size_t n, double* out) {
    RCN_TRY(check_ctx(c));
    if ((!imgs || !out) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "features: NULL pointer");
    if (n == 0 || c->fd.F == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    const size_t img_b = n * (size_t)c->fd.H * c->fd.W, cnt = n * (size_t)c->fd.F;
    HIP_TRY(c, c->xstage.ensure(img_b)); HIP_TRY(c, c->ostage.ensure(cnt * c->esz()));
    HIP_TRY(c, hipMemcpyAsync(c->xstage.p, imgs, img_b, hipMemcpyHostToDevice, c->stream));
    RCN_TRY(rcn_hip_features_dev(c, (const uint8_t*)c->xstage.p, n, c->ostage.p, 0));
    return download(c, c->ostage.p, out, cnt);     // raw features are integers < 2^24: exact in either dtype
}

static int gen_scales_impl(rcn_hip_ctx* c, const void* dev, size_t count, double* mean, double* sd) {
    const int grid = 1024;
    HIP_TRY(c, c->redpart.ensure(grid * sizeof(double)));
    std::vector<double> part(grid);
    auto run = [&](bool sq, double m, double* result) -> int {
        if (c->dtype == RCN_HIP_F64) {
            if (sq) hipLaunchKernelGGL((k_reduce<double, true>), dim3(grid), dim3(256), 0, c->stream, (const double*)dev, count, m, (double*)c->redpart.p);
            else hipLaunchKernelGGL((k_reduce<double, false>), dim3(grid), dim3(256), 0, c->stream, (const double*)dev, count, m, (double*)c->redpart.p);
        } else {
            if (sq) hipLaunchKernelGGL((k_reduce<float, true>), dim3(grid), dim3(256), 0, c->stream, (const float*)dev, count, m, (double*)c->redpart.p);
            else hipLaunchKernelGGL((k_reduce<float, false>), dim3(grid), dim3(256), 0, c->stream, (const float*)dev, count, m, (double*)c->redpart.p);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(part.data(), c->redpart.p, grid * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double t = 0.0;
        for (int i = 0; i < grid; ++i) t += part[i];
        *result = t;
        return RCN_HIP_OK;
    };
    double s = 0.0, q = 0.0;
    RCN_TRY(run(false, 0.0, &s));
    const double mu = s / (double)count;                      // rcn.rs:240
    RCN_TRY(run(true, mu, &q));
    const double sdv = std::sqrt(q / (double)count);          // rcn.rs:247
    if (mu != c->mean || sdv != c->sd) drop_img_graphs(c);    // captured feature launches carry the old scale_set by value
    c->mean = mu; c->sd = sdv;                                // rcn.rs:249-250
    if (mean) *mean = mu;
    if (sd) *sd = sdv;
    return RCN_HIP_OK;
}

int rcn_hip_gen_scales_dev(rcn_hip_ctx* c, const void* feats, size_t n, double* mean, double* sd) {
    RCN_TRY(check_ctx(c));
    if (!feats || n == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "gen_scales: empty input (the reference indexes iv[0], rcn.rs:233)");
    DevGuard g(c->device);
    return gen_scales_impl(c, feats, n * (size_t)c->fd.F, mean, sd);
}

int rcn_hip_gen_scales(rcn_hip_ctx* c, const double* feats, size_t n, double* mean, double* sd) {
    RCN_TRY(check_ctx(c));
    if (!feats || n == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "gen_scales: empty input (the reference indexes iv[0], rcn.rs:233)");
    DevGuard g(c->device);
    // statistics are taken in f64 on the device regardless of the ctx dtype so that raw (integer) features lose nothing
    const size_t cnt = n * (size_t)c->fd.F;
    HIP_TRY(c, c->xstage.ensure(cnt * 8));
    HIP_TRY(c, hipMemcpyAsync(c->xstage.p, feats, cnt * 8, hipMemcpyHostToDevice, c->stream));
    const int saved = c->dtype;
    c->dtype = RCN_HIP_F64;
    const int st = gen_scales_impl(c, c->xstage.p, cnt, mean, sd);
    c->dtype = saved;
    return st;
}

int rcn_hip_set_scale(rcn_hip_ctx* c, double mean, double sd) {
    RCN_TRY(check_ctx(c));
    if (mean != c->mean || sd != c->sd) {
        // the graphs of rcn_hip_train_epoch_images_dev hold (mean, sd, reciprocal, kernel variant) by value: a replay after
        // this call would standardise with the old scale_set (rcn_hip.h promises the current one)
        DevGuard g(c->device);
        drop_img_graphs(c);
    }
    c->mean = mean; c->sd = sd;
    return RCN_HIP_OK;
}
int rcn_hip_get_scale(const rcn_hip_ctx* c, double* mean, double* sd) {
    if (!c || !mean || !sd) return RCN_HIP_ERR_INVALID_ARG;
    *mean = c->mean; *sd = c->sd;
    return RCN_HIP_OK;
}

int rcn_hip_standardize_dev(rcn_hip_ctx* c, void* feats, size_t count) {
    RCN_TRY(check_ctx(c));
    if (!feats && count) return fail(c, RCN_HIP_ERR_INVALID_ARG, "standardize: NULL pointer");
    if (count == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    if (c->dtype == RCN_HIP_F64)
        hipLaunchKernelGGL((k_standardize<double>), dim3(grid_for(count, 256)), dim3(256), 0, c->stream, (double*)feats, count, c->mean, c->sd);
    else
        hipLaunchKernelGGL((k_standardize<float>), dim3(grid_for(count, 256)), dim3(256), 0, c->stream, (float*)feats, count, (float)c->mean, (float)c->sd);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

int rcn_hip_standardize(rcn_hip_ctx* c, double* feats, size_t count) {
    RCN_TRY(check_ctx(c));
    if (!feats && count) return fail(c, RCN_HIP_ERR_INVALID_ARG, "standardize: NULL pointer");
    if (count == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    // host-buffer form works in f64 on the device whatever the ctx dtype (the caller's data is f64)
    HIP_TRY(c, c->xstage.ensure(count * 8));
    HIP_TRY(c, hipMemcpyAsync(c->xstage.p, feats, count * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL((k_standardize<double>), dim3(grid_for(count, 256)), dim3(256), 0, c->stream, (double*)c->xstage.p, count, c->mean, c->sd);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(feats, c->xstage.p, count * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

// ---------------------------------------------------------------- dense network
int rcn_hip_train_batch_dev(rcn_hip_ctx* c, const void* x, const void* y, size_t B, double eta, void* loss_dev) {
    RCN_TRY(check_ctx(c));
    if (!x || !y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_batch: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_batch: batch size must be in 1..2^30 (eta / 0 in the reference)");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    return enqueue_train_step(c, x, y, nullptr, B, eta, loss_dev);
}

int rcn_hip_train_batch(rcn_hip_ctx* c, const double* x, const double* y, size_t B, double eta, double* loss_out) {
    RCN_TRY(check_ctx(c));
    if (!x || !y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_batch: NULL pointer");
    if (B == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_batch: empty batch");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(upload(c, c->xstage, x, B * (size_t)c->nd.dims[0]));
    RCN_TRY(upload(c, c->ystage, y, B * (size_t)c->nd.dims[c->nd.L]));
    HIP_TRY(c, c->misc.ensure(64));
    RCN_TRY(rcn_hip_train_batch_dev(c, c->xstage.p, c->ystage.p, B, eta, loss_out ? c->misc.p : nullptr));
    if (loss_out) return download(c, c->misc.p, loss_out, 1);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

static int epoch_impl(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool launch,
                      bool from_images = false) {
    RCN_TRY(check_ctx(c));
    if (!X || !Y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_epoch: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_epoch: batch size must be in 1..2^30");
    if (nb == 0) return RCN_HIP_OK;                 // chunks_exact yields nothing (rcn.rs:147)
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(xcd_entry_check(c));
    RCN_TRY(ensure_dense_ws(c, B));
    if (use_pipe(c, B)) { RCN_TRY(ensure_pipe_ws(c, B)); RCN_TRY(ensure_pack_ws(c, B, nb)); }
    const bool step = use_pipe(c, B) && use_step(c, B);
#ifdef RCN_HIP_EXPERIMENTS
    if (step) RCN_TRY(ensure_step_ws(c, B));
#endif
    if (from_images && !(use_pipe(c, B) && feat_is_cpcp28(c)))
        return fail(c, RCN_HIP_ERR_UNSUPPORTED, "train_epoch_images: needs the default conv/pool stack on 28x28 input and a layer stack / batch size the "
                                                  "feature-sliced pipeline covers; use rcn_hip_features_dev + rcn_hip_train_epoch_dev otherwise");
    if (use_pipe(c, B) && use_xcd(c, B)) {
        // one resident kernel per segment of the epoch image, all of its workgroups on one XCD (dense_xcd.hpp): nothing to capture
        RCN_TRY(ensure_xcd_ws(c, B));
        if (!launch) return RCN_HIP_OK;
        return enqueue_xcd_epoch(c, X, Y, perm, B, nb, eta, loss_dev, from_images, false, 0, 0);
    }
#ifdef RCN_HIP_EXPERIMENTS
    if (use_persist(c, B) && !from_images) {
        // no graph: one resident kernel per segment of the epoch image runs all of its steps
        if (!launch) return RCN_HIP_OK;
        return enqueue_persist_epoch(c, X, Y, perm, B, nb, eta, loss_dev);
    }
#endif
    // LDS attributes are per kernel variant and cached (set_dyn_lds); hipFuncSetAttribute is not a stream operation,
    // so the first capture of a variant may set it while capturing.

    auto& cache = from_images ? c->img_graphs : c->graphs;
    const EpochKey key{X, Y, perm, B, nb, eta, loss_dev};
    auto it = cache.find(key);
    if (it == cache.end()) {
        const size_t F = c->nd.dims[0], Cc = c->nd.dims[c->nd.L], es = c->esz();
        hipGraph_t graph = nullptr;
        HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int st = RCN_HIP_OK;
#ifdef RCN_HIP_EXPERIMENTS
        if (step) {
            st = enqueue_step_epoch(c, X, Y, perm, B, nb, eta, loss_dev, from_images);
        } else
#endif
        if (use_pipe(c, B)) {
            st = c->dtype == RCN_HIP_F64 ? enqueue_pipe_steps<double>(c, X, Y, perm, B, nb, eta, loss_dev, from_images)
                                         : enqueue_pipe_steps<float>(c, X, Y, perm, B, nb, eta, loss_dev, from_images);
        } else
        for (size_t j = 0; j < nb && st == RCN_HIP_OK; ++j) {
            const void* xb = perm ? X : (const char*)X + j * B * F * es;
            const void* yb = perm ? Y : (const char*)Y + j * B * Cc * es;
            const int32_t* ib = perm ? perm + j * B : nullptr;
            void* lj = loss_dev ? (char*)loss_dev + j * es : nullptr;
            st = enqueue_train_step(c, xb, yb, ib, B, eta, lj);
        }
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (st != RCN_HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
        HIP_TRY(c, e);
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(c, e);
        if (cache.size() >= 16) drop_graphs(c);
        it = cache.emplace(key, exec).first;
    }
    if (launch) HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_train_epoch_images_dev(rcn_hip_ctx* c, const uint8_t* imgs, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev) {
    return epoch_impl(c, imgs, Y, perm, B, nb, eta, loss_dev, true, true);
}

int rcn_hip_prepare_epoch_images_dev(rcn_hip_ctx* c, const uint8_t* imgs, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev) {
    return epoch_impl(c, imgs, Y, perm, B, nb, eta, loss_dev, false, true);
}

int rcn_hip_train_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev) {
    return epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, true);
}

int rcn_hip_prepare_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev) {
    return epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, false);
}

// ---- one epoch of RCN::train as the reference structures it: shuffle once (rcn.rs:146), then walk the chunks (rcn.rs:147-149) ----
// begin: the shuffled order is materialised ONCE as the slice-major epoch image (k_pack_epoch, or the fused feature kernel from u8
// pictures); steps: train_batch over batches j0 .. j0+n of that image, any number of calls, no re-packing.
static int epoch_begin_impl(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, bool from_images) {
    RCN_TRY(check_ctx(c));
    if (!X || !Y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "epoch_begin: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "epoch_begin: batch size must be in 1..2^30");
    RCN_TRY(need_dense(c));
    c->epoch_nb = 0;
    if (nb == 0) return RCN_HIP_OK;
    if (!use_pipe(c, B) || (from_images && !feat_is_cpcp28(c)))
        return fail(c, RCN_HIP_ERR_UNSUPPORTED, "epoch_begin: this layer stack / batch size does not run on the feature-sliced pipeline (or, from images, the "
                                                  "conv/pool stack is not the default one on 28x28); use rcn_hip_train_epoch_dev");
    const size_t seg = nb <= pack_segment(c, B) ? nb : pack_segment(c, B);
    if (nb > 2 * seg)
        return fail(c, RCN_HIP_ERR_UNSUPPORTED, "epoch_begin: the epoch image holds at most " + std::to_string(2 * seg) + " batches of this size; use "
                                                  "rcn_hip_train_epoch_dev, which re-packs segment by segment");
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    RCN_TRY(ensure_pipe_ws(c, B));
    RCN_TRY(ensure_pack_ws(c, B, nb));
    for (size_t j = 0; j < nb; j += seg) {
        const size_t n = nb - j < seg ? nb - j : seg;
        const int half = (int)((j / seg) % 2);
        if (from_images)
            RCN_TRY(c->dtype == RCN_HIP_F64 ? launch_feat_pack<double>(c, (const uint8_t*)X, Y, perm, B, j, n, half, seg)
                                            : launch_feat_pack<float>(c, (const uint8_t*)X, Y, perm, B, j, n, half, seg));
        else
            RCN_TRY(c->dtype == RCN_HIP_F64 ? launch_pack<double>(c, X, Y, perm, B, j, n, half, seg) : launch_pack<float>(c, X, Y, perm, B, j, n, half, seg));
    }
    c->epoch_B = B; c->epoch_nb = nb; c->epoch_seg = seg;
    if (!c->replaying) {
        c->last_begin = rcn_hip_ctx::BeginRec{};
        c->last_begin.X = X; c->last_begin.Y = Y; c->last_begin.perm = perm; c->last_begin.B = B; c->last_begin.nb = nb; c->last_begin.from_images = from_images;
        c->last_begin.valid = true;
        c->last_begin.src = perm_source_of(c, perm);
    }
    return RCN_HIP_OK;
}

int rcn_hip_epoch_begin_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb) {
    return epoch_begin_impl(c, X, Y, perm, B, nb, false);
}

int rcn_hip_epoch_begin_images_dev(rcn_hip_ctx* c, const uint8_t* imgs, const void* Y, const int32_t* perm, size_t B, size_t nb) {
    return epoch_begin_impl(c, imgs, Y, perm, B, nb, true);
}

static int epoch_steps_impl(rcn_hip_ctx* c, size_t j0, size_t n, double eta, void* loss_dev, bool launch) {
    RCN_TRY(check_ctx(c));
    if (c->epoch_nb == 0) return fail(c, RCN_HIP_ERR_STATE, "epoch_steps: no epoch begun (rcn_hip_epoch_begin_dev), or another training call has re-packed the image since");
    if (j0 > c->epoch_nb || n > c->epoch_nb - j0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "epoch_steps: batches beyond the begun epoch");
    if (n == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(xcd_entry_check(c));
    if (c->epoch_nb == 0) return fail(c, RCN_HIP_ERR_STATE, "epoch_steps: the begun epoch did not survive the step-down from the resident kernel");
    const size_t B = c->epoch_B, nb_epoch = c->epoch_nb, seg = c->epoch_seg;
    RCN_TRY(ensure_dense_ws(c, B));
    RCN_TRY(ensure_pipe_ws(c, B));
    if (use_xcd(c, B)) {
        RCN_TRY(ensure_xcd_ws(c, B));
        if (!launch) return RCN_HIP_OK;
        return enqueue_xcd_epoch(c, nullptr, nullptr, nullptr, B, n, eta, loss_dev, false, true, j0, seg);
    }
    const EpochKey key{c->xpack.p, c->ypack.p, nullptr, B, n, eta, loss_dev, j0 + 1 + (seg << 32)};
    auto it = c->step_graphs.find(key);
    if (it == c->step_graphs.end()) {
        hipGraph_t graph = nullptr;
        HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int st = c->dtype == RCN_HIP_F64 ? enqueue_pipe_steps<double>(c, nullptr, nullptr, nullptr, B, n, eta, loss_dev, false, true, j0, seg)
                                               : enqueue_pipe_steps<float>(c, nullptr, nullptr, nullptr, B, n, eta, loss_dev, false, true, j0, seg);
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        c->epoch_B = B; c->epoch_nb = nb_epoch; c->epoch_seg = seg;
        if (st != RCN_HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
        HIP_TRY(c, e);
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(c, e);
        if (c->step_graphs.size() >= 64) { drop_graphs(c); }
        it = c->step_graphs.emplace(key, exec).first;
    }
    if (launch) HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_epoch_steps_dev(rcn_hip_ctx* c, size_t first_batch, size_t n_batches, double eta, void* loss_dev) {
    return epoch_steps_impl(c, first_batch, n_batches, eta, loss_dev, true);
}

int rcn_hip_prepare_epoch_steps_dev(rcn_hip_ctx* c, size_t first_batch, size_t n_batches, double eta, void* loss_dev) {
    return epoch_steps_impl(c, first_batch, n_batches, eta, loss_dev, false);
}

int rcn_hip_shuffle_dev(rcn_hip_ctx* c, int32_t* perm, size_t n, size_t passes, uint64_t seed) {
    RCN_TRY(check_ctx(c));
    if (!perm || n == 0 || n > 0x40000000ULL || passes == 0 || n * passes > 0xffffffffULL) return fail(c, RCN_HIP_ERR_INVALID_ARG, "shuffle: bad argument");
    DevGuard g(c->device);
    int bits = 2;
    while (((size_t)1 << bits) < n) bits += 2;                 // even number of bits: balanced Feistel halves
    const size_t total = n * passes;
    hipLaunchKernelGGL(k_shuffle_indices, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (int*)perm, (unsigned)n, (unsigned)passes,
                       (unsigned long long)seed, bits / 2);
    HIP_TRY(c, hipGetLastError());
    {
        rcn_hip_ctx::PermSource ps;
        ps.kind = 1; ps.buf = perm; ps.n = n; ps.passes = passes; ps.seed = seed;
        note_perm_source(c, std::move(ps));
    }
    return RCN_HIP_OK;
}

static int batch_gradient_impl(rcn_hip_ctx* c, const void* x, const void* y, const int32_t* idx, size_t B, void* grad, void* loss_sum) {
    RCN_TRY(check_ctx(c));
    if (!x || !y || !grad) return fail(c, RCN_HIP_ERR_INVALID_ARG, "batch_gradient: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "batch_gradient: batch size must be in 1..2^30");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    if (c->dtype == RCN_HIP_F64) {
        RCN_TRY(launch_fwd<double>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<double>(c, false, x, idx, B, 0.0, grad, loss_sum, 1.0));
    } else {
        RCN_TRY(launch_fwd<float>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<float>(c, false, x, idx, B, 0.0, grad, loss_sum, 1.0));
    }
    return RCN_HIP_OK;
}

int rcn_hip_batch_gradient_dev(rcn_hip_ctx* c, const void* x, const void* y, size_t B, void* grad, void* loss_sum) {
    return batch_gradient_impl(c, x, y, nullptr, B, grad, loss_sum);
}

int rcn_hip_batch_gradient_perm_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, void* grad, void* loss_sum) {
    if (!perm) return fail(c, RCN_HIP_ERR_INVALID_ARG, "batch_gradient_perm: NULL index pointer");
    return batch_gradient_impl(c, X, Y, perm, B, grad, loss_sum);
}

int rcn_hip_apply_gradient_dev(rcn_hip_ctx* c, const void* grad, double scale) {
    RCN_TRY(check_ctx(c));
    if (!grad) return fail(c, RCN_HIP_ERR_INVALID_ARG, "apply_gradient: NULL pointer");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    const int n = c->nd.P;
    if (c->dtype == RCN_HIP_F64)
        hipLaunchKernelGGL((k_apply_gradient<double>), dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (double*)c->params.p, (const double*)grad, scale, n);
    else
        hipLaunchKernelGGL((k_apply_gradient<float>), dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (float*)c->params.p, (const float*)grad, (float)scale, n);
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// ---------------------------------------------------------------- data-parallel training over RCCL
// The sum over samples of rcn.rs:190-205 is split over ranks: every rank computes the summed gradient of its shard of
// each global batch, ONE ncclAllReduce(sum) of the flat gradient (+ the loss in its last element) combines them over
// xGMI, and every rank applies the identical update W <- W - (eta / B_global) * sum dW (rcn.rs:214,221 with the global
// batch length), so replicas stay bit-identical without a broadcast.  The whole loop is enqueued from here on the
// context's stream -- no host round trip, no Python between steps.
#define NCCL_TRY(ctx, expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess)                                                                               \
            return fail(ctx, RCN_HIP_ERR_HIP, std::string(#expr) + ": " + rcn::Rccl::get().GetErrorString(r_)); \
    } while (0)

// ---- admission of the peer exchange -----------------------------------------------------------------------------------
// Sets up the peer-read all-reduce for a group of ranks, and keeps it only if EVERY rank could map every peer and a known-answer
// exchange came back exact on every rank; then asks the same of the in-kernel (tagged-word) form.  Every rank executes the same
// sequence of collectives whatever fails locally (a failure only lowers its vote), so a rank that cannot use xGMI peer reads makes
// the whole group stay on the previous form instead of deadlocking it.  The collectives come from a transport: RCCL on the
// communicator of rcn_hip_dp_init, or two caller-supplied callbacks (rcn_hip_dp_p2p_admit: any out-of-band channel).
//
// Outcome (identical on every rank):  0 = no peer exchange (the loop uses ncclAllReduce, or the caller's own all-reduce),
// 1 = peer exchange at kernel boundaries, 2 = peer exchange inside the gradient kernel.
//
// Fault injection for the tests, RCN_HIP_DP_FAULT="<stage>:<rank>[,<stage>:<rank>...]": the named rank behaves as if that stage had
// failed locally -- export | attach | kat (known-answer mismatch) | ll (tagged-word self-test mismatch) | llskip (the rank never
// launches its side of the tagged-word self-test, so its peers really time out).
struct P2PTransport {
    std::function<int(const void* mine, void* all, size_t bytes_per_rank)> allgather;     // host buffers, rank order
    std::function<int(int& v)> vote_min;                                                    // v <- min over ranks
};

static bool p2p_fault(const rcn_hip_ctx* c, const char* stage) {
    if (c->dp_fault.empty()) return false;
    const std::string want = std::string(stage) + ":" + std::to_string(c->dp_rank);
    const std::string& all = c->dp_fault;
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t end = all.find(',', pos);
        if (all.substr(pos, end == std::string::npos ? std::string::npos : end - pos) == want) return true;
        if (end == std::string::npos) break;
        pos = end + 1;
    }
    return false;
}

// The vote sequence itself, separated from what the stages do: `ops` is the context's device work (p2p_admission below) or a script
// (rcn_hip_dp_admission_rehearse: the same sequence over the caller's transport without any GPU, which is how the CPU test drives
// the native logic over gloo).  Every stage returns 1 (this rank is fine) or 0; every rank calls the transport the same number of times
// in the same order whatever its own stages returned.
struct AdmissionOps {
    std::function<int(char* handles)> do_export;            // export this rank's buffers -> handles
    std::function<int(const char* all_handles)> attach;      // map every peer
    std::function<int()> known_answer;                       // the kernel-boundary exchange, exact sums, no timeout
    std::function<int()> wants_fused;                        // configuration: may the exchange run inside a step kernel at all?
    std::function<int()> tagged_words;                       // the in-kernel (pull) exchange's known-answer test
    std::function<int()> wants_push;                         // configuration: f32 context?
    std::function<int()> pushed_words;                       // the pushed reduce-scatter + all-gather's known-answer test
    std::function<int()> clear_sticky;                       // after a failed in-kernel stage: 1 if the kernel-boundary form survives
};
struct AdmissionOutcome { bool on = false, fused = false, push = false; };

static int admission_protocol(int world, const AdmissionOps& ops, const P2PTransport& t, AdmissionOutcome& out) {
    char mine[kP2PHandleBytes] = {};
    int ok = ops.do_export(mine);
    std::vector<char> all((size_t)world * kP2PHandleBytes);
    int st = RCN_HIP_OK;
    do {
        if ((st = t.allgather(mine, all.data(), kP2PHandleBytes)) != RCN_HIP_OK) break;
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank export?
        if (!ok) break;
        ok = ops.attach(all.data());
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank map every peer?
        if (!ok) break;
        ok = ops.known_answer();
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank see exact sums, without a timeout?
        if (!ok) break;
        out.on = true;
        // second, independent question: may the exchange also run INSIDE the gradient kernel (tagged words, no flags)?  A failed
        // wait here leaves the sticky error word set, which would disable the kernel-boundary protocol too, so it is cleared
        // (after every rank has drained: the vote synchronises) when only this stage failed.
        int okf = ops.wants_fused();
        if ((st = t.vote_min(okf)) != RCN_HIP_OK) break;         // every rank must want it (same configuration everywhere, normally)
        if (!okf) break;
        okf = ops.tagged_words();
        if ((st = t.vote_min(okf)) != RCN_HIP_OK) break;
        if (okf) out.fused = true;
        else {
            // (every rank is here -- the vote above gave all of them the same answer -- so clearing is voted too: a rank that cannot
            // clear its sticky word takes the whole group off the peer exchange, not only itself.  Found by the CPU rehearsal of this
            // sequence, tests/test_dp_gloo.py: rounds 1-2 decided this locally and the ranks could land on different forms.)
            int okc = ops.clear_sticky();
            if ((st = t.vote_min(okc)) != RCN_HIP_OK) break;
            if (!okc) out.on = false;
        }
        if (!out.on || !out.fused) break;
        // third question, asked only of a group that passed everything before it: the pushed reduce-scatter + all-gather the resident
        // kernel's data-parallel form runs (f32 contexts; remote STORES into the peers' memory and local polls, where the two forms
        // above only ever read a peer's memory)
        int okp = ops.wants_push();
        if ((st = t.vote_min(okp)) != RCN_HIP_OK) break;
        if (!okp) break;
        okp = ops.pushed_words();
        if ((st = t.vote_min(okp)) != RCN_HIP_OK) break;
        if (okp) out.push = true;
        else {
            int okc = ops.clear_sticky();
            if ((st = t.vote_min(okc)) != RCN_HIP_OK) break;
            if (!okc) { out.on = false; out.fused = false; }
        }
    } while (0);
    return st;
}

static int p2p_admission(rcn_hip_ctx* c, const P2PTransport& t) {
    const int world = c->dp_world, rank = c->dp_rank;
    AdmissionOps ops;
    ops.do_export = [&](char* h) { return (p2p_export(c, h) == RCN_HIP_OK && !p2p_fault(c, "export")) ? 1 : 0; };
    ops.attach = [&](const char* all) { return (p2p_attach(c, all, rank, world) == RCN_HIP_OK && !p2p_fault(c, "attach")) ? 1 : 0; };
    ops.known_answer = [&]() {
        unsigned bad = 0, err = 0;
        return (p2p_selftest(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "kat")) ? 1 : 0;
    };
    ops.wants_fused = [&]() { return c->opt.dp_fused ? 1 : 0; };
    ops.tagged_words = [&]() {
        unsigned bad = 0, err = 0;
        if (p2p_fault(c, "llskip")) { c->p2p.seq += 16; return 0; }          // this rank stays silent: its peers' waits expire
        return (p2p_selftest_fused(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "ll")) ? 1 : 0;
    };
    ops.wants_push = [&]() { return (c->opt.dp_fused && c->dtype == RCN_HIP_F32) ? 1 : 0; };
    ops.pushed_words = [&]() {
        unsigned bad = 0, err = 0;
        if (p2p_fault(c, "pushskip")) { c->p2p.seq += 16; return 0; }
        return (p2p_selftest_push(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "push")) ? 1 : 0;
    };
    ops.clear_sticky = [&]() {
        // every rank has drained (the vote synchronised them); clear the sticky word and the pinned mirror of it
        const bool ok = hipMemsetAsync(c->p2p.err_dev, 0, 4, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        if (c->p2p.err_host) *c->p2p.err_host = 0;
        return ok ? 1 : 0;
    };
    AdmissionOutcome out;
    const int st = admission_protocol(world, ops, t, out);
    c->p2p.on = out.on; c->p2p.fused = out.on && out.fused; c->p2p.push = out.on && out.push;
    if (!c->p2p.on) { const int rk = c->dp_rank, w = c->dp_world; p2p_release(c); c->dp_rank = rk; c->dp_world = w; }
    c->err.clear();                                              // a failed attempt is not an error: the loop runs on the previous form
    return st;
}

static int p2p_bootstrap_over_rccl(rcn_hip_ctx* c) {
    rcn::Rccl& r = rcn::Rccl::get();
    const int world = c->dp_world;
    DevBuf xch;
    HIP_TRY(c, xch.ensure((size_t)(world + 1) * kP2PHandleBytes + 64));
    char* d_all = (char*)xch.p;
    char* d_mine = d_all + (size_t)world * kP2PHandleBytes;
    int* d_vote = (int*)(d_mine + kP2PHandleBytes);
    P2PTransport t;
    t.vote_min = [&](int& v) -> int {
        HIP_TRY(c, hipMemcpyAsync(d_vote, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(c, r.AllReduce(d_vote, d_vote, 1, ncclInt, ncclMin, c->comm, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&v, d_vote, sizeof v, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RCN_HIP_OK;
    };
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int {
        HIP_TRY(c, hipMemcpyAsync(d_mine, mine, bytes, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(c, r.AllGather(d_mine, d_all, bytes, ncclChar, c->comm, c->stream));
        HIP_TRY(c, hipMemcpyAsync(all, d_all, (size_t)world * bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RCN_HIP_OK;
    };
    const int st = p2p_admission(c, t);
    xch.release();
    return st;
}

int rcn_hip_dp_unique_id(void* id_out) {
    if (!id_out) return RCN_HIP_ERR_INVALID_ARG;
    rcn::Rccl& r = rcn::Rccl::get();
    if (!r.ok) return RCN_HIP_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == RCN_HIP_DP_ID_BYTES, "rcn_hip.h: RCN_HIP_DP_ID_BYTES");
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return RCN_HIP_ERR_HIP;
    std::memcpy(id_out, &id, sizeof id);
    return RCN_HIP_OK;
}

int rcn_hip_dp_init(rcn_hip_ctx* c, const void* id_bytes, int rank, int world) {
    RCN_TRY(check_ctx(c));
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_init: bad id / rank / world");
    rcn::Rccl& r = rcn::Rccl::get();
    if (!r.ok) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "dp_init: " + r.err);
    DevGuard g(c->device);
    if (c->comm) { HIP_TRY(c, hipStreamSynchronize(c->stream)); NCCL_TRY(c, r.CommDestroy(c->comm)); c->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    NCCL_TRY(c, r.CommInitRank(&c->comm, world, id, rank));
    c->dp_rank = rank;
    c->dp_world = world;
    p2p_release(c);
    const bool force = c->opt.dp_p2p == 2;            // 2: also at world size 1 (exercises the whole set-up path on one GPU)
    if ((world > 1 || force) && world <= kP2PMaxWorld && c->opt.dp_p2p != 0 && c->dense_err.empty()) RCN_TRY(p2p_bootstrap_over_rccl(c));
    return RCN_HIP_OK;
}

// A peer wait that expired inside the resident kernel's data-parallel form is a property of the group just torn down (a peer that
// left), not of this context's single-GPU resident path: reported by dp_finalize, then cleared with the group.
static void clear_xcd_dp_timeout(rcn_hip_ctx* c) {
    if (c->xcd_dp_used && c->xerr_host && *c->xerr_host == 1u) (void)xcd_clear_error(c);
    c->xcd_dp_used = false;
}

int rcn_hip_dp_finalize(rcn_hip_ctx* c) {
    RCN_TRY(check_ctx(c));
    DevGuard g(c->device);
    if (!c->comm) {
        if (c->p2p.exported) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const int sticky = sticky_errors(c);
            const std::string sticky_msg = c->err;
            p2p_release(c); c->dp_rank = 0; c->dp_world = 1;
            clear_xcd_dp_timeout(c);
            if (sticky != RCN_HIP_OK) return fail(c, sticky, sticky_msg);
        }
        return RCN_HIP_OK;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int sticky = sticky_errors(c);        // reported, but the group is torn down all the same
    const std::string sticky_msg = c->err;
    NCCL_TRY(c, rcn::Rccl::get().CommDestroy(c->comm));
    c->comm = nullptr;
    p2p_release(c);
    c->dp_rank = 0;
    c->dp_world = 1;
    clear_xcd_dp_timeout(c);
    if (sticky != RCN_HIP_OK) return fail(c, sticky, sticky_msg);
    return RCN_HIP_OK;
}

/* ---- the peer all-reduce without RCCL: explicit handle exchange (what rcn_hip_dp_init does internally over RCCL) ---- */
int rcn_hip_dp_p2p_export(rcn_hip_ctx* c, void* handles_out) {
    RCN_TRY(check_ctx(c));
    if (!handles_out) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_export: NULL pointer");
    static_assert(kP2PHandleBytes == RCN_HIP_DP_P2P_HANDLE_BYTES, "rcn_hip.h: RCN_HIP_DP_P2P_HANDLE_BYTES");
    DevGuard g(c->device);
    return p2p_export(c, handles_out);
}

int rcn_hip_dp_p2p_attach(rcn_hip_ctx* c, const void* all_handles, int rank, int world) {
    RCN_TRY(check_ctx(c));
    if (!all_handles) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_attach: NULL pointer");
    DevGuard g(c->device);
    RCN_TRY(p2p_attach(c, all_handles, rank, world));
    c->p2p.on = world > 1;
    return RCN_HIP_OK;
}

int rcn_hip_dp_p2p_selftest(rcn_hip_ctx* c, int iters, unsigned* mismatches, unsigned* timed_out) {
    RCN_TRY(check_ctx(c));
    if (!mismatches || !timed_out || iters < 1) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_selftest: bad arguments");
    DevGuard g(c->device);
    RCN_TRY(p2p_selftest(c, iters, mismatches, timed_out));
    if (*mismatches || *timed_out) { c->p2p.on = false; return RCN_HIP_OK; }
    // the in-kernel form of the exchange, same verdict rule (the caller's ranks see the same result and decide alike)
    unsigned bad2 = 0, to2 = 0;
    if (c->opt.dp_fused) {
        RCN_TRY(p2p_selftest_fused(c, iters, &bad2, &to2));
        c->p2p.fused = bad2 == 0 && to2 == 0;
    }
    *mismatches += bad2;
    *timed_out |= to2;
    if (c->opt.dp_fused && c->dtype == RCN_HIP_F32 && bad2 == 0 && to2 == 0) {
        unsigned bad3 = 0, to3 = 0;
        RCN_TRY(p2p_selftest_push(c, iters, &bad3, &to3));
        c->p2p.push = bad3 == 0 && to3 == 0;
        *mismatches += bad3;
        *timed_out |= to3;
    }
    return RCN_HIP_OK;
}

int rcn_hip_dp_p2p_admit(rcn_hip_ctx* c, int rank, int world, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min, void* user) {
    RCN_TRY(check_ctx(c));
    if (!allgather || !vote_min || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world)
        return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_admit: bad callbacks / rank / world (1..8 ranks)");
    RCN_TRY(need_dense(c));
    DevGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    p2p_release(c);
    c->dp_rank = rank;
    c->dp_world = world;
    P2PTransport t;
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int {
        return allgather(user, mine, all, bytes) == 0 ? RCN_HIP_OK : fail(c, RCN_HIP_ERR_HIP, "dp_p2p_admit: the caller's allgather failed");
    };
    t.vote_min = [&](int& v) -> int {
        return vote_min(user, &v) == 0 ? RCN_HIP_OK : fail(c, RCN_HIP_ERR_HIP, "dp_p2p_admit: the caller's vote failed");
    };
    return p2p_admission(c, t);
}

int rcn_hip_dp_admission_rehearse(int rank, int world, const char* faults, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min, void* user,
                                   int* form_out, int* resident_out) {
    if (!allgather || !vote_min || !form_out || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world) return RCN_HIP_ERR_INVALID_ARG;
    const std::string all_faults = faults ? faults : "";
    auto faulty = [&](const char* stage) {
        const std::string want = std::string(stage) + ":" + std::to_string(rank);
        size_t pos = 0;
        while (pos <= all_faults.size()) {
            const size_t end = all_faults.find(',', pos);
            if (all_faults.substr(pos, end == std::string::npos ? std::string::npos : end - pos) == want) return true;
            if (end == std::string::npos) break;
            pos = end + 1;
        }
        return false;
    };
    AdmissionOps ops;
    ops.do_export = [&](char* h) { std::memset(h, 0, kP2PHandleBytes); h[0] = (char)(rank + 1); return faulty("export") ? 0 : 1; };
    ops.attach = [&](const char* all) {
        for (int r = 0; r < world; ++r)
            if (all[(size_t)r * kP2PHandleBytes] != (char)(r + 1)) return 0;          // the transport delivered every rank's bytes, in rank order
        return faulty("attach") ? 0 : 1;
    };
    ops.known_answer = [&]() { return faulty("kat") ? 0 : 1; };
    ops.wants_fused = [&]() { return faulty("nofused") ? 0 : 1; };
    ops.tagged_words = [&]() { return (faulty("ll") || faulty("llskip")) ? 0 : 1; };
    ops.wants_push = [&]() { return faulty("f64") ? 0 : 1; };
    ops.pushed_words = [&]() { return (faulty("push") || faulty("pushskip")) ? 0 : 1; };
    ops.clear_sticky = [&]() { return faulty("clear") ? 0 : 1; };
    P2PTransport t;
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int { return allgather(user, mine, all, bytes) == 0 ? RCN_HIP_OK : RCN_HIP_ERR_HIP; };
    t.vote_min = [&](int& v) -> int { return vote_min(user, &v) == 0 ? RCN_HIP_OK : RCN_HIP_ERR_HIP; };
    AdmissionOutcome out;
    const int st = admission_protocol(world, ops, t, out);
    *form_out = out.on ? (out.fused ? 2 : 1) : 0;
    if (resident_out) *resident_out = (out.on && out.push) ? 1 : 0;
    return st;
}

int rcn_hip_dp_p2p_active(const rcn_hip_ctx* c) { return c && c->p2p.on ? (c->p2p.fused ? 2 : 1) : 0; }

int rcn_hip_train_epoch_gathers(rcn_hip_ctx* c, size_t B) {
    if (!c) return 0;
    DevGuard g(c->device);
    return use_xcd(c, B) && xcd_gather(c) ? 1 : 0;
}

int rcn_hip_dp_resident(rcn_hip_ctx* c, size_t B_shard) {
    if (!c) return 0;
    DevGuard g(c->device);
    return dp_on_xcd(c, B_shard) ? 1 : 0;
}

int rcn_hip_dp_epoch_steps_dev(rcn_hip_ctx* c, size_t first_batch, size_t n_batches, double eta, void* loss_dev) {
    RCN_TRY(check_ctx(c));
    if (!c->comm && !c->p2p.on) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_epoch_steps: rcn_hip_dp_init was not called");
    if (c->p2p.on && *c->p2p.err_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "dp_epoch_steps: the peer exchange timed out in an earlier call; the replicas are no longer in step");
    if (c->epoch_nb == 0) return fail(c, RCN_HIP_ERR_STATE, "dp_epoch_steps: no epoch begun (rcn_hip_epoch_begin_dev), or another training call has re-packed the image since");
    if (first_batch > c->epoch_nb || n_batches > c->epoch_nb - first_batch) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_epoch_steps: batches beyond the begun epoch");
    if (n_batches == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    const size_t B = c->epoch_B, seg = c->epoch_seg;
    DevGuard g(c->device);
    if (!(dp_on_xcd(c, B)))
        return fail(c, RCN_HIP_ERR_UNSUPPORTED, "dp_epoch_steps: only where the data-parallel step runs on the resident kernel (rcn_hip_dp_resident); "
                                                "rcn_hip_dp_train_epoch_dev packs and runs its batches itself on every form");
    RCN_TRY(ensure_dense_ws(c, B));
    RCN_TRY(ensure_pipe_ws(c, B));
    RCN_TRY(ensure_xcd_ws(c, B));
    RCN_TRY(enqueue_xcd_epoch(c, nullptr, nullptr, nullptr, B, n_batches, eta, loss_dev, false, true, first_batch, seg, true));
    HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_dp_world(const rcn_hip_ctx* c) { return c ? c->dp_world : 0; }
int rcn_hip_dp_rank(const rcn_hip_ctx* c) { return c ? c->dp_rank : -1; }

int rcn_hip_dp_broadcast_params(rcn_hip_ctx* c, int root) {
    RCN_TRY(check_ctx(c));
    if (!c->comm) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_broadcast_params: rcn_hip_dp_init was not called");
    if (root < 0 || root >= c->dp_world) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_broadcast_params: bad root");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    NCCL_TRY(c, rcn::Rccl::get().Broadcast(c->params.p, c->params.p, (size_t)c->nd.P, c->dtype == RCN_HIP_F64 ? ncclDouble : ncclFloat, root,
                                           c->comm, c->stream));
    return RCN_HIP_OK;
}

static int dp_epoch_impl(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool launch) {
    RCN_TRY(check_ctx(c));
    if (!c->comm && !c->p2p.on) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: rcn_hip_dp_init was not called");
    if (c->p2p.on && *c->p2p.err_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "dp_train_epoch: the peer all-reduce timed out waiting for rank " + std::to_string((int)*c->p2p.err_host - 1) +
                                            "'s peers in an earlier call; the replicas are no longer in step");
    if (!X || !Y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: shard batch size must be in 1..2^30");
    if (nb == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    const size_t es = c->esz(), P = (size_t)c->nd.P, F = c->nd.dims[0], Cc = c->nd.dims[c->nd.L];
    HIP_TRY(c, ws_ensure(c, c->grad, (P + 1) * es));          // [gradient | loss]: one all-reduce carries both
    rcn::Rccl& r = rcn::Rccl::get();
    const double Bg = (double)B * (double)c->dp_world;          // the global batch.len() of rcn.rs:214
    const double scale = eta / Bg, loss_scale = 1.0 / (2.0 * Bg);
    char* gbuf = (char*)c->grad.p;
    void* lslot = gbuf + P * es;
    const bool f64 = c->dtype == RCN_HIP_F64;
    if (dp_on_xcd(c, B)) {
        // the resident one-XCD kernel with the exchange between its gradient MFMAs and its update (dense_xcd.hpp, DP = true): one
        // launch per segment of the epoch image, nothing to capture
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, nb));
        RCN_TRY(ensure_xcd_ws(c, B));
        if (!launch) return RCN_HIP_OK;
        RCN_TRY(enqueue_xcd_epoch(c, X, Y, perm, B, nb, eta, loss_dev, false, false, 0, 0, true));
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
        return RCN_HIP_OK;
    }
    if (c->p2p.on && c->dense_path != 1 && p2_supported(c->nd, B)) {
        // the lean pipeline with the exchange inside its third kernel (dense_p2_dp.hpp)
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, nb));
        // captured once per (pointers, B, n_batches, eta) and replayed: three launches per step would otherwise be bound by the
        // host's launch rate (~6 us each), not by the GPU.  Sequence numbers inside the graph are offsets from a device word.
        if (!c->opt.dp_graph) {
            if (!launch) return RCN_HIP_OK;
            RCN_TRY(f64 ? enqueue_pipe_steps_dp<double>(c, X, Y, perm, B, nb, eta, loss_dev, false, c->p2p.fused)
                        : enqueue_pipe_steps_dp<float>(c, X, Y, perm, B, nb, eta, loss_dev, false, c->p2p.fused));
        } else {
            const EpochKey key{X, Y, perm, B, nb, eta, loss_dev};
            auto it = c->dp_graphs.find(key);
            if (it == c->dp_graphs.end()) {
                hipGraph_t graph = nullptr;
                HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int st = f64 ? enqueue_pipe_steps_dp<double>(c, X, Y, perm, B, nb, eta, loss_dev, true, c->p2p.fused)
                                   : enqueue_pipe_steps_dp<float>(c, X, Y, perm, B, nb, eta, loss_dev, true, c->p2p.fused);
                hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (st != RCN_HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
                HIP_TRY(c, e);
                hipGraphExec_t exec = nullptr;
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                HIP_TRY(c, e);
                if (c->dp_graphs.size() >= 16) drop_graphs(c);
                it = c->dp_graphs.emplace(key, exec).first;
            }
            if (!launch) return RCN_HIP_OK;                   // rcn_hip_dp_prepare_epoch_dev: instantiated, not run
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, c->stream, c->p2p.err_dev + 16, c->p2p.seq);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
            c->p2p.seq += (unsigned)nb;
        }
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));   // read at the next call
        return RCN_HIP_OK;
    }
    if (!launch) return RCN_HIP_OK;                           // nothing to prepare on the eager paths
    if (c->p2p.on) {
        // gradient kernels write straight into this rank's exported slot; ONE kernel then waits for the peers' flags, reads
        // all `world` slots over xGMI, adds them in rank order and applies the update (dp_p2p.hpp)
        for (size_t j = 0; j < nb; ++j) {
            const void* xb = perm ? X : (const char*)X + j * B * F * es;
            const void* yb = perm ? Y : (const char*)Y + j * B * Cc * es;
            const int32_t* ib = perm ? perm + j * B : nullptr;
            char* slot = (char*)c->p2p.local_buf + (size_t)((c->p2p.seq + 1) & 1u) * c->p2p.stride * es;
            void* lj = loss_dev ? (char*)loss_dev + j * es : nullptr;
            if (f64) {
                RCN_TRY(launch_fwd<double>(c, true, xb, yb, ib, B, nullptr));
                RCN_TRY(launch_wgrad<double>(c, false, xb, ib, B, 0.0, slot, slot + P * es, loss_scale));
                RCN_TRY(p2p_step<double>(c, 0, scale, lj, p2p_timeout_ticks(c)));
            } else {
                RCN_TRY(launch_fwd<float>(c, true, xb, yb, ib, B, nullptr));
                RCN_TRY(launch_wgrad<float>(c, false, xb, ib, B, 0.0, slot, slot + P * es, loss_scale));
                RCN_TRY(p2p_step<float>(c, 0, scale, lj, p2p_timeout_ticks(c)));
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));   // read at the next call
        return RCN_HIP_OK;
    }
    for (size_t j = 0; j < nb; ++j) {
        const void* xb = perm ? X : (const char*)X + j * B * F * es;
        const void* yb = perm ? Y : (const char*)Y + j * B * Cc * es;
        const int32_t* ib = perm ? perm + j * B : nullptr;
        if (f64) {
            RCN_TRY(launch_fwd<double>(c, true, xb, yb, ib, B, nullptr));
            RCN_TRY(launch_wgrad<double>(c, false, xb, ib, B, 0.0, gbuf, lslot, loss_scale));
        } else {
            RCN_TRY(launch_fwd<float>(c, true, xb, yb, ib, B, nullptr));
            RCN_TRY(launch_wgrad<float>(c, false, xb, ib, B, 0.0, gbuf, lslot, loss_scale));
        }
        NCCL_TRY(c, r.AllReduce(gbuf, gbuf, P + 1, f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, c->stream));
        if (f64)
            hipLaunchKernelGGL((k_apply_gradient<double>), dim3(grid_for((int)P, 256)), dim3(256), 0, c->stream, (double*)c->params.p,
                               (const double*)gbuf, scale, (int)P);
        else
            hipLaunchKernelGGL((k_apply_gradient<float>), dim3(grid_for((int)P, 256)), dim3(256), 0, c->stream, (float*)c->params.p,
                               (const float*)gbuf, (float)scale, (int)P);
        HIP_TRY(c, hipGetLastError());
        if (loss_dev) HIP_TRY(c, hipMemcpyAsync((char*)loss_dev + j * es, lslot, es, hipMemcpyDeviceToDevice, c->stream));
    }
    return RCN_HIP_OK;
}

int rcn_hip_dp_train_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta,
                               void* loss_dev) {
    return dp_epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, true);
}

int rcn_hip_dp_prepare_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta,
                                 void* loss_dev) {
    return dp_epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, false);
}

// ---------------------------------------------------------------- RCN::train's data flow with the sets resident in HBM
int rcn_hip_load_data(rcn_hip_ctx* c, int slot, const uint8_t* imgs, const int32_t* labels, size_t n, double* mean, double* sd) {
    RCN_TRY(check_ctx(c));
    if (slot < 0 || slot > 1) return fail(c, RCN_HIP_ERR_INVALID_ARG, "load_data: slot must be 0 (training set) or 1 (testing set)");
    if (!imgs || !labels || n == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "load_data: empty set (gen_scales indexes iv[0], rcn.rs:233)");
    RCN_TRY(need_dense(c));
    const int Cc = c->nd.dims[c->nd.L];
    for (size_t i = 0; i < n; ++i)
        if (labels[i] < 0 || labels[i] >= Cc) return fail(c, RCN_HIP_ERR_SHAPE, "load_data: a label is not below `classes` (the one-hot vector of rcn.rs:466-471 would not match the output layer)");
    DevGuard g(c->device);
    auto& rs = c->sets[slot];
    const size_t img_b = n * (size_t)c->fd.H * c->fd.W, F = (size_t)c->fd.F, es = c->esz();
    drop_graphs(c);                                   // epoch graphs hold the old set's pointers
    rs.n = 0;
    HIP_TRY(c, rs.imgs.ensure(img_b));
    HIP_TRY(c, rs.X.ensure(n * F * es));
    HIP_TRY(c, rs.Y.ensure(n * (size_t)Cc * es));
    HIP_TRY(c, rs.perm.ensure(n * sizeof(int32_t)));
    HIP_TRY(c, hipMemcpyAsync(rs.imgs.p, imgs, img_b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rs.perm.p, labels, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));     // staged in the index buffer
    if (c->dtype == RCN_HIP_F64) hipLaunchKernelGGL((k_one_hot<double>), dim3(grid_for(n * Cc, 256)), dim3(256), 0, c->stream, (const int32_t*)rs.perm.p, n, Cc, (double*)rs.Y.p);
    else hipLaunchKernelGGL((k_one_hot<float>), dim3(grid_for(n * Cc, 256)), dim3(256), 0, c->stream, (const int32_t*)rs.perm.p, n, Cc, (float*)rs.Y.p);
    HIP_TRY(c, hipGetLastError());
    RCN_TRY(rcn_hip_features_dev(c, (const uint8_t*)rs.imgs.p, n, rs.X.p, 0));           // rcn.rs:399-401
    RCN_TRY(gen_scales_impl(c, rs.X.p, n * F, mean, sd));                                // rcn.rs:406 (overwrites scale_set; blocks)
    RCN_TRY(rcn_hip_standardize_dev(c, rs.X.p, n * F));                                  // rcn.rs:407-412
    rs.n = n;
    return RCN_HIP_OK;
}

int rcn_hip_train_set_epoch(rcn_hip_ctx* c, int slot, const int32_t* perm, uint64_t shuffle_seed, size_t B, double eta, double* loss_out) {
    RCN_TRY(check_ctx(c));
    if (slot < 0 || slot > 1 || c->sets[slot].n == 0) return fail(c, RCN_HIP_ERR_STATE, "train_set_epoch: rcn_hip_load_data has not filled this slot");
    if (B == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_set_epoch: batch size 0 (chunks_exact panics, rcn.rs:147)");
    auto& rs = c->sets[slot];
    const size_t n = rs.n, nb = n / B;                 // chunks_exact drops the tail (rcn.rs:147)
    if (perm)
        for (size_t i = 0; i < nb * B; ++i)
            if (perm[i] < 0 || (size_t)perm[i] >= n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "train_set_epoch: index out of range");
    DevGuard g(c->device);
    if (perm) {
        rcn_hip_ctx::PermSource ps;
        ps.kind = 2; ps.buf = (int32_t*)rs.perm.p; ps.n = nb * B; ps.passes = 1;
        ps.host.assign(perm, perm + nb * B);
        HIP_TRY(c, hipMemcpyAsync(rs.perm.p, perm, nb * B * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        note_perm_source(c, std::move(ps));
    }
    else RCN_TRY(rcn_hip_shuffle_dev(c, (int32_t*)rs.perm.p, n, 1, shuffle_seed ? shuffle_seed : ((uint64_t)std::random_device{}() << 32) ^ std::random_device{}()));   // rcn.rs:146
    if (nb == 0) return RCN_HIP_OK;
    void* loss_dev = nullptr;
    if (loss_out) { HIP_TRY(c, rs.loss.ensure(nb * c->esz())); loss_dev = rs.loss.p; }
    RCN_TRY(rcn_hip_train_epoch_dev(c, rs.X.p, rs.Y.p, (const int32_t*)rs.perm.p, B, nb, eta, loss_dev));
    if (loss_out) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        RCN_TRY(sticky_errors(c));                     // (a step-down from the resident kernel re-runs what was lost, costs included, before they are read)
        return download(c, loss_dev, loss_out, nb);
    }
    return RCN_HIP_OK;
}

int rcn_hip_evaluate_set(rcn_hip_ctx* c, int slot, int64_t* accepted) {
    RCN_TRY(check_ctx(c));
    if (!accepted) return fail(c, RCN_HIP_ERR_INVALID_ARG, "evaluate_set: NULL pointer");
    if (slot < 0 || slot > 1 || c->sets[slot].n == 0) return fail(c, RCN_HIP_ERR_STATE, "evaluate_set: rcn_hip_load_data has not filled this slot");
    RCN_TRY(need_params(c));
    return rcn_hip_evaluate_dev(c, c->sets[slot].X.p, c->sets[slot].Y.p, c->sets[slot].n, accepted);
}

int rcn_hip_set_size(const rcn_hip_ctx* c, int slot, int64_t* n) {
    if (!c || !n || slot < 0 || slot > 1) return RCN_HIP_ERR_INVALID_ARG;
    *n = (int64_t)c->sets[slot].n;
    return RCN_HIP_OK;
}

int rcn_hip_forward_dev(rcn_hip_ctx* c, const void* x, size_t n, void* out) {
    RCN_TRY(check_ctx(c));
    if ((!x || !out) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "forward: NULL pointer");
    if (n == 0) return RCN_HIP_OK;
    if (n > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "forward: too many samples in one call");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    if (dense_is_wide(c->nd, c->esz())) RCN_TRY(ensure_dense_ws(c, n));       // the layer-by-layer path keeps hidden activations in global memory
    if (c->dtype == RCN_HIP_F64) return launch_fwd<double>(c, false, x, nullptr, nullptr, n, out);
    return launch_fwd<float>(c, false, x, nullptr, nullptr, n, out);
}

int rcn_hip_forward(rcn_hip_ctx* c, const double* x, size_t n, double* out) {
    RCN_TRY(check_ctx(c));
    if ((!x || !out) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "forward: NULL pointer");
    if (n == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    const size_t Cc = c->nd.dims[c->nd.L];
    RCN_TRY(upload(c, c->xstage, x, n * (size_t)c->nd.dims[0]));
    HIP_TRY(c, c->ostage.ensure(n * Cc * c->esz()));
    RCN_TRY(rcn_hip_forward_dev(c, c->xstage.p, n, c->ostage.p));
    return download(c, c->ostage.p, out, n * Cc);
}

static int argmax_dev(rcn_hip_ctx* c, const void* outv, size_t n, int32_t* host_cls) {
    const int Cc = c->nd.dims[c->nd.L];
    HIP_TRY(c, c->misc.ensure(n * sizeof(int32_t) + 64));
    if (c->dtype == RCN_HIP_F64)
        hipLaunchKernelGGL((k_argmax_last<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)outv, (int)n, Cc, (int*)c->misc.p);
    else
        hipLaunchKernelGGL((k_argmax_last<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)outv, (int)n, Cc, (int*)c->misc.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(host_cls, c->misc.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_classify(rcn_hip_ctx* c, const double* x, size_t n, int32_t* cls) {
    RCN_TRY(check_ctx(c));
    if ((!x || !cls) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "classify: NULL pointer");
    if (n == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(upload(c, c->xstage, x, n * (size_t)c->nd.dims[0]));
    HIP_TRY(c, c->ostage.ensure(n * (size_t)c->nd.dims[c->nd.L] * c->esz()));
    RCN_TRY(rcn_hip_forward_dev(c, c->xstage.p, n, c->ostage.p));
    return argmax_dev(c, c->ostage.p, n, cls);
}

int rcn_hip_evaluate_dev(rcn_hip_ctx* c, const void* x, const void* y, size_t n, int64_t* accepted) {
    RCN_TRY(check_ctx(c));
    if (!accepted || ((!x || !y) && n)) return fail(c, RCN_HIP_ERR_INVALID_ARG, "evaluate: NULL pointer");
    *accepted = 0;
    if (n == 0) return RCN_HIP_OK;
    DevGuard g(c->device);
    const int Cc = c->nd.dims[c->nd.L];
    // the accuracy of rcn.rs:150-164 is read from the parameters the epoch left: if resident launches are still unverified, drain the
    // stream first -- a failure is healed (or reported) before the forward pass runs, not after
    if (c->xerr_host && (!c->redo.empty() || c->xerr_host[0] != 0 || c->xcd_dp_used)) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        RCN_TRY(sticky_errors(c));
    }
    HIP_TRY(c, c->ostage.ensure(n * (size_t)Cc * c->esz()));
    RCN_TRY(rcn_hip_forward_dev(c, x, n, c->ostage.p));
    HIP_TRY(c, c->misc.ensure(64));
    HIP_TRY(c, hipMemsetAsync(c->misc.p, 0, 16, c->stream));
    if (c->dtype == RCN_HIP_F64)
        hipLaunchKernelGGL((k_eval_accept<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->ostage.p, (const double*)y, (int)n, Cc,
                           (unsigned long long*)c->misc.p);
    else
        hipLaunchKernelGGL((k_eval_accept<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->ostage.p, (const float*)y, (int)n, Cc,
                           (unsigned long long*)c->misc.p);
    HIP_TRY(c, hipGetLastError());
    unsigned long long cnt = 0;
    HIP_TRY(c, hipMemcpyAsync(&cnt, c->misc.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *accepted = (int64_t)cnt;
    return RCN_HIP_OK;
}

int rcn_hip_evaluate(rcn_hip_ctx* c, const double* x, const double* y, size_t n, int64_t* accepted) {
    RCN_TRY(check_ctx(c));
    if (!accepted || ((!x || !y) && n)) return fail(c, RCN_HIP_ERR_INVALID_ARG, "evaluate: NULL pointer");
    *accepted = 0;
    if (n == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(upload(c, c->xstage, x, n * (size_t)c->nd.dims[0]));
    RCN_TRY(upload(c, c->ystage, y, n * (size_t)c->nd.dims[c->nd.L]));
    return rcn_hip_evaluate_dev(c, c->xstage.p, c->ystage.p, n, accepted);
}

int rcn_hip_time_kernels_dev(rcn_hip_ctx* c, const void* x, const void* y, size_t B, int reps, double* us_a, double* us_b, double* us_pair) {
    RCN_TRY(check_ctx(c));
    if (!x || !y || !us_a || !us_b || reps < 1 || B == 0) return fail(c, RCN_HIP_ERR_INVALID_ARG, "time_kernels: bad argument");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    HIP_TRY(c, c->grad.ensure((size_t)c->nd.P * c->esz()));
    const bool f64 = c->dtype == RCN_HIP_F64, pipe = use_pipe(c, B);
    if (pipe) RCN_TRY(ensure_pipe_ws(c, B));
    if (pipe && use_xcd(c, B)) {
        // the resident kernel: ONE launch runs every step of the image's first segment; timed as a whole (zero step: no drift), reported
        // per step in *us_second and *us_pair (there is no first / second kernel)
        RCN_TRY(ensure_xcd_ws(c, B));
        // the form the last training call ran in: rows gathered by the kernel itself (over that call's rows), or the packed image
        // (only over the matrix the caller hands in now: the remembered pointers are not trusted to be alive otherwise)
        const bool tg = xcd_gather(c) && c->xg.B == B && c->xg.nb >= 2 && c->xg.X == (const float*)x && c->xg.Y == (const float*)y;
        size_t n = tg ? (c->xg.nb < 64 ? c->xg.nb : 64) : ((c->packed_B == B && c->packed_nb >= 2) ? c->packed_nb : 0);
        auto timed_launch = [&]() {
            return tg ? enqueue_xcd_steps(c, c->xg.X, c->xg.Y, B, n, 0.0, nullptr, false, c->xg.perm, true)
                      : enqueue_xcd_steps(c, (const float*)c->xpack.p, (const float*)c->ypack.p, B, n, 0.0, nullptr);
        };
        if (n == 0) {
            RCN_TRY(ensure_pack_ws(c, B, 1));
            RCN_TRY(launch_pack<float>(c, x, y, nullptr, B, 0, 1, 0, 1));
            n = 1;
        }
        const size_t saved_nb = c->epoch_nb;      // timing on the image does not end a begun epoch (nothing is re-packed unless n was 0)
        RCN_TRY(timed_launch());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipEvent_t e0, e1;
        HIP_TRY(c, hipEventCreate(&e0));
        HIP_TRY(c, hipEventCreate(&e1));
        const int launches = (int)((reps + n - 1) / n) < 4 ? 4 : (int)((reps + n - 1) / n);
        float best = 0.f, total = 0.f;
        int st = RCN_HIP_OK;
        for (int i = 0; i < launches && st == RCN_HIP_OK; ++i) {
            hipError_t e = hipEventRecord(e0, c->stream);
            if (e == hipSuccess) st = timed_launch();
            if (e == hipSuccess && st == RCN_HIP_OK) e = hipEventRecord(e1, c->stream);
            if (e == hipSuccess && st == RCN_HIP_OK) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess && st == RCN_HIP_OK) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess && st == RCN_HIP_OK) st = fail(c, RCN_HIP_ERR_HIP, hipGetErrorString(e));
            total += ms;
            (void)best;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (n > 1 && !tg) c->epoch_nb = saved_nb;
        *us_a = 0.0;
        *us_b = (double)total * 1000.0 / ((double)launches * (double)n);
        if (us_pair) *us_pair = *us_b;
        return st;
    }
    // which == 0: first kernel of a step (k_dense_fwd | k_pipe_b), which == 1: second (k_dense_wgrad | k_pipe_a).
    // Updates run with scale 0 / gradient-out so the parameters do not drift while timing.
    // Feature-sliced path: if the context still holds the packed image of a whole epoch at this batch size (the normal
    // case right after rcn_hip_train_epoch_dev), launch i works on batches i, i+1 of that image, so the timing sees the
    // same cold slice reads as the real epoch loop; otherwise the given batch is packed and reused.
    size_t rot = (pipe && c->packed_B == B && c->packed_nb >= 2) ? c->packed_nb : 0;
    const size_t xstride = (size_t)pipe_slices(c->nd) * B * 16 * c->esz(), ystride = B * (size_t)c->nd.dims[c->nd.L] * c->esz();
    size_t it = 0;
    auto launch = [&](int which) -> int {
        if (pipe) {
            const size_t j = rot ? (it++ % (rot - 1)) : 0;
            const void* xp = (const char*)c->xpack.p + j * xstride;
            const void* xn = rot ? (const void*)((const char*)xp + xstride) : xp;
            if (which == 0) return f64 ? launch_pipe_b<double>(c, (const char*)c->ypack.p + j * ystride, B) : launch_pipe_b<float>(c, (const char*)c->ypack.p + j * ystride, B);
            return f64 ? launch_pipe_a<double>(c, xp, xn, B, 0.0, nullptr, 1.0, true, true) : launch_pipe_a<float>(c, xp, xn, B, 0.0, nullptr, 1.0, true, true);
        }
        if (which == 0) return f64 ? launch_fwd<double>(c, true, x, y, nullptr, B, nullptr) : launch_fwd<float>(c, true, x, y, nullptr, B, nullptr);
        return f64 ? launch_wgrad<double>(c, false, x, nullptr, B, 0.0, c->grad.p, nullptr, 1.0) : launch_wgrad<float>(c, false, x, nullptr, B, 0.0, c->grad.p, nullptr, 1.0);
    };
    // one complete step's worth of intermediates + warm code / LDS attributes, outside capture
    if (pipe) {
        if (!rot) {
            RCN_TRY(ensure_pack_ws(c, B, 1));
            RCN_TRY(f64 ? launch_pack<double>(c, x, y, nullptr, B, 0, 1, 0, 1) : launch_pack<float>(c, x, y, nullptr, B, 0, 1, 0, 1));
        }
        RCN_TRY(f64 ? launch_pipe_a<double>(c, c->xpack.p, c->xpack.p, B, 0.0, nullptr, 1.0, false, true)
                    : launch_pipe_a<float>(c, c->xpack.p, c->xpack.p, B, 0.0, nullptr, 1.0, false, true));
    }
    RCN_TRY(launch(0));
    RCN_TRY(launch(1));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    double res[3] = {0, 0, 0};
    int st = RCN_HIP_OK;
    // which == 2: the two kernels alternating, as in the real loop (reps pairs)
    for (int which = 0; which < (us_pair ? 3 : 2) && st == RCN_HIP_OK; ++which) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        it = 0;
        for (int i = 0; i < reps && e == hipSuccess && st == RCN_HIP_OK; ++i) {
            if (which < 2) st = launch(which);
            else { const size_t keep = it; st = launch(0); it = keep; if (st == RCN_HIP_OK) st = launch(1); }
        }
        hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
        if (e == hipSuccess) e = e2;
        if (e == hipSuccess && st == RCN_HIP_OK) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e == hipSuccess && st == RCN_HIP_OK) {
            e = hipGraphLaunch(exec, c->stream);                       // untimed warm replay
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
            if (e == hipSuccess) e = hipGraphLaunch(exec, c->stream);
            if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            res[which] = (double)ms * 1000.0 / reps;
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess && st == RCN_HIP_OK) st = fail(c, RCN_HIP_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *us_a = res[0]; *us_b = res[1];
    if (us_pair) *us_pair = res[2];
    return st;
}

#ifdef RCN_STAMPS
int rcn_hip_debug_read_stamps(rcn_hip_ctx* c, unsigned long long* out) {
    DevGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rcn_stamps), sizeof(unsigned long long) * 2 * 512 * 16));
    return RCN_HIP_OK;
}
#endif

constexpr size_t kPinImgBytes = 64 * 1024, kPinClsBytes = 4096;    // serving path: up to 64 KB of pixels / 1024 classes per call

int rcn_hip_classify_images(rcn_hip_ctx* c, const uint8_t* imgs, size_t n, int32_t* cls) {
    RCN_TRY(check_ctx(c));
    if ((!imgs || !cls) && n) return fail(c, RCN_HIP_ERR_INVALID_ARG, "classify_images: NULL pointer");
    if (n == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    const size_t img_b = n * (size_t)c->fd.H * c->fd.W;
    const int Cc = c->nd.dims[c->nd.L];
    HIP_TRY(c, c->xstage.ensure(n * (size_t)c->fd.F * c->esz()));
    HIP_TRY(c, c->ostage.ensure(n * (size_t)Cc * c->esz()));
    if (img_b <= kPinImgBytes && n * sizeof(int32_t) <= kPinClsBytes) {
        // Latency path (one request of the reference's backend, backend/src/main.rs:22-42): pixels are copied by the CPU into
        // a pinned block the GPU reads in place, the class index is written straight back into it; three small launches and
        // ONE synchronisation instead of two staged copies around them.
        if (!c->pin_host) {
            HIP_TRY(c, hipHostMalloc(&c->pin_host, kPinImgBytes + kPinClsBytes, hipHostMallocMapped));
            HIP_TRY(c, hipHostGetDevicePointer(&c->pin_dev, c->pin_host, 0));
        }
        std::memcpy(c->pin_host, imgs, img_b);
        int* cls_dev = (int*)((char*)c->pin_dev + kPinImgBytes);
        if (feat_is_cpcp28(c) && c->dense_err.empty() && serve_supported(c->nd)) {
            // ONE launch per request (serve.hpp).  A single image waits on the result word itself: the kernel's last act is
            // a system-scope store of the class into this host-mapped block, which the host sees a few microseconds before
            // the stream's completion signal would wake it.  Bounded: after 2 ms fall back to the stream synchronise.
            volatile int32_t* res = (volatile int32_t*)((char*)c->pin_host + kPinImgBytes);
            if (n == 1) res[0] = -1;
            if (c->dtype == RCN_HIP_F64)
                hipLaunchKernelGGL((k_serve<double>), dim3((unsigned)n), dim3(kServeThreads), 0, c->stream, c->nd, (const double*)c->params.p,
                                   (const uint8_t*)c->pin_dev, c->mean, c->sd, cls_dev, (double*)nullptr);
            else
                hipLaunchKernelGGL((k_serve<float>), dim3((unsigned)n), dim3(kServeThreads), 0, c->stream, c->nd, (const float*)c->params.p,
                                   (const uint8_t*)c->pin_dev, (float)c->mean, (float)c->sd, cls_dev, (float*)nullptr);
            HIP_TRY(c, hipGetLastError());
            bool got = false;
            if (n == 1) {
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spin = 0;; ++spin) {
                    if (res[0] >= 0) { got = true; break; }
                    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                }
            }
            if (!got) HIP_TRY(c, hipStreamSynchronize(c->stream));
            std::memcpy(cls, (char*)c->pin_host + kPinImgBytes, n * sizeof(int32_t));       // rcn.rs:92-97
            return RCN_HIP_OK;
        }
        RCN_TRY(rcn_hip_features_dev(c, (const uint8_t*)c->pin_dev, n, c->xstage.p, 1));     // rcn.rs:84-89
        RCN_TRY(rcn_hip_forward_dev(c, c->xstage.p, n, c->ostage.p));                        // rcn.rs:91
        if (c->dtype == RCN_HIP_F64)
            hipLaunchKernelGGL((k_argmax_last<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->ostage.p, (int)n, Cc, cls_dev);
        else
            hipLaunchKernelGGL((k_argmax_last<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)c->ostage.p, (int)n, Cc, cls_dev);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::memcpy(cls, (char*)c->pin_host + kPinImgBytes, n * sizeof(int32_t));           // rcn.rs:92-97
        return RCN_HIP_OK;
    }
    HIP_TRY(c, c->scratch0.ensure(img_b));
    HIP_TRY(c, hipMemcpyAsync(c->scratch0.p, imgs, img_b, hipMemcpyHostToDevice, c->stream));
    RCN_TRY(rcn_hip_features_dev(c, (const uint8_t*)c->scratch0.p, n, c->xstage.p, 1));   // rcn.rs:84-89
    RCN_TRY(rcn_hip_forward_dev(c, c->xstage.p, n, c->ostage.p));                          // rcn.rs:91
    return argmax_dev(c, c->ostage.p, n, cls);                                             // rcn.rs:92-97
}

}  // extern "C"
