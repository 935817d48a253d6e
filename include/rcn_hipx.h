/*
 * rcn_hipx.h -- C ABI of the north-star EXTENSION ("Track X"): a trainable convolution network on gfx950.
 *
 * BASELINE.json's north_star asks for trainable Conv2d forward/backward lowered to im2col + MFMA GEMM, dense layers,
 * softmax / cross-entropy and SGD.  The reference crate has none of these (its "convolution" layers are four fixed
 * Sobel filters with no backward pass, rcn/src/utils/kernel.rs:38-53, rcn/src/rcn.rs:317-356; its dense part is
 * sigmoid / MSE, rcn.rs:260-314), so nothing here replaces a reference function and parity is defined against this
 * repository's own f64 oracle (oracle/convnet_oracle.py) and finite differences -- "parity unpinned" by construction.
 * The reference-parity hot path lives in rcn_hip.h.
 *
 * Data: activations NHWC fp32; a batch is [B][H][W][C] contiguous.  Logical parameters per layer: W[K][Cout] row-major
 * with K = (kh*3 + kw)*Cin + ci for a 3x3 convolution (pad 1, stride 1) and K = input features for a dense layer
 * (features of a conv stack are flattened in (h, w, c) order), followed by b[Cout].
 */
#ifndef RCN_HIPX_H
#define RCN_HIPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rcn_hipx_net rcn_hipx_net;

typedef enum {
    RCN_HIPX_CONV3X3_RELU = 0,   /* 3x3, stride 1, pad 1, + bias, ReLU;  out = output channels (multiple of 32)            */
    RCN_HIPX_MAXPOOL2 = 1,       /* 2x2 / stride 2 max-pool (even H, W); out ignored                                         */
    RCN_HIPX_DENSE_RELU = 2,     /* dense + bias + ReLU; out = units (multiple of 32)                                       */
    RCN_HIPX_DENSE = 3           /* final dense + bias (logits); out = classes (any; padded to 32 internally)               */
} rcn_hipx_layer_kind;

typedef struct rcn_hipx_layer { int32_t kind; int32_t out; } rcn_hipx_layer;

/* status: 0 ok, -1 invalid argument, -2 shape, -3 unsupported, -4 HIP error, -5 no device, -6 state, -7 out of memory */
int  rcn_hipx_create(int device, int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int max_batch,
                     void* hip_stream /* NULL: own stream */, rcn_hipx_net** out);
void rcn_hipx_destroy(rcn_hipx_net* net);
const char* rcn_hipx_last_error(const rcn_hipx_net* net);
int  rcn_hipx_synchronize(rcn_hipx_net* net);
int  rcn_hipx_param_count(const rcn_hipx_net* net, int64_t* logical, int64_t* padded);
int  rcn_hipx_classes(const rcn_hipx_net* net);
/* GEMM operand precision of the forward, input-gradient and weight-gradient convolutions / dense layers.  Two places compute in
 * fp32 in EITHER mode, by shape alone: a first layer whose whole 3x3xCin patch is one k-block (9*Cin <= 32: nothing of the MFMA
 * rate to gain), and the classifier head when it is the fused one (logits layer of <= 32 classes on a ReLU dense layer of <= 256
 * units).  RCN_HIPX_FP32 (default): fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), exact fp32 products.  RCN_HIPX_BF16: operands rounded to bf16 on their way into LDS,
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; activations, gradients, parameters and the SGD update stay fp32 in HBM.
 * Results then agree with an f64 evaluation to ~1e-2 relative instead of ~1e-4.
 * RCN_HIPX_BF16_STORED: RCN_HIPX_BF16, and the convolutional stage's tensors -- every convolution's and pool's output map and its
 * gradient -- are KEPT in HBM as bf16 (parameters, dense-layer tensors, the input batch, the SGD update stay fp32).  What the
 * arithmetic rounds is the same as in RCN_HIPX_BF16 -- every consumer of those tensors rounds them to bf16 on the way into LDS anyway
 * -- with two exceptions: the first layer's weight gradient (fp32 kernel) and every layer's bias gradient (summed in fp32 from dZ)
 * now see dZ already rounded.  The stage's HBM traffic halves.  Covers nets whose first layer has 1 or 3 input channels, whose other
 * convolutions run on the LDS-tiled bf16 kernels (32 or a multiple of 64 input channels, options "halo" and "bf16_pipe" on) and
 * whose pools are fused into the convolution in front of them (even maps); rcn_hipx_set_precision walks the net's plan first and
 * returns -3 -- nothing changed -- with the reason in rcn_hipx_last_error if a layer is not covered. */
enum { RCN_HIPX_FP32 = 0, RCN_HIPX_BF16 = 1, RCN_HIPX_BF16_STORED = 2 };
int  rcn_hipx_set_precision(rcn_hipx_net* net, int mode);
/* logical layout, host memory, all layers back to back: W_0[K][Cout], b_0[Cout], W_1 ... */
/* Which fp32 3x3 convolution kernels run (bf16 mode has its own rule).  GEMM: the implicit-GEMM kernels only.  AUTO (default): the
 * LDS-tiled kernels where at least 70 % of a 128-pixel block's rows are real pixels and the layer is not a split-K case, the first
 * layer's own kernels for 1 / 3 input channels.  LDS: the LDS-tiled kernels wherever their shape constraints hold (tests).
 * The environment variable RCN_HIPX_HALO_F32 (0 / 1 / 2) only seeds a new net's mode. */
enum { RCN_HIPX_TILING_GEMM = 0, RCN_HIPX_TILING_AUTO = 1, RCN_HIPX_TILING_LDS = 2 };
int  rcn_hipx_set_tiling(rcn_hipx_net* net, int mode);
/* Backward pass: run the weight gradients on a second stream beside the input-gradient chain: 0 = no (default), 1 = every layer's,
 * 2 = the dense layers' only.  Measured on MI355X (CIFAR shape, fp32): no gain (0.420 / 0.418 / 0.418 ms) -- the kernels fill the chip
 * on their own; with a reduction launch per layer on the second stream (before k_reduce_all) mode 1 was 7 % slower.
 * RCN_HIPX_OVERLAP seeds a new net's mode.  Same kernels, same sums, bit-identical results in every mode. */
int  rcn_hipx_set_overlap(rcn_hipx_net* net, int on);
/* Kernel-selection knobs of ONE net (round 4: they used to be process-wide statics read from the environment at first use).  The
 * environment variable named beside an option only seeds its default when a net -- or a plan -- is created; two nets of one process can
 * differ, and changing an option drops the net's captured graphs.  -1 for an unknown name or a value out of range.
 *   "halo"            RCN_HIPX_HALO            0 | 1   bf16 mode: the LDS-tiled 3x3 kernels (1)
 *   "bf16_pipe"       RCN_HIPX_BF16_PIPE       0 | 1   bf16 mode: their software-pipelined form (1)
 *   "bf16_1cb"        RCN_HIPX_BF16_1CB        0 | 1   bf16 mode: the resident-weights form for 32-channel layers (1)
 *   "bf16_rows16"     RCN_HIPX_BF16_ROWS16     0 | 1   bf16 storage: 16 x 16 pixel blocks (two row groups per wave) where the map's height allows
 *   "halo_wgrad"      RCN_HIPX_HALO_WGRAD      0 | 1   the LDS-tiled weight-gradient kernels (1)
 *   "fuse_pool_bwd"   RCN_HIPX_FUSE_POOL_BWD   0 | 1   the gradient kernels unpool while staging (1)
 *   "head"            RCN_HIPX_HEAD            0 | 1   the classifier head as one launch (1)
 *   "xcd_remap"       RCN_HIPX_XCD_REMAP       0 | 1   implicit-GEMM weight gradient: XCD-aware block order (0)
 *   "pix_per_chunk" "wg_target" "wgh_f32_target" "wgh_target" "wgb_policy" "wgf_policy" "halo_f32_slots"
 *                                                      weight-gradient chunking and the resident-grid size (csrc/rcn_hipx_api.hip: XOptions) */
int  rcn_hipx_set_option(rcn_hipx_net* net, const char* name, int value);
int  rcn_hipx_get_option(const rcn_hipx_net* net, const char* name, int* value);
int  rcn_hipx_set_params(rcn_hipx_net* net, const float* flat);
int  rcn_hipx_get_params(rcn_hipx_net* net, float* flat);
int  rcn_hipx_init_params(rcn_hipx_net* net, uint64_t seed);            /* He-normal weights, zero biases */
/* logits_dev: [B][classes] */
int  rcn_hipx_forward_dev(rcn_hipx_net* net, const float* x_dev, int B, float* logits_dev);
/* one SGD step on mean cross-entropy: forward, backward, W <- W - lr * dW.  loss_dev (nullable): mean loss before the step.
 * Replayed as one hipGraph per (pointers, B, lr). */
int  rcn_hipx_train_step_dev(rcn_hipx_net* net, const float* x_dev, const int32_t* labels_dev, int B, float lr, float* loss_dev);
/* data-parallel halves: gradients of the MEAN loss over this shard into the padded flat layout (rcn_hipx_param_count's
 * `padded`), and p <- p - scale * g from such a buffer. */
int  rcn_hipx_gradients_dev(rcn_hipx_net* net, const float* x_dev, const int32_t* labels_dev, int B, float* grad_dev, float* loss_dev);
int  rcn_hipx_apply_dev(rcn_hipx_net* net, const float* grad_dev, float scale);
/* The same gradients in BUCKETS, so that a data-parallel step can all-reduce one bucket of layers while the backward pass of the layers
 * below it still runs (6.7 MB of gradient for BASELINE configs[3]).  The layers with parameters, in the order the
 * backward pass finishes them (last to first), are cut into buckets of at least min_bucket_bytes of gradient; the padded flat layout
 * is in layer order, so a bucket is ONE contiguous slice of grad_dev.
 *   _begin_dev:  zeroes grad_dev, runs forward + loss (+ the fused classifier head); *n_buckets = how many buckets follow.
 *   _bucket_dev: k = 0 .. n_buckets - 1 in order: the backward pass through bucket k's layers and ONE reduction launch of their slabs;
 *                grad_dev[*off, *off + *len) (floats) is final on the net's stream when the call's work is -- the caller records an
 *                event there and starts the slice's all-reduce on another stream.
 * Results are bit-identical to rcn_hipx_gradients_dev (same kernels, same sums; only the reduction launch is split). */
int  rcn_hipx_gradients_begin_dev(rcn_hipx_net* net, const float* x_dev, const int32_t* labels_dev, int B, float* grad_dev, float* loss_dev,
                                  int64_t min_bucket_bytes, int* n_buckets);
int  rcn_hipx_gradients_bucket_dev(rcn_hipx_net* net, int k, int64_t* off, int64_t* len);
/* rcn_hipx_plan's walk for that bucketed step: per bucket its launches and the slice that becomes final (no GPU needed). */
int  rcn_hipx_plan_buckets(int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int batch, int precision, int tiling,
                           int64_t min_bucket_bytes, char* out, int cap);
/* logical-layout copy of a padded gradient buffer (tests) */
int  rcn_hipx_unpad_host(rcn_hipx_net* net, const float* padded_dev, float* logical_host);
/* Which kernels a training step of this net WOULD launch, one line per launch, written to `out` (NUL-terminated, truncated at `cap`).
 * Pure host code -- no GPU is needed or touched: the dispatch code of the step runs with its launches replaced by notes, so the text is
 * the library's own decision, not a restatement of it (tests/test_convnet_plan.py holds the BASELINE configurations' plans).
 * `precision` takes all three modes; for RCN_HIPX_BF16_STORED the call returns -3 with the reason in `out` when a layer of the net is not
 * covered by the kernels that take bf16 tensors. */
int  rcn_hipx_plan(int in_h, int in_w, int in_c, const rcn_hipx_layer* layers, int n_layers, int batch, int precision, int tiling, char* out, int cap);
/* The same walk for an EXISTING net at batch `batch` (<= max_batch) with that net's own precision, tiling and options: the plan and the
 * step that follows agree by construction (rcn_hipx_plan describes a net created now, seeded from the environment). */
int  rcn_hipx_plan_net(const rcn_hipx_net* net, int batch, char* out, int cap);
/* algorithmic FLOPs of one training step at batch B (2 * MACs; forward + dgrad + wgrad) */
int  rcn_hipx_step_flops(const rcn_hipx_net* net, int B, double* flops);

#ifdef __cplusplus
}
#endif
#endif
