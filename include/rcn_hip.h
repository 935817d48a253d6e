/*
 * rcn_hip.h -- C ABI of the MI355X-native (gfx950) implementation of the `rcn` crate's hot path:
 * Sobel-separable conv + ReLU + 2x2 max-pool feature extraction, standardise+clamp, the dense
 * sigmoid/MSE forward + backward pass and the minibatch SGD update.
 *
 * The reference (jtstrader/mercer-research, crate `rcn`) is a pure-Rust CPU program with NO
 * FFI / plugin boundary; this header cuts one at the private seams of `RCN`
 * (flatten_feature_set / gen_scales / train_batch / classify_test) and at its public operator
 * traits (Convolve2D / Pool2D).  Each entry point names the reference code it replaces as
 * file:line under rcn/src/.  The Rust-side binding a maintainer would add is in INTEGRATION.md
 * (and rust/rcn-hip-sys/src/lib.rs).
 *
 * Conventions
 *   - plain C: pointers + sizes only, no C++ / torch types, no exceptions or unwinding across
 *     the boundary.  Every function returns an rcn_hip_status (0 = ok, <0 = error) unless noted;
 *     rcn_hip_last_error(ctx) gives the text.  Where the reference would panic!() the call
 *     returns RCN_HIP_ERR_SHAPE / RCN_HIP_ERR_UNSUPPORTED instead.
 *   - "host" pointers are caller-owned host memory, copied in/out synchronously (blocking).
 *     "_dev" entry points take device pointers (same HIP device as the context), are enqueued on
 *     the context's stream and do not synchronise.  The library never frees caller memory.
 *   - host matrices are f64 column-major exactly as nalgebra::DMatrix stores them and as the
 *     reference's bincode checkpoint holds them (utils/serialization.rs:16-24,98): element (r,c)
 *     of an R x C matrix lives at [c*R + r].  Sample batches are "sample-major": sample i's
 *     vector is contiguous at [i*len .. (i+1)*len).
 *   - device buffers handed to _dev calls hold the context's arithmetic type (rcn_hip_dtype):
 *     float for RCN_HIP_F32, double for RCN_HIP_F64, same index layout as the host forms.
 *   - threading: a context is Send, not Sync (RCN::train takes &mut self, rcn.rs:126; the backend
 *     gives each worker a private model, backend/src/main.rs:64-70): calls on one context must be
 *     serialised by the caller; different contexts are independent.
 */
#ifndef RCN_HIP_H
#define RCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCN_HIP_ABI_VERSION 1

typedef struct rcn_hip_ctx rcn_hip_ctx;

typedef enum rcn_hip_status {
    RCN_HIP_OK = 0,
    RCN_HIP_ERR_INVALID_ARG = -1,  /* NULL pointer, bad enum, bad size                                       */
    RCN_HIP_ERR_SHAPE = -2,        /* the reference panics on this shape (kernel.rs:127,133,156,200,247; gemv dims) */
    RCN_HIP_ERR_UNSUPPORTED = -3,  /* Pooling::Average -> panic!("Not implemented") kernel.rs:283,341; size limits  */
    RCN_HIP_ERR_HIP = -4,          /* a HIP runtime call failed                                             */
    RCN_HIP_ERR_NO_DEVICE = -5,    /* no gfx950 device / device ordinal out of range                        */
    RCN_HIP_ERR_STATE = -6,        /* call order (e.g. train before parameters were set)                    */
    RCN_HIP_ERR_OOM = -7
} rcn_hip_status;

/* enum values = declaration order in the reference = bincode variant tags */
typedef enum { RCN_HIP_PAD_NONE = 0, RCN_HIP_PAD_SAME = 1 } rcn_hip_padding;                 /* utils/kernel.rs:25-28 */
typedef enum { RCN_HIP_POOL_AVERAGE = 0, RCN_HIP_POOL_MAX = 1 } rcn_hip_pooling;             /* utils/kernel.rs:32-35 */
typedef enum { RCN_HIP_OP_TOP = 0, RCN_HIP_OP_BOTTOM = 1, RCN_HIP_OP_LEFT = 2, RCN_HIP_OP_RIGHT = 3 } rcn_hip_sep_op; /* kernel.rs:16-21 */
typedef enum { RCN_HIP_LAYER_CONVOLVE2D = 0, RCN_HIP_LAYER_POOL2D = 1 } rcn_hip_layer_kind;  /* rcn.rs:35-38 */
typedef enum { RCN_HIP_F32 = 0, RCN_HIP_F64 = 1 } rcn_hip_dtype;

/* RCNLayer::Convolve2D(Padding) | RCNLayer::Pool2D(Pooling)                              rcn.rs:35-38 */
typedef struct rcn_hip_layer { int32_t kind; int32_t arg; } rcn_hip_layer;

/* Mirrors the arguments of RCN::new (rcn.rs:58-75) plus what a device context needs. */
typedef struct rcn_hip_cfg {
    uint32_t struct_size;          /* = sizeof(rcn_hip_cfg)                                                   */
    int32_t  device;               /* HIP device ordinal                                                      */
    int32_t  dtype;                /* rcn_hip_dtype: device arithmetic type (the reference is f64 throughout) */
    int32_t  in_h, in_w;           /* input image height / width (rows / cols of get_pixel_matrix, lib.rs:27) */
    int32_t  n_convpool;           /* convpool_cfg.len()                                                      */
    const rcn_hip_layer* convpool; /* convpool_cfg                                                            */
    int32_t  n_hidden;             /* feedforward_cfg.len()  (>= 1: rcn.rs:444 indexes [0])                   */
    const int32_t* hidden;         /* feedforward_cfg                                                         */
    int32_t  classes;              /* classes                                                                 */
    void*    stream;               /* hipStream_t to enqueue on; NULL = the context creates its own stream    */
} rcn_hip_cfg;

/* ---------------------------------------------------------------- lifecycle */
int  rcn_hip_abi_version(void);
const char* rcn_hip_status_string(int status);
/* RCN::new (rcn.rs:58-75) + the dimension bookkeeping of load_weights_and_bias (rcn.rs:425-457).
 * Fails with RCN_HIP_ERR_SHAPE / _UNSUPPORTED where EVERY later use panics in the reference (conv on
 * <3x3 / pool on <2x2 maps, Pooling::Average).  A stack whose feature path works but whose dense part
 * cannot -- first-layer fan-in 4^c/2^p*l (rcn.rs:443) != flattened feature length, so gemv panics at
 * rcn.rs:287, or no Convolve2D layer at all -- creates fine (as RCN::new does); the feature entry points
 * work and every dense entry point returns RCN_HIP_ERR_SHAPE.
 * *out is set even on failure (when non-NULL) so that rcn_hip_last_error can be read; destroy it. */
int  rcn_hip_create(const rcn_hip_cfg* cfg, rcn_hip_ctx** out);
void rcn_hip_destroy(rcn_hip_ctx* ctx);
const char* rcn_hip_last_error(const rcn_hip_ctx* ctx);      /* "" if none; valid until the next call on ctx */
int  rcn_hip_set_stream(rcn_hip_ctx* ctx, void* hip_stream);
int  rcn_hip_synchronize(rcn_hip_ctx* ctx);

/* ---------------------------------------------------------------- introspection */
int  rcn_hip_feature_len(const rcn_hip_ctx* ctx, int64_t* out);               /* len of flatten_feature_set's vector */
int  rcn_hip_num_layers(const rcn_hip_ctx* ctx);                              /* feedforward_cfg.len()+1, rcn.rs:426 */
int  rcn_hip_layer_dims(const rcn_hip_ctx* ctx, int layer, int32_t* rows_out, int32_t* cols_in); /* W_l is rows x cols */
int  rcn_hip_param_count(const rcn_hip_ctx* ctx, int64_t* out);               /* total scalars in all W_l, b_l */

/* ---------------------------------------------------------------- parameters (Weights / Bias, rcn.rs:28,31) */
/* W: rows x cols column-major f64 (= Weights.0 / the bincode `data` field), b: rows f64 (= Bias.0) */
int  rcn_hip_set_params(rcn_hip_ctx* ctx, int layer, const double* W_colmajor, const double* b);
int  rcn_hip_get_params(rcn_hip_ctx* ctx, int layer, double* W_colmajor, double* b);
/* load_weights_and_bias (rcn.rs:425-457): every W, b ~ N(0,1), un-scaled (rcn.rs:500-523).  The reference
 * draws from the unseeded thread_rng; here the stream is a seeded generator (seed 0 = nondeterministic). */
int  rcn_hip_init_params(rcn_hip_ctx* ctx, uint64_t seed);
/* device view of the flat parameter buffer [W_0|b_0|W_1|b_1|...] in the context dtype (for DP broadcast /
 * all-reduce of replicas); valid until destroy */
int  rcn_hip_params_dev(rcn_hip_ctx* ctx, void** dev_ptr, int64_t* count);

/* ---------------------------------------------------------------- operator API (traits Convolve2D / Pool2D) */
/* Pure shape helpers (no device work).  Return RCN_HIP_ERR_SHAPE exactly where the reference panics. */
int  rcn_hip_conv_out_shape(int R, int C, int kr, int kc, int padding, int* out_R, int* out_C);
int  rcn_hip_pool_out_shape(int R, int C, int padding, int* out_R, int* out_C);
/* Convolve2D::convolve_2d (utils/kernel.rs:110-194): cross-correlation of `n` R x C matrices with one
 * kr x kc kernel, Padding::None or Padding::Same (incl. the pad-copy index quirk of :154-158).
 * f64 arithmetic on the device in the reference's summation order.  m: n matrices back to back. */
int  rcn_hip_convolve_2d(rcn_hip_ctx* ctx, const double* m, int n, int R, int C,
                         const double* kernel, int kr, int kc, int padding, double* out);
/* Convolve2D::convolve_2d_separated (kernel.rs:196-207): relu(conv(conv(m, col3x1), row1x3)) */
int  rcn_hip_convolve_2d_separated(rcn_hip_ctx* ctx, const double* m, int n, int R, int C,
                                   int sep_op, int padding, double* out);
/* Convolve2D::relu (kernel.rs:209-216) over `count` scalars */
int  rcn_hip_relu(rcn_hip_ctx* ctx, const double* m, size_t count, double* out);
/* Pool2D::pool_2d (kernel.rs:245-349): 2x2 stride-2 max; Same zero-pads odd dims, None truncates */
int  rcn_hip_pool_2d(rcn_hip_ctx* ctx, const double* m, int n, int R, int C, int padding, int pooling, double* out);

/* ---------------------------------------------------------------- feature pipeline */
/* get_pixel_matrix + flatten_feature_set for `n` images (lib.rs:27-41, rcn.rs:317-356): imgs is
 * n x in_h x in_w u8, row-major pixel order (the order image::pixels() yields); out is n x F f64,
 * per-sample map order / column-major flatten as the reference.  NOT standardised. */
int  rcn_hip_features(rcn_hip_ctx* ctx, const uint8_t* imgs, size_t n, double* out);
/* device form: out_dev n x F in the context dtype.  standardize != 0 additionally applies
 * max((x-mean)/sd, 0) with the context's scale_set (rcn.rs:407-412) in the same kernel. */
int  rcn_hip_features_dev(rcn_hip_ctx* ctx, const uint8_t* imgs_dev, size_t n, void* out_dev, int standardize);
/* gen_scales (rcn.rs:230-251): population mean / sd over all n*F values; also stores them as the context's
 * scale_set, as the reference does (rcn.rs:249-250). */
int  rcn_hip_gen_scales(rcn_hip_ctx* ctx, const double* feats, size_t n, double* mean, double* sd);
int  rcn_hip_gen_scales_dev(rcn_hip_ctx* ctx, const void* feats_dev, size_t n, double* mean, double* sd); /* blocks */
int  rcn_hip_set_scale(rcn_hip_ctx* ctx, double mean, double sd);            /* RCN.scale_set, rcn.rs:21 */
int  rcn_hip_get_scale(const rcn_hip_ctx* ctx, double* mean, double* sd);
/* x <- max((x-mean)/sd, 0) in place (rcn.rs:407-412 / 86-89) */
int  rcn_hip_standardize(rcn_hip_ctx* ctx, double* feats, size_t count);
int  rcn_hip_standardize_dev(rcn_hip_ctx* ctx, void* feats_dev, size_t count);

/* ---------------------------------------------------------------- dense network */
/* train_batch (rcn.rs:176-223): per-sample backprop (rcn.rs:260-314) summed over the batch, then
 * W <- W - (eta/B) sum dW, b <- b - (eta/B) sum db.  x: B x F, y: B x classes (one-hot in the reference,
 * any target accepted), sample-major f64.  loss_out (nullable) receives the quadratic cost
 * 1/(2B) sum ||a_L - y||^2 evaluated before the update (the reference never computes it). */
int  rcn_hip_train_batch(rcn_hip_ctx* ctx, const double* x, const double* y, size_t B, double eta, double* loss_out);
int  rcn_hip_train_batch_dev(rcn_hip_ctx* ctx, const void* x_dev, const void* y_dev, size_t B, double eta,
                             void* loss_dev /* nullable, 1 scalar of the ctx dtype */);
/* The batch loop of RCN::train (rcn.rs:147-149): n_batches consecutive train_batch calls over the resident
 * set X_dev (N x F), Y_dev (N x classes).  Batch j takes samples perm_dev[j*B .. (j+1)*B) (int32 indices
 * into X/Y; NULL = identity, i.e. chunks_exact over the set as stored).  The caller shuffles
 * (rcn.rs:146) by filling perm_dev.  loss_dev: nullable, n_batches scalars.  Launch-bound loop: captured
 * once into a hipGraph per (pointers, B, n_batches, eta) and replayed. */
int  rcn_hip_train_epoch_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev,
                             size_t B, size_t n_batches, double eta, void* loss_dev);
/* Same arguments: captures and instantiates the hipGraph that rcn_hip_train_epoch_dev would replay for them, without
 * running it (one-time set-up cost moved out of the loop; perm_dev's CONTENTS may change between replays, the pointers
 * and sizes may not).  n_batches may span several passes over the set: the caller concatenates one permutation per
 * pass into perm_dev (each pass is rcn.rs:146-149 once). */
int  rcn_hip_prepare_epoch_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev,
                               size_t B, size_t n_batches, double eta, void* loss_dev);
/* (Opt-in form, option "xcd_gather" = 1; measured slower than the packed image on MI355X -- csrc/rcn_hip_api_xcd.ipp.)
 * 1 when rcn_hip_train_epoch_dev at this batch size runs on the resident one-XCD kernel in its GATHER form: no packed epoch image is
 * written -- one launch walks the whole call and every worker fetches its 128 bytes of each row of the batch after next while it
 * works on the current one (csrc/dense_xcd.hpp).  The feature matrix is then read from memory exactly once per step and
 * rcn_hip_epoch_begin_dev's one-off materialisation buys nothing; 0: the call packs its batches segment by segment. */
int  rcn_hip_train_epoch_gathers(rcn_hip_ctx* ctx, size_t B);
/* 1 when rcn_hip_train_epoch_dev / rcn_hip_epoch_steps_dev / rcn_hip_train_set_epoch at this batch size run on the resident one-XCD
 * kernel (csrc/dense_xcd.hpp: the reference's two layer stacks, f32 batches of 1..256, f64 -- the reference's own type, rcn.rs:28-31 --
 * batches of 1..256 too, one XCD verified by the placement probe, not stepped down); 0: the two-kernel pipeline or the sample-tile kernels. */
int  rcn_hip_train_epoch_resident(rcn_hip_ctx* ctx, size_t B);
/* End to end: the same epoch straight from the resident u8 pictures (imgs_dev [N][H][W], perm_dev indexes pictures).  Per
 * segment of the epoch ONE kernel does flatten_feature_set (rcn.rs:317-356), the standardisation with the current scale_set
 * (rcn.rs:407-412) and the gather into the training kernels' layout -- no [N][F] feature matrix is ever written; results
 * are bit-identical to rcn_hip_features_dev(standardize = 1) followed by rcn_hip_train_epoch_dev.  Default conv/pool stack on
 * 28x28 input and the feature-sliced pipeline only (RCN_HIP_ERR_UNSUPPORTED otherwise). */
int  rcn_hip_train_epoch_images_dev(rcn_hip_ctx* ctx, const uint8_t* imgs_dev, const void* Y_dev, const int32_t* perm_dev,
                                    size_t B, size_t n_batches, double eta, void* loss_dev);
int  rcn_hip_prepare_epoch_images_dev(rcn_hip_ctx* ctx, const uint8_t* imgs_dev, const void* Y_dev, const int32_t* perm_dev,
                                      size_t B, size_t n_batches, double eta, void* loss_dev);
/* One epoch of RCN::train the way the reference structures it (rcn.rs:144-149): `training_set.shuffle` happens ONCE, then
 * `chunks_exact(batch_size)` walks it.  rcn_hip_epoch_begin_dev materialises the shuffled order once -- batches 0..n_batches of
 * (X_dev, Y_dev, perm_dev), same meaning as in rcn_hip_train_epoch_dev -- as the training kernels' slice-major epoch image
 * (one gather pass); rcn_hip_epoch_steps_dev then runs train_batch (rcn.rs:176-223) over batches first_batch ..
 * first_batch + n_batches of the begun epoch, in any number of calls, without packing again (one hipGraph per
 * (first_batch, n_batches, eta, loss_dev), captured at first use; rcn_hip_prepare_epoch_steps_dev only instantiates it).
 * The image holds up to two segments (~128 MB) of batches: RCN_HIP_ERR_UNSUPPORTED beyond that, or when the layer stack /
 * batch size does not run on the feature-sliced pipeline -- rcn_hip_train_epoch_dev covers those.  Any other training call
 * that re-packs the image ends the begun epoch (RCN_HIP_ERR_STATE from the next steps call).  Results are bit-identical to
 * rcn_hip_train_epoch_dev over the same batches.  The _images form is rcn_hip_train_epoch_images_dev's begin. */
int  rcn_hip_epoch_begin_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev, size_t B, size_t n_batches);
int  rcn_hip_epoch_begin_images_dev(rcn_hip_ctx* ctx, const uint8_t* imgs_dev, const void* Y_dev, const int32_t* perm_dev, size_t B, size_t n_batches);
int  rcn_hip_epoch_steps_dev(rcn_hip_ctx* ctx, size_t first_batch, size_t n_batches, double eta, void* loss_dev /* nullable, n_batches scalars */);
int  rcn_hip_prepare_epoch_steps_dev(rcn_hip_ctx* ctx, size_t first_batch, size_t n_batches, double eta, void* loss_dev);
/* training_set.shuffle (rcn.rs:146) on the device: writes `passes` independent pseudo-random permutations of 0..n-1
 * (pass p at perm_dev[p*n ..]) keyed by `seed` -- ready to be passed to rcn_hip_train_epoch_dev.  Enqueued on the
 * context's stream (one small kernel); the reference draws from the unseeded thread_rng. */
int  rcn_hip_shuffle_dev(rcn_hip_ctx* ctx, int32_t* perm_dev, size_t n, size_t passes, uint64_t seed);
/* Data-parallel halves of train_batch: summed gradients of one shard into a flat buffer laid out like the
 * parameters (no update), and the update from an (all-reduced) flat gradient: p <- p - scale * g. */
int  rcn_hip_batch_gradient_dev(rcn_hip_ctx* ctx, const void* x_dev, const void* y_dev, size_t B,
                                void* grad_dev, void* loss_sum_dev /* nullable: sum ||a_L-y||^2, not normalised */);
/* same, the shard being rows perm_dev[0..B) of the resident set X_dev / Y_dev (no gather copy) */
int  rcn_hip_batch_gradient_perm_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev, size_t B,
                                     void* grad_dev, void* loss_sum_dev);
int  rcn_hip_apply_gradient_dev(rcn_hip_ctx* ctx, const void* grad_dev, double scale);
/* The same data-parallel step as a native loop over RCCL (xGMI), one rank per context / GPU / process.  The reference
 * has no multi-device mode; this is its train_batch (rcn.rs:176-223) with the batch split over ranks: per step each
 * rank runs the gradient kernels on its B_shard rows, one ncclAllReduce(sum) combines the flat gradient (and the loss),
 * and every rank applies W <- W - eta / (B_shard * world) * sum dW.  Bootstrap: rank 0 calls rcn_hip_dp_unique_id and
 * ships the RCN_HIP_DP_ID_BYTES bytes to the other ranks by any out-of-band channel (MPI, a socket, a file); every
 * rank then calls rcn_hip_dp_init (collective).  RCCL is loaded with dlopen at the first of these calls;
 * RCN_HIP_ERR_UNSUPPORTED if it cannot be found. */
#define RCN_HIP_DP_ID_BYTES 128
int  rcn_hip_dp_unique_id(void* id_out /* RCN_HIP_DP_ID_BYTES */);
int  rcn_hip_dp_init(rcn_hip_ctx* ctx, const void* id, int rank, int world);       /* collective */
int  rcn_hip_dp_finalize(rcn_hip_ctx* ctx);                                         /* also done by rcn_hip_destroy */
int  rcn_hip_dp_world(const rcn_hip_ctx* ctx);
int  rcn_hip_dp_rank(const rcn_hip_ctx* ctx);
int  rcn_hip_dp_broadcast_params(rcn_hip_ctx* ctx, int root);                       /* collective: one model (rcn.rs:139-141) */
/* n_batches global steps; this rank's shard of step j is rows perm_dev[j*B_shard ..] of ITS resident X_dev / Y_dev
 * (perm_dev NULL: consecutive rows).  loss_dev (nullable): n_batches GLOBAL costs sum ||a-y||^2 / (2 B_global).
 * Collective; asynchronous on the context's stream. */
int  rcn_hip_dp_train_epoch_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev,
                                size_t B_shard, size_t n_batches, double eta, void* loss_dev);
/* Same arguments: instantiates (does not run) the hipGraph rcn_hip_dp_train_epoch_dev would replay for them, when the loop
 * is one that replays a graph (the peer-exchange pipeline); a no-op otherwise.  Not collective. */
int  rcn_hip_dp_prepare_epoch_dev(rcn_hip_ctx* ctx, const void* X_dev, const void* Y_dev, const int32_t* perm_dev,
                                  size_t B_shard, size_t n_batches, double eta, void* loss_dev);
/* The all-reduce of that loop.  For 2..8 ranks of one node rcn_hip_dp_init also sets up a one-shot PEER-READ all-reduce
 * over xGMI (every rank reads all ranks' gradient buffers through hipIpc mappings, adds them in rank order and applies
 * the update in the same kernel; csrc/dp_p2p.hpp) and keeps it only if every rank mapped every peer and a known-answer
 * exchange came back exact on every rank; otherwise, or with option "dp_p2p" = 0, the loop uses
 * ncclAllReduce ("dp_p2p" = 2 also sets it up for a single rank, for tests).  rcn_hip_dp_p2p_active tells which.  The three calls below are the same set-up with the handle exchange
 * done by the caller instead of RCCL (any transport; used by the tests): export -> gather every rank's
 * RCN_HIP_DP_P2P_HANDLE_BYTES in rank order -> attach (collective; after it rcn_hip_dp_train_epoch_dev uses the peer
 * all-reduce and needs no RCCL communicator) -> optionally selftest (collective; counts wrong sums over `iters` exchanges
 * of a known pattern, *timed_out != 0 if a wait expired). */
#define RCN_HIP_DP_P2P_HANDLE_BYTES 128
int  rcn_hip_dp_p2p_export(rcn_hip_ctx* ctx, void* handles_out /* RCN_HIP_DP_P2P_HANDLE_BYTES */);
int  rcn_hip_dp_p2p_attach(rcn_hip_ctx* ctx, const void* all_handles /* world x RCN_HIP_DP_P2P_HANDLE_BYTES */, int rank, int world);
int  rcn_hip_dp_p2p_selftest(rcn_hip_ctx* ctx, int iters, unsigned* mismatches, unsigned* timed_out);
/* The whole admission procedure of rcn_hip_dp_init -- export, gather, attach, known-answer exchange, tagged-word self-test, one
 * min-vote over the ranks after every stage so that all ranks land on the same form -- over a transport the caller supplies
 * instead of RCCL: `allgather` receives this rank's `bytes` bytes and must fill `all` with every rank's, in rank order;
 * `vote_min` must replace *v by the minimum of *v over the ranks; both return 0 on success and are called the same number of
 * times on every rank.  Afterwards rcn_hip_dp_p2p_active tells which form was admitted (identical on every rank); with 0 the
 * context has no exchange of its own and the caller combines rcn_hip_batch_gradient_dev / rcn_hip_apply_gradient_dev itself.
 * RCN_HIP_DP_FAULT="<stage>:<rank>,..." (stages export, attach, kat, ll, llskip, push, pushskip) makes a rank fail a stage on purpose (tests). */
typedef int (*rcn_hip_allgather_fn)(void* user, const void* mine, void* all, size_t bytes);
typedef int (*rcn_hip_vote_min_fn)(void* user, int* v);
int  rcn_hip_dp_p2p_admit(rcn_hip_ctx* ctx, int rank, int world, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min, void* user);
/* The same vote sequence WITHOUT a device: every stage succeeds unless `faults` ("<stage>:<rank>,...", the stages above plus nofused,
 * f64, clear) names it for this rank.  Needs no context and no GPU -- a launcher can rehearse its transport with it, and the CPU test
 * (tests/test_dp_gloo.py) drives the native admission logic over gloo with it.  *form_out: 0 / 1 / 2 as rcn_hip_dp_p2p_active,
 * *resident_out (nullable): 1 when the pushed exchange of the resident kernel's data-parallel form was admitted too. */
int  rcn_hip_dp_admission_rehearse(int rank, int world, const char* faults, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min,
                                   void* user, int* form_out, int* resident_out);
/* The data-parallel counterpart of rcn_hip_epoch_steps_dev: n_batches data-parallel steps over this rank's shard batches
 * [first_batch, first_batch + n_batches) of the epoch image rcn_hip_epoch_begin_dev packed on THIS rank (every rank begins its own
 * shard's epoch; nothing is gathered or packed inside the call, so an epoch is packed once however many calls walk it).  Collective
 * like rcn_hip_dp_train_epoch_dev; available where rcn_hip_dp_resident says 1, RCN_HIP_ERR_UNSUPPORTED elsewhere. */
int  rcn_hip_dp_epoch_steps_dev(rcn_hip_ctx* ctx, size_t first_batch, size_t n_batches, double eta, void* loss_dev);
/* 1 when rcn_hip_dp_train_epoch_dev at this shard size runs the data-parallel step on the resident one-XCD kernel with the exchange
 * inside it (csrc/dense_xcd.hpp, DP form; csrc/dp_push.hpp: the shards' partial sums are PUSHED to one owner rank per slice pair of
 * W_0, added there in rank order and pushed back -- a reduce-scatter and an all-gather on self-validating words, 1.75 P words per
 * step and rank at eight ranks instead of the 7 P of an all-to-all read).  Needs: the pushed exchange admitted by its own
 * known-answer vote, an f32 context, one hidden layer, a shard of exactly 32, 64, 128 or 256 samples.  Else 0. */
int  rcn_hip_dp_resident(rcn_hip_ctx* ctx, size_t B_shard);
/* Diagnostic for a first run on more than one GPU: with option "xcd_dp_phase" = 1 the resident kernel's data-parallel launches at a shard
 * of 256 (or 128) carry per-worker clocks (two reads of the 100 MHz clock per step and worker: the figures say where a step WAITS, they are not
 * the cost of an unclocked step).  After such a call: out[0], out[1] = mean / max over the feature workers that OWN their slice pair of
 * the microseconds per step between entering the exchange and having pushed the totals (waiting for the other ranks' partial sums:
 * the reduce-scatter); out[2], out[3] = the same for the member workers (push the partials, wait for the owner's totals: both hops);
 * out[4], out[5] = the tail tiles' all-to-all; out[6] = microseconds per step of the whole loop; out[7] = steps of the launch. Blocks. */
int  rcn_hip_dp_phase_us(rcn_hip_ctx* ctx, double* out, size_t cap);
int  rcn_hip_dp_p2p_active(const rcn_hip_ctx* ctx);   /* 0 ncclAllReduce, 1 peer exchange at kernel boundaries, 2 also inside the gradient kernel */
/* ---- RCN::train's data flow with both data sets RESIDENT in HBM (rcn.rs:126-167): what a host-language `RCN::train` calls.
 * rcn_hip_load_data = the arithmetic of load_data after the image decode (rcn.rs:399-414) for `n` images: imgs n x in_h x in_w u8,
 * labels n class indices (position in the sorted directory listing, rcn.rs:374-377,401) -> flatten_feature_set, gen_scales (this
 * OVERWRITES the context's scale_set, as the reference does: after train + test loads it holds the TEST set's statistics,
 * rcn.rs:134-137), standardise + clamp, one-hot expectations (rcn.rs:466-471); features and expectations stay on the device in
 * slot 0 (training set) / 1 (testing set).  Blocks.  mean / sd (nullable) receive the set's statistics.
 * rcn_hip_train_set_epoch = one pass of rcn.rs:146-149 over a loaded slot: `perm` (host, nullable) is the shuffled order -- the
 * first floor(n / B) * B entries are used, chunks_exact drops the tail -- or NULL to shuffle on the device with `shuffle_seed`
 * (0 = non-deterministic like thread_rng); loss_out (host, nullable) receives the floor(n / B) per-step costs (then the call blocks).
 * rcn_hip_evaluate_set = the per-epoch accuracy count of rcn.rs:152-157 over a loaded slot. */
int  rcn_hip_load_data(rcn_hip_ctx* ctx, int slot, const uint8_t* imgs, const int32_t* labels, size_t n, double* mean, double* sd);
int  rcn_hip_train_set_epoch(rcn_hip_ctx* ctx, int slot, const int32_t* perm, uint64_t shuffle_seed, size_t B, double eta, double* loss_out);
int  rcn_hip_evaluate_set(rcn_hip_ctx* ctx, int slot, int64_t* accepted);
int  rcn_hip_set_size(const rcn_hip_ctx* ctx, int slot, int64_t* n);
/* classify_test (rcn.rs:105-116) for n samples: a <- sigmoid(W a + b) through every layer. out: n x classes */
int  rcn_hip_forward(rcn_hip_ctx* ctx, const double* x, size_t n, double* out);
int  rcn_hip_forward_dev(rcn_hip_ctx* ctx, const void* x_dev, size_t n, void* out_dev);
/* arg-max of classify (rcn.rs:92-97): max_by(total_cmp) keeps the LAST maximal index */
int  rcn_hip_classify(rcn_hip_ctx* ctx, const double* x, size_t n, int32_t* class_out);
/* per-epoch evaluation of RCN::train (rcn.rs:152-157): counts samples whose one-hot(v == max) equals y */
int  rcn_hip_evaluate(rcn_hip_ctx* ctx, const double* x, const double* y, size_t n, int64_t* accepted);
int  rcn_hip_evaluate_dev(rcn_hip_ctx* ctx, const void* x_dev, const void* y_dev, size_t n, int64_t* accepted); /* blocks */
/* RCN::classify minus the PNG decode (rcn.rs:82-98): features -> standardise with scale_set -> forward -> arg-max */
int  rcn_hip_classify_images(rcn_hip_ctx* ctx, const uint8_t* imgs, size_t n, int32_t* class_out);

/* ---------------------------------------------------------------- tuning / measurement aids */
/* Which kernels implement train_batch / train_epoch: 0 = automatic (the feature-sliced pipeline for batches <= 1024 when the layer
 * stack allows it -- as ONE resident kernel per epoch segment whose workgroups share one XCD and hand over through its L2
 * (csrc/dense_xcd.hpp) where that form applies: f32 context, one hidden layer <= 32 (or two: <= 32, <= 16), classes <= 16, ANY
 * batch of 1..256 samples (the reference trains at batch_size 10, rcn/src/main.rs:36-37; instantiations for 32 / 64 / 128 / 256,
 * smaller batches padded with masked samples), and a device on which a placement probe finds the blocks with blockIdx.x % 8 == 0
 * on one XCD; as two kernels per step (csrc/dense_p2.hpp, dense_pipe.hpp) otherwise -- and the sample-tile kernels for everything
 * else), 1 = always the sample-tile kernels, 2 = the feature-sliced pipeline as two kernels per step, 5 = the resident one-XCD
 * kernel (RCN_HIP_ERR_UNSUPPORTED where it does not apply; option "xcd" = 0 keeps mode 0 off it).
 * The resident kernel needs its workgroups on the GPU at once.  On a device shared with another process one of its bounded waits
 * can expire; nothing such a launch computed reaches the parameters, and a single-GPU context then STEPS DOWN by itself: at the
 * next point where the library drains the stream (rcn_hip_synchronize, rcn_hip_get_params, rcn_hip_evaluate*, a call that returns
 * costs to the host, or the next training call that finds the failure) it clears the error, selects the two-kernel pipeline for the
 * rest of the context's life and re-runs there every step that was not applied, from the arguments the calls were given -- index
 * rows the library shuffled or uploaded itself are re-created, everything else those calls read must be unchanged, the usual
 * contract of an asynchronous call.  rcn_hip_fallbacks_taken counts these step-downs; option "xcd_auto_fallback" = 0 restores the
 * sticky RCN_HIP_ERR_HIP (cleared by rcn_hip_set_params / rcn_hip_init_params / rcn_hip_set_dense_path(ctx, 1 or 2)).
 * 3 and 4 are parked experiments compiled only into librcn_hip_exp.so.  All compute the same step (summation grouping differs,
 * within the stated tolerances). */
int  rcn_hip_set_dense_path(rcn_hip_ctx* ctx, int mode);
int  rcn_hip_fallbacks_taken(const rcn_hip_ctx* ctx);      /* not a status: the number of step-downs described above */
/* The record of the newest bounded wait of the resident one-XCD kernel that expired in this context -- written by the first worker that
 * gave up (csrc/dense_xcd.hpp: xcd_raise), kept after the library has healed the context.  words[0..12] = site (1 placement vote,
 * 2 tail-tile flag, 3 slab flag, 4 delta flag, 5 pushed reduce-scatter, 6 pushed all-gather, 7 tail all-to-all, 8 cost all-to-all,
 * 9 closing round), worker, step within the launch, launch id, missing producers / workers / ranks (low, high 32 bits; closing round:
 * the arrivals seen; sites 5-8: the low word is the rank mask, the high word the waiting lane's first parameter index -- the text then
 * also says which exchange steps this rank's memory holds at the awaited words), awaited tag or exchange step, XCC_ID of the worker, rank, world, which blocks were workers (blockIdx % 8), workers
 * of the launch, error code (1 expired, 2 workers on different XCDs).  Returns the number of words written, 0 when nothing is on record.
 * _text: the same in words plus the workspace's placement / flag tables as the failed launch left them ("" when nothing is on record). */
int  rcn_hip_last_timeout(const rcn_hip_ctx* ctx, uint32_t* words, size_t cap);
const char* rcn_hip_last_timeout_text(const rcn_hip_ctx* ctx);

/* Per-context options.  The environment variable named beside an option only seeds its default when the context is created; nothing
 * reads the environment afterwards, so two contexts of one process can differ.  RCN_HIP_ERR_INVALID_ARG for an unknown name or a value
 * out of range.  Changing an option drops the context's captured graphs.
 *   "xcd"                 RCN_HIP_XCD                 0 | 1     dense path 0 may select the resident one-XCD kernel (1)
 *   "xcd_select"          RCN_HIP_XCD_SELECT          0..15     TEST-ONLY (ranks sharing ONE device in the test harness): which blocks of a resident
 *                                                               launch are its workers -- 0..7: blockIdx.x % 8 == value; 8..15: the blocks that landed on
 *                                                               PHYSICAL XCD value - 8 (which XCD a dispatch starts its round-robin on differs from queue
 *                                                               to queue: profiles/r4_dp_4rank_timeout_record.txt); a GPU per rank keeps 0
 *   "xcd_gather"          RCN_HIP_XCD_GATHER          0 | 1     rows fetched by the resident kernel itself, batch 256 (measured slower; 0)
 *   "xcd_timeout_ticks"   RCN_HIP_XCD_TIMEOUT_TICKS   >= 1      bound of every wait inside the resident kernel, 100 MHz ticks (20000000 = 0.2 s)
 *   "xcd_exact_lds"       RCN_HIP_XCD_EXACT_LDS       0 | 1     TEST-ONLY (same harness): the resident kernel asks for exactly the LDS it uses, so that
 *                                                               two contexts' kernels fit on one device at batches <= 64 (0: one worker per CU)
 *   "xcd_auto_fallback"   RCN_HIP_XCD_AUTO_FALLBACK   0 | 1     self-healing step-down of the single-GPU resident kernel (1)
 *   "xcd_replay_caller_rows" RCN_HIP_XCD_REPLAY_CALLER_ROWS 0 | 1  the step-down may also re-run steps whose index rows the CALLER wrote (taken as unchanged
 *                                                               since the call); 0: only stored-order calls and rows from rcn_hip_shuffle_dev / an upload of
 *                                                               the library are re-run, anything else keeps the sticky error (0)
 *   "xcd_fault_launch"    RCN_HIP_XCD_FAULT_LAUNCH    >= 0      TEST HOOK, never set in production: the n-th resident launch of the context fails (0: none)
 *   "xcd_fault_mode"      RCN_HIP_XCD_FAULT_MODE      0 | 1     TEST HOOK: how -- 0 a worker never becomes resident, 1 a worker reaches the closing round late
 *   "xcd_dp_phase"        RCN_HIP_XCD_DP_PHASE        0 | 1     DIAGNOSTIC: data-parallel launches at a shard of 256 carry phase clocks (rcn_hip_dp_phase_us; 0)
 *   "dp_p2p"              RCN_HIP_DP_P2P              0 | 1 | 2 peer exchange over xGMI: never / when world > 1 / also at world 1 (tests)
 *   "dp_fused"            RCN_HIP_DP_FUSED            0 | 1     the exchange may run inside a step kernel (1)
 *   "dp_timeout_ticks"    RCN_HIP_DP_TIMEOUT_TICKS    >= 1      bound of a peer wait (100000000 = 1 s)
 *   "dp_cached_buf"       RCN_HIP_DP_CACHED_BUF       0 | 1     exported buffers in cached device memory (A/B measurements; 0)
 *   "dp_graph"            RCN_HIP_DP_GRAPH            0 | 1     the three-kernel data-parallel step replays as one hipGraph (1)
 *   "feat_waves"          RCN_HIP_FEAT_WAVES          1 | 2     waves per picture in k_features_cpcp (measured neutral; 1)
 *   "no_fragimg"          RCN_HIP_NO_FRAGIMG          0 | 1     k_p2_b gathers its tail parameters itself (0)
 *   "exact_div_only"      RCN_HIP_EXACT_DIV_ONLY      0 | 1     the f32 standardisation always divides (0)
 *   "pack_segment_bytes"  RCN_HIP_PACK_SEGMENT_BYTES  >= 1      one half of the epoch image (64 MiB) */
int  rcn_hip_set_option(rcn_hip_ctx* ctx, const char* name, int64_t value);
int  rcn_hip_get_option(const rcn_hip_ctx* ctx, const char* name, int64_t* value);
/* Which kernel implements flatten_feature_set: 0 = automatic (the fused conv+pool kernel specialised for the default
 * stack conv(Same),pool(Max),conv(Same),pool(Max) on 28x28 input when the configuration is exactly that, the generic
 * layer-walking kernel otherwise), 1 = always the generic kernel.  Both are bit-identical (integer-valued arithmetic). */
int  rcn_hip_set_feature_kernel(rcn_hip_ctx* ctx, int mode);

/* Times the two kernels of one train_batch at batch size B with HIP events on the context's stream: `reps`
 * back-to-back launches of each kernel (one hipGraph of `reps` dependent nodes per kernel, so the host launch rate does
 * not bound the result), events recorded immediately before and after each graph.  Returns the mean microseconds per
 * launch.  us_first / us_second are (k_dense_fwd, k_dense_wgrad) on the sample-tile path and (k_pipe_b, k_pipe_a) on
 * the feature-sliced path; updates run with a zero step so the parameters do not drift while timing.  On the
 * feature-sliced path, when the context still holds the packed image of an epoch at this batch size (i.e. right after
 * rcn_hip_train_epoch_dev), successive launches walk that epoch's batches, so the timing includes the same cold reads as
 * the real loop.  us_pair (nullable) times `reps` repetitions of (first, second) alternating, as the real loop issues
 * them: the two kernels of a step cost more alternating than each back to back with itself (+1.7 us per pair measured),
 * so a profiler's per-dispatch average corresponds to us_pair split in the ratio us_first : us_second.  Where the resident
 * one-XCD kernel runs the step there is no first / second kernel: one launch runs all steps of the packed image's first
 * segment; *us_first = 0 and *us_second = *us_pair = that launch's duration divided by its number of steps.  Blocks. */
int  rcn_hip_time_kernels_dev(rcn_hip_ctx* ctx, const void* x_dev, const void* y_dev, size_t B, int reps,
                              double* us_first, double* us_second, double* us_pair /* nullable: the two alternating */);

#ifdef __cplusplus
}
#endif
#endif /* RCN_HIP_H */
