//! Raw FFI declarations for `include/rcn_hip.h` plus a thin safe wrapper.
//!
//! **Not compiled or tested here** -- this repository's environment has no Rust toolchain.  The C ABI itself is
//! exercised through Python `ctypes` (tests/) with the same signatures; keep this file in lock-step with the header.
//!
//! Each function names the reference code it replaces (file:line under `rcn/src/`).
#![allow(non_camel_case_types)]

use std::ffi::{c_char, c_double, c_int, c_void, CStr};

#[repr(C)]
pub struct rcn_hip_ctx {
    _private: [u8; 0],
}

pub const RCN_HIP_OK: c_int = 0;
pub const RCN_HIP_ERR_SHAPE: c_int = -2; // the reference panics (kernel.rs:127,133,156,200,247; gemv dims)
pub const RCN_HIP_ERR_UNSUPPORTED: c_int = -3; // Pooling::Average -> panic!("Not implemented")

pub const RCN_HIP_F32: i32 = 0;
pub const RCN_HIP_F64: i32 = 1;
pub const RCN_HIP_LAYER_CONVOLVE2D: i32 = 0; // rcn.rs:35-38 (bincode variant indices)
pub const RCN_HIP_LAYER_POOL2D: i32 = 1;
pub const RCN_HIP_PAD_NONE: i32 = 0; // utils/kernel.rs:25-28
pub const RCN_HIP_PAD_SAME: i32 = 1;
pub const RCN_HIP_POOL_AVERAGE: i32 = 0; // utils/kernel.rs:32-35
pub const RCN_HIP_POOL_MAX: i32 = 1;

/// `RCNLayer::Convolve2D(Padding) | RCNLayer::Pool2D(Pooling)` (rcn.rs:35-38); tags = bincode variant indices.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rcn_hip_layer {
    pub kind: i32, // 0 Convolve2D, 1 Pool2D
    pub arg: i32,  // Padding: 0 None, 1 Same (kernel.rs:25-28) | Pooling: 0 Average, 1 Max (kernel.rs:32-35)
}

#[repr(C)]
pub struct rcn_hip_cfg {
    pub struct_size: u32,
    pub device: i32,
    pub dtype: i32,
    pub in_h: i32,
    pub in_w: i32,
    pub n_convpool: i32,
    pub convpool: *const rcn_hip_layer,
    pub n_hidden: i32,
    pub hidden: *const i32,
    pub classes: i32,
    pub stream: *mut c_void,
}

extern "C" {
    pub fn rcn_hip_abi_version() -> c_int;
    pub fn rcn_hip_status_string(status: c_int) -> *const c_char;
    pub fn rcn_hip_create(cfg: *const rcn_hip_cfg, out: *mut *mut rcn_hip_ctx) -> c_int; // RCN::new, rcn.rs:58-75
    pub fn rcn_hip_destroy(ctx: *mut rcn_hip_ctx);
    pub fn rcn_hip_last_error(ctx: *const rcn_hip_ctx) -> *const c_char;
    pub fn rcn_hip_set_stream(ctx: *mut rcn_hip_ctx, stream: *mut c_void) -> c_int;
    pub fn rcn_hip_synchronize(ctx: *mut rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_feature_len(ctx: *const rcn_hip_ctx, out: *mut i64) -> c_int;
    pub fn rcn_hip_num_layers(ctx: *const rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_layer_dims(ctx: *const rcn_hip_ctx, layer: c_int, rows: *mut i32, cols: *mut i32) -> c_int;
    pub fn rcn_hip_param_count(ctx: *const rcn_hip_ctx, out: *mut i64) -> c_int;
    pub fn rcn_hip_set_params(ctx: *mut rcn_hip_ctx, layer: c_int, w_colmajor: *const c_double, b: *const c_double) -> c_int;
    pub fn rcn_hip_get_params(ctx: *mut rcn_hip_ctx, layer: c_int, w_colmajor: *mut c_double, b: *mut c_double) -> c_int;
    pub fn rcn_hip_init_params(ctx: *mut rcn_hip_ctx, seed: u64) -> c_int; // load_weights_and_bias, rcn.rs:425-457
    pub fn rcn_hip_params_dev(ctx: *mut rcn_hip_ctx, dev_ptr: *mut *mut c_void, count: *mut i64) -> c_int;
    pub fn rcn_hip_conv_out_shape(r: c_int, c: c_int, kr: c_int, kc: c_int, padding: c_int, or: *mut c_int, oc: *mut c_int) -> c_int;
    pub fn rcn_hip_pool_out_shape(r: c_int, c: c_int, padding: c_int, or: *mut c_int, oc: *mut c_int) -> c_int;
    // Convolve2D / Pool2D traits, utils/kernel.rs:110-216, 245-349
    pub fn rcn_hip_convolve_2d(ctx: *mut rcn_hip_ctx, m: *const c_double, n: c_int, r: c_int, c: c_int, kernel: *const c_double,
                               kr: c_int, kc: c_int, padding: c_int, out: *mut c_double) -> c_int;
    pub fn rcn_hip_convolve_2d_separated(ctx: *mut rcn_hip_ctx, m: *const c_double, n: c_int, r: c_int, c: c_int, sep_op: c_int,
                                         padding: c_int, out: *mut c_double) -> c_int;
    pub fn rcn_hip_relu(ctx: *mut rcn_hip_ctx, m: *const c_double, count: usize, out: *mut c_double) -> c_int;
    pub fn rcn_hip_pool_2d(ctx: *mut rcn_hip_ctx, m: *const c_double, n: c_int, r: c_int, c: c_int, padding: c_int, pooling: c_int,
                           out: *mut c_double) -> c_int;
    // flatten_feature_set / gen_scales / standardise, rcn.rs:317-356, 230-251, 407-412
    pub fn rcn_hip_features(ctx: *mut rcn_hip_ctx, imgs: *const u8, n: usize, out: *mut c_double) -> c_int;
    pub fn rcn_hip_features_dev(ctx: *mut rcn_hip_ctx, imgs_dev: *const u8, n: usize, out_dev: *mut c_void, standardize: c_int) -> c_int;
    pub fn rcn_hip_gen_scales(ctx: *mut rcn_hip_ctx, feats: *const c_double, n: usize, mean: *mut c_double, sd: *mut c_double) -> c_int;
    pub fn rcn_hip_gen_scales_dev(ctx: *mut rcn_hip_ctx, feats_dev: *const c_void, n: usize, mean: *mut c_double, sd: *mut c_double) -> c_int;
    pub fn rcn_hip_set_scale(ctx: *mut rcn_hip_ctx, mean: c_double, sd: c_double) -> c_int;
    pub fn rcn_hip_get_scale(ctx: *const rcn_hip_ctx, mean: *mut c_double, sd: *mut c_double) -> c_int;
    pub fn rcn_hip_standardize(ctx: *mut rcn_hip_ctx, feats: *mut c_double, count: usize) -> c_int;
    pub fn rcn_hip_standardize_dev(ctx: *mut rcn_hip_ctx, feats_dev: *mut c_void, count: usize) -> c_int;
    // train_batch / classify_test / the epoch loop, rcn.rs:176-223, 105-116, 144-165
    pub fn rcn_hip_train_batch(ctx: *mut rcn_hip_ctx, x: *const c_double, y: *const c_double, b: usize, eta: c_double, loss_out: *mut c_double) -> c_int;
    pub fn rcn_hip_train_batch_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, b: usize, eta: c_double, loss_dev: *mut c_void) -> c_int;
    pub fn rcn_hip_train_epoch_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b: usize, n_batches: usize,
                                   eta: c_double, loss_dev: *mut c_void) -> c_int;
    pub fn rcn_hip_prepare_epoch_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b: usize, n_batches: usize,
                                     eta: c_double, loss_dev: *mut c_void) -> c_int;
    // one epoch as the reference structures it: shuffle once (rcn.rs:146), then walk the chunks (rcn.rs:147-149)
    pub fn rcn_hip_epoch_begin_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b: usize, n_batches: usize) -> c_int;
    pub fn rcn_hip_epoch_begin_images_dev(ctx: *mut rcn_hip_ctx, imgs: *const u8, y: *const c_void, perm: *const i32, b: usize, n_batches: usize) -> c_int;
    pub fn rcn_hip_epoch_steps_dev(ctx: *mut rcn_hip_ctx, first_batch: usize, n_batches: usize, eta: c_double, loss_dev: *mut c_void) -> c_int;
    pub fn rcn_hip_prepare_epoch_steps_dev(ctx: *mut rcn_hip_ctx, first_batch: usize, n_batches: usize, eta: c_double, loss_dev: *mut c_void) -> c_int;
    pub fn rcn_hip_shuffle_dev(ctx: *mut rcn_hip_ctx, perm_dev: *mut i32, n: usize, passes: usize, seed: u64) -> c_int; // rcn.rs:146
    pub fn rcn_hip_batch_gradient_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, b: usize, grad: *mut c_void, loss_sum: *mut c_void) -> c_int;
    pub fn rcn_hip_batch_gradient_perm_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b: usize, grad: *mut c_void,
                                           loss_sum: *mut c_void) -> c_int;
    pub fn rcn_hip_apply_gradient_dev(ctx: *mut rcn_hip_ctx, grad: *const c_void, scale: c_double) -> c_int;
    pub fn rcn_hip_train_epoch_images_dev(ctx: *mut rcn_hip_ctx, imgs: *const u8, y: *const c_void, perm: *const i32, b: usize, n_batches: usize, eta: c_double,
                                          loss: *mut c_void) -> c_int;
    pub fn rcn_hip_prepare_epoch_images_dev(ctx: *mut rcn_hip_ctx, imgs: *const u8, y: *const c_void, perm: *const i32, b: usize, n_batches: usize, eta: c_double,
                                            loss: *mut c_void) -> c_int;
    pub fn rcn_hip_dp_unique_id(id_out: *mut c_void) -> c_int;
    pub fn rcn_hip_dp_init(ctx: *mut rcn_hip_ctx, id: *const c_void, rank: c_int, world: c_int) -> c_int;
    pub fn rcn_hip_dp_finalize(ctx: *mut rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_dp_world(ctx: *const rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_dp_rank(ctx: *const rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_dp_broadcast_params(ctx: *mut rcn_hip_ctx, root: c_int) -> c_int;
    pub fn rcn_hip_dp_p2p_export(ctx: *mut rcn_hip_ctx, handles_out: *mut c_void) -> c_int;
    pub fn rcn_hip_dp_p2p_attach(ctx: *mut rcn_hip_ctx, all_handles: *const c_void, rank: c_int, world: c_int) -> c_int;
    pub fn rcn_hip_dp_p2p_selftest(ctx: *mut rcn_hip_ctx, iters: c_int, mismatches: *mut u32, timed_out: *mut u32) -> c_int;
    pub fn rcn_hip_dp_p2p_admit(ctx: *mut rcn_hip_ctx, rank: c_int, world: c_int,
                                allgather: Option<unsafe extern "C" fn(user: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: usize) -> c_int>,
                                vote_min: Option<unsafe extern "C" fn(user: *mut c_void, v: *mut c_int) -> c_int>, user: *mut c_void) -> c_int;
    pub fn rcn_hip_dp_admission_rehearse(rank: c_int, world: c_int, faults: *const c_char,
                                         allgather: Option<unsafe extern "C" fn(user: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: usize) -> c_int>,
                                         vote_min: Option<unsafe extern "C" fn(user: *mut c_void, v: *mut c_int) -> c_int>, user: *mut c_void,
                                         form_out: *mut c_int, resident_out: *mut c_int) -> c_int;
    pub fn rcn_hip_dp_epoch_steps_dev(ctx: *mut rcn_hip_ctx, first_batch: usize, n_batches: usize, eta: f64, loss_dev: *mut c_void) -> c_int;
    pub fn rcn_hip_train_epoch_gathers(ctx: *mut rcn_hip_ctx, b: usize) -> c_int;
    pub fn rcn_hip_train_epoch_resident(ctx: *mut rcn_hip_ctx, b: usize) -> c_int;
    pub fn rcn_hip_dp_resident(ctx: *mut rcn_hip_ctx, b_shard: usize) -> c_int;
    pub fn rcn_hip_dp_phase_us(ctx: *mut rcn_hip_ctx, out: *mut c_double, cap: usize) -> c_int;
    pub fn rcn_hip_dp_p2p_active(ctx: *const rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_set_feature_kernel(ctx: *mut rcn_hip_ctx, mode: c_int) -> c_int;
    pub fn rcn_hip_dp_prepare_epoch_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b_shard: usize, n_batches: usize,
                                        eta: c_double, loss: *mut c_void) -> c_int;
    pub fn rcn_hip_dp_train_epoch_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, perm: *const i32, b_shard: usize, n_batches: usize,
                                      eta: c_double, loss: *mut c_void) -> c_int;
    // RCN::train's data flow with both sets resident in HBM (rcn.rs:126-167)
    pub fn rcn_hip_load_data(ctx: *mut rcn_hip_ctx, slot: c_int, imgs: *const u8, labels: *const i32, n: usize, mean: *mut c_double, sd: *mut c_double) -> c_int;
    pub fn rcn_hip_train_set_epoch(ctx: *mut rcn_hip_ctx, slot: c_int, perm: *const i32, shuffle_seed: u64, b: usize, eta: c_double, loss_out: *mut c_double) -> c_int;
    pub fn rcn_hip_evaluate_set(ctx: *mut rcn_hip_ctx, slot: c_int, accepted: *mut i64) -> c_int;
    pub fn rcn_hip_set_size(ctx: *const rcn_hip_ctx, slot: c_int, n: *mut i64) -> c_int;
    pub fn rcn_hip_forward(ctx: *mut rcn_hip_ctx, x: *const c_double, n: usize, out: *mut c_double) -> c_int;
    pub fn rcn_hip_forward_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, n: usize, out: *mut c_void) -> c_int;
    pub fn rcn_hip_classify(ctx: *mut rcn_hip_ctx, x: *const c_double, n: usize, class_out: *mut i32) -> c_int;
    pub fn rcn_hip_evaluate(ctx: *mut rcn_hip_ctx, x: *const c_double, y: *const c_double, n: usize, accepted: *mut i64) -> c_int;
    pub fn rcn_hip_evaluate_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, n: usize, accepted: *mut i64) -> c_int;
    pub fn rcn_hip_classify_images(ctx: *mut rcn_hip_ctx, imgs: *const u8, n: usize, class_out: *mut i32) -> c_int;
    pub fn rcn_hip_set_dense_path(ctx: *mut rcn_hip_ctx, mode: c_int) -> c_int;
    pub fn rcn_hip_fallbacks_taken(ctx: *const rcn_hip_ctx) -> c_int;
    pub fn rcn_hip_last_timeout(ctx: *const rcn_hip_ctx, words: *mut u32, cap: usize) -> c_int;
    pub fn rcn_hip_last_timeout_text(ctx: *const rcn_hip_ctx) -> *const c_char;
    pub fn rcn_hip_set_option(ctx: *mut rcn_hip_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn rcn_hip_get_option(ctx: *const rcn_hip_ctx, name: *const c_char, value: *mut i64) -> c_int;
    pub fn rcn_hip_time_kernels_dev(ctx: *mut rcn_hip_ctx, x: *const c_void, y: *const c_void, b: usize, reps: c_int,
                                    us_first: *mut c_double, us_second: *mut c_double, us_pair: *mut c_double) -> c_int;
}

/// Error carrying the status and the library's message.  A `Shape` / `Unsupported` status is where the pure-Rust
/// reference would have panicked; the wrapper turns it back into a panic so callers observe identical behaviour.
#[derive(Debug)]
pub struct HipError {
    pub status: i32,
    pub message: String,
}

impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rcn_hip status {}: {}", self.status, self.message)
    }
}

impl std::error::Error for HipError {}

/// `rcn_hip_last_timeout`: site 1 placement vote, 2 tail-tile flag, 3 slab flag, 4 delta flag, 5 pushed reduce-scatter, 6 pushed all-gather,
/// 7 tail all-to-all, 8 cost all-to-all, 9 closing round; `missing` = the producers / workers / ranks the wait was still missing
/// (closing round: the arrivals seen; sites 5-8: rank mask in the low half, the waiting lane's first parameter index in the high half); `text` adds the workspace's placement and flag tables as the failed launch left them.
#[derive(Debug, Clone)]
pub struct TimeoutRecord {
    pub site: u32,
    pub worker: u32,
    pub step: i32,
    pub launch: u32,
    pub missing: u64,
    pub tag: u32,
    pub xcc: u32,
    pub rank: u32,
    pub world: u32,
    pub workers: u32,
    pub code: u32,
    pub text: String,
}

/// Safe owner of one context.  `Send` but not `Sync`, like `RCN` behind `&mut self` (rcn.rs:126).
pub struct Context {
    raw: *mut rcn_hip_ctx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(classes: usize, convpool: &[rcn_hip_layer], hidden: &[usize], in_h: usize, in_w: usize, dtype: i32, device: i32) -> Result<Self, HipError> {
        let hidden_i32: Vec<i32> = hidden.iter().map(|&h| h as i32).collect();
        let cfg = rcn_hip_cfg {
            struct_size: std::mem::size_of::<rcn_hip_cfg>() as u32,
            device,
            dtype,
            in_h: in_h as i32,
            in_w: in_w as i32,
            n_convpool: convpool.len() as i32,
            convpool: convpool.as_ptr(),
            n_hidden: hidden_i32.len() as i32,
            hidden: hidden_i32.as_ptr(),
            classes: classes as i32,
            stream: std::ptr::null_mut(),
        };
        let mut raw = std::ptr::null_mut();
        let st = unsafe { rcn_hip_create(&cfg, &mut raw) };
        let ctx = Context { raw };
        if st != RCN_HIP_OK {
            return Err(ctx.error(st));
        }
        Ok(ctx)
    }

    fn error(&self, status: i32) -> HipError {
        let msg = unsafe {
            let p = if self.raw.is_null() { rcn_hip_status_string(status) } else { rcn_hip_last_error(self.raw) };
            CStr::from_ptr(p).to_string_lossy().into_owned()
        };
        HipError { status, message: msg }
    }

    /// Maps a status to `Ok`, an `Err`, or -- for the statuses that stand for a reference `panic!` -- a panic.
    fn check(&self, status: i32) -> Result<(), HipError> {
        match status {
            RCN_HIP_OK => Ok(()),
            RCN_HIP_ERR_SHAPE | RCN_HIP_ERR_UNSUPPORTED => panic!("{}", self.error(status).message),
            s => Err(self.error(s)),
        }
    }

    pub fn feature_len(&self) -> usize {
        let mut n = 0i64;
        unsafe { rcn_hip_feature_len(self.raw, &mut n) };
        n as usize
    }

    /// `flatten_feature_set` for one decoded grayscale image (row-major pixels as `image::pixels()` yields them).
    pub fn features(&mut self, pixels: &[u8], out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_features(self.raw, pixels.as_ptr(), 1, out.as_mut_ptr()) };
        self.check(st)
    }

    /// `train_batch` (rcn.rs:176-223) on a batch already flattened to sample-major `x` / `y`.
    pub fn train_batch(&mut self, x: &[f64], y: &[f64], batch: usize, eta: f64) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_train_batch(self.raw, x.as_ptr(), y.as_ptr(), batch, eta, std::ptr::null_mut()) };
        self.check(st)
    }

    /// `classify_test` (rcn.rs:105-116).
    pub fn forward(&mut self, x: &[f64], n: usize, out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_forward(self.raw, x.as_ptr(), n, out.as_mut_ptr()) };
        self.check(st)
    }

    /// `Weights.0` / `Bias.0` of layer `l` in nalgebra's own storage order (`DMatrix::as_slice()` is column-major).
    pub fn set_params(&mut self, layer: usize, w_colmajor: &[f64], b: &[f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_set_params(self.raw, layer as c_int, w_colmajor.as_ptr(), b.as_ptr()) };
        self.check(st)
    }

    pub fn get_params(&mut self, layer: usize, w_colmajor: &mut [f64], b: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_get_params(self.raw, layer as c_int, w_colmajor.as_mut_ptr(), b.as_mut_ptr()) };
        self.check(st)
    }

    pub fn raw(&mut self) -> *mut rcn_hip_ctx {
        self.raw
    }

    // ---- operator API (traits Convolve2D / Pool2D, utils/kernel.rs:61-100, 219-236); matrices f64 column-major ----
    /// Output shape of `convolve_2d`; panics exactly where the reference does (utils/kernel.rs:123-135).
    pub fn conv_out_shape(&self, r: usize, c: usize, kr: usize, kc: usize, padding: i32) -> (usize, usize) {
        let (mut or, mut oc) = (0 as c_int, 0 as c_int);
        let st = unsafe { rcn_hip_conv_out_shape(r as c_int, c as c_int, kr as c_int, kc as c_int, padding, &mut or, &mut oc) };
        if st != RCN_HIP_OK {
            panic!("convolve_2d expects 'self.shape() >= kernel_shape() > 0' and an odd kernel under Padding::Same (utils/kernel.rs:123-135)");
        }
        (or as usize, oc as usize)
    }

    /// Output shape of `pool_2d`; panics for matrices below 2x2 (utils/kernel.rs:246-251).
    pub fn pool_out_shape(&self, r: usize, c: usize, padding: i32) -> (usize, usize) {
        let (mut or, mut oc) = (0 as c_int, 0 as c_int);
        let st = unsafe { rcn_hip_pool_out_shape(r as c_int, c as c_int, padding, &mut or, &mut oc) };
        if st != RCN_HIP_OK {
            panic!("stride_2d expected a matrix with dimensions greater than (2, 2), got ({}, {})", r, c);
        }
        (or as usize, oc as usize)
    }

    #[allow(clippy::too_many_arguments)]
    pub fn convolve_2d(&mut self, m: &[f64], r: usize, c: usize, kernel: &[f64], kr: usize, kc: usize, padding: i32, out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe {
            rcn_hip_convolve_2d(self.raw, m.as_ptr(), 1, r as c_int, c as c_int, kernel.as_ptr(), kr as c_int, kc as c_int, padding, out.as_mut_ptr())
        };
        self.check(st)
    }

    pub fn convolve_2d_separated(&mut self, m: &[f64], r: usize, c: usize, sep_op: i32, padding: i32, out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_convolve_2d_separated(self.raw, m.as_ptr(), 1, r as c_int, c as c_int, sep_op, padding, out.as_mut_ptr()) };
        self.check(st)
    }

    pub fn relu(&mut self, m: &[f64], out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_relu(self.raw, m.as_ptr(), m.len(), out.as_mut_ptr()) };
        self.check(st)
    }

    pub fn pool_2d(&mut self, m: &[f64], r: usize, c: usize, padding: i32, pooling: i32, out: &mut [f64]) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_pool_2d(self.raw, m.as_ptr(), 1, r as c_int, c as c_int, padding, pooling, out.as_mut_ptr()) };
        self.check(st)
    }

    // ---- RCN::train's data flow with both sets resident in HBM (rcn.rs:126-167) ----
    /// `load_data` after the image decode (rcn.rs:399-414) into slot 0 (training) / 1 (testing); returns the set's (mean, sd),
    /// which is also the context's `scale_set` from now on (rcn.rs:249-250).
    pub fn load_data(&mut self, slot: i32, pixels: &[u8], labels: &[i32]) -> Result<(f64, f64), HipError> {
        let (mut mean, mut sd) = (0.0, 0.0);
        let st = unsafe { rcn_hip_load_data(self.raw, slot, pixels.as_ptr(), labels.as_ptr(), labels.len(), &mut mean, &mut sd) };
        self.check(st).map(|_| (mean, sd))
    }

    /// One pass of rcn.rs:146-149 over a loaded slot: `order` is the shuffled index list (`training_set.shuffle`), the
    /// batches are `order.chunks_exact(batch_size)`.
    pub fn train_set_epoch(&mut self, slot: i32, order: &[i32], batch_size: usize, eta: f64) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_train_set_epoch(self.raw, slot, order.as_ptr(), 0, batch_size, eta, std::ptr::null_mut()) };
        self.check(st)
    }

    /// The accuracy count of rcn.rs:152-157 over a loaded slot.
    pub fn evaluate_set(&mut self, slot: i32) -> Result<usize, HipError> {
        let mut acc = 0i64;
        let st = unsafe { rcn_hip_evaluate_set(self.raw, slot, &mut acc) };
        self.check(st).map(|_| acc as usize)
    }

    /// Does `train_set_epoch` at this batch size run on the resident one-XCD kernel (f32 and f64 contexts, batches of 1..256)?
    pub fn train_epoch_resident(&mut self, batch_size: usize) -> bool {
        unsafe { rcn_hip_train_epoch_resident(self.raw, batch_size) != 0 }
    }

    /// How often this context stepped down from the resident kernel to the two-kernel pipeline by itself (a co-tenant on the device).
    pub fn fallbacks_taken(&self) -> i32 {
        unsafe { rcn_hip_fallbacks_taken(self.raw) }
    }

    /// The record of the newest bounded wait of the resident kernel that expired in this context (kept after the library healed it):
    /// which wait, which worker, at which step of which launch, and who was still missing.
    pub fn last_timeout(&self) -> Option<TimeoutRecord> {
        let mut w = [0u32; 16];
        let n = unsafe { rcn_hip_last_timeout(self.raw, w.as_mut_ptr(), w.len()) };
        if n < 13 {
            return None;
        }
        let text = unsafe { CStr::from_ptr(rcn_hip_last_timeout_text(self.raw)).to_string_lossy().into_owned() };
        Some(TimeoutRecord {
            site: w[0], worker: w[1], step: w[2] as i32, launch: w[3], missing: (w[4] as u64) | ((w[5] as u64) << 32), tag: w[6], xcc: w[7],
            rank: w[8], world: w[9], workers: w[11], code: w[12], text,
        })
    }

    /// `RCN::classify` minus the image decode (rcn.rs:84-97): features, standardise with `scale_set`, forward, last arg-max.
    pub fn classify_image(&mut self, pixels: &[u8]) -> Result<usize, HipError> {
        let mut cls = 0i32;
        let st = unsafe { rcn_hip_classify_images(self.raw, pixels.as_ptr(), 1, &mut cls) };
        self.check(st).map(|_| cls as usize)
    }

    /// `load_weights_and_bias` (rcn.rs:425-457); seed 0 = non-deterministic like `thread_rng`.
    pub fn init_params(&mut self, seed: u64) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_init_params(self.raw, seed) };
        self.check(st)
    }

    pub fn set_scale(&mut self, mean: f64, sd: f64) -> Result<(), HipError> {
        let st = unsafe { rcn_hip_set_scale(self.raw, mean, sd) };
        self.check(st)
    }

    pub fn scale(&self) -> (f64, f64) {
        let (mut mean, mut sd) = (0.0, 0.0);
        unsafe { rcn_hip_get_scale(self.raw, &mut mean, &mut sd) };
        (mean, sd)
    }

    pub fn num_layers(&self) -> usize {
        unsafe { rcn_hip_num_layers(self.raw) as usize }
    }

    /// (rows, cols) of `Weights` l: rows = outputs, cols = inputs (rcn.rs:502).
    pub fn layer_dims(&self, layer: usize) -> (usize, usize) {
        let (mut r, mut c) = (0i32, 0i32);
        unsafe { rcn_hip_layer_dims(self.raw, layer as c_int, &mut r, &mut c) };
        (r as usize, c as usize)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { rcn_hip_destroy(self.raw) }
    }
}
