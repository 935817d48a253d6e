// rcn_hip_api_dense.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): C ABI: dense network (forward, train_batch, epochs).
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
