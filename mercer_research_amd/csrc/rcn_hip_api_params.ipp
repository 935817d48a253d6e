// rcn_hip_api_params.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): C ABI: parameters.
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
