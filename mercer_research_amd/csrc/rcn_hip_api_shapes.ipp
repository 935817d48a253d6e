// rcn_hip_api_shapes.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): shape logic shared by the operator API and the feature-stack validation; host <-> device dtype conversion.
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
