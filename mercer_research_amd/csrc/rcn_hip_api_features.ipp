// rcn_hip_api_features.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): C ABI: feature pipeline.
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
