// rcn_hip_api_operators.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): C ABI: the Convolve2D / Pool2D operator entry points.
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
