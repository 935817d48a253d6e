// rcn_hip_api_dense_launch.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): dense launches: sample-tile / pipeline / two-kernel forms, the parked one-launch-per-step and one-kernel-per-segment experiments.
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
