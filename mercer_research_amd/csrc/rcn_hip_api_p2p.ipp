// rcn_hip_api_p2p.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): peer exchange plumbing (dp_p2p.hpp, dp_push.hpp): IPC hand-off of the buffers, graph teardown.
// ---- peer-read all-reduce plumbing (dp_p2p.hpp) ------------------------------------------------------------------
// Uncached (fine-grained) device memory is never handed back to the runtime while the process lives: it is parked here and reused by the
// next data-parallel group.  Measured in round 3: after a 3.9 MB hipDeviceMallocUncached block had been hipFree'd, the next context's
// ORDINARY hipMalloc allocations came back on that memory still behaving uncached -- plain stores no longer stayed in the XCD's L2 and
// the resident kernel's hand-offs (payload, drain, flag: dense_xcd.hpp) were read stale: deterministic wrong costs in the first step
// of a context created right after a data-parallel one, gone with RCN_HIP_DP_CACHED_BUF=1 and gone with this cache.
// Bounded (round 4): a request takes the SMALLEST parked block of its device that is large enough (not only an exact match), so a process
// that builds groups for nets of different sizes re-uses what it parked; what stays parked is the price of never handing such memory
// back -- at most one block per distinct size class a process ever used, a few MB for the nets this library trains (the default net's
// exported buffer: 3.9 MB), reported by parked_bytes() for whoever wants to watch it.
struct UncachedCache {
    std::mutex mu;
    std::vector<std::tuple<int, size_t, void*>> free_list;        // (device, capacity in bytes, pointer)
    // *cap_out: the capacity of the block handed out (>= bytes) -- what park() must be given back
    hipError_t alloc(int device, size_t bytes, void** out, size_t* cap_out) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); ++i)
                if (std::get<0>(free_list[i]) == device && std::get<1>(free_list[i]) >= bytes &&
                    (best == free_list.size() || std::get<1>(free_list[i]) < std::get<1>(free_list[best])))
                    best = i;
            if (best != free_list.size() && std::get<1>(free_list[best]) <= 4 * bytes + (1u << 20)) {      // (not a 100 MB block for a 4 KB request)
                *out = std::get<2>(free_list[best]);
                *cap_out = std::get<1>(free_list[best]);
                free_list.erase(free_list.begin() + best);
                return hipSuccess;
            }
        }
        *cap_out = bytes;
        return hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached);
    }
    void park(int device, size_t cap, void* p) {
        std::lock_guard<std::mutex> lk(mu);
        free_list.emplace_back(device, cap, p);
    }
    size_t parked_bytes() {
        std::lock_guard<std::mutex> lk(mu);
        size_t s = 0;
        for (auto& e : free_list) s += std::get<1>(e);
        return s;
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
    if (q.local_flags) uncached_cache().park(c->device, q.flags_cap, q.local_flags);
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
    else HIP_TRY(c, uncached_cache().alloc(c->device, bytes, &q.local_buf, &q.local_bytes));       // (local_bytes: the block's capacity, what is parked again)
    HIP_TRY(c, uncached_cache().alloc(c->device, kP2PFlagBytes, (void**)&q.local_flags, &q.flags_cap));
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
