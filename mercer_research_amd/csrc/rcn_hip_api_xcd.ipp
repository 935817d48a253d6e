// rcn_hip_api_xcd.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): host side of the resident one-XCD epoch kernel (dense_xcd.hpp): workspace, placement probe, launches, self-healing step-down.
// ---- the resident one-XCD epoch kernel (dense_xcd.hpp) ------------------------------------------------------------------
constexpr size_t kXcdProbeLds = xcd_lds_floats(256) * sizeof(float);

// Are the 32 blocks with blockIdx.x % 8 == 0 of a 256-block launch with this LDS footprint on ONE XCD, and every other block
// elsewhere?  Asked once per context, synchronously, before the resident kernel is ever selected (the kernel checks again itself).
int xcd_probe(rcn_hip_ctx* c) {
    if (c->xcd_probe != 0) return RCN_HIP_OK;
    c->xcd_probe = -1;
    const size_t lds = kXcdProbeLds;
    RCN_TRY(set_dyn_lds(c, k_xcd_probe, lds));
    DevBuf out;
    HIP_TRY(c, out.ensure(8 * kXcdWorkers * sizeof(unsigned)));
    std::vector<unsigned> host(8 * kXcdWorkers);
    int good = 0;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_TRY(c, hipMemsetAsync(out.p, 0, host.size() * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(k_xcd_probe, dim3(8 * kXcdWorkers), dim3(kXcdThreads), lds, c->stream, (unsigned*)out.p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(host.data(), out.p, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bool ok = true;
        for (size_t b = 0; b < host.size(); ++b) {                       // block b sits on the XCD of block b % 8, and those eight differ
            if (!(host[b] & 0x100u)) ok = false;
            for (size_t k = 0; k < 8; ++k)
                if ((b % 8 == k) != ((host[b] & 0xfu) == (host[k] & 0xfu))) ok = false;
        }
        good += ok ? 1 : 0;
    }
    out.release();
    if (good == 3) c->xcd_probe = 1;
    return RCN_HIP_OK;
}

template <typename T> static XcdBufsT<T> xcd_bufs(rcn_hip_ctx* c, size_t BT);
// What the first expired wait of the resident kernel recorded (dense_xcd.hpp: xcd_raise), as text for rcn_hip_last_error; with_tables
// (the stream is drained): also the placement and flag tables of the workspace as the failed launch left them.
static const char* xcd_site_name(unsigned site) {
    switch (site) {
    case kXcdSitePlacement: return "placement vote";
    case kXcdSiteTailFlag: return "tail-tile flag (flagT), awaited by a sample group";
    case kXcdSiteSlabFlag: return "slab flag (flagA), awaited by a sample group";
    case kXcdSiteDeltaFlag: return "delta flag (flagB), awaited by a feature worker / tail tile";
    case kXcdSitePushOwner: return "pushed reduce-scatter: the owner waiting for the other ranks' partial sums";
    case kXcdSitePushMember: return "pushed all-gather: a rank waiting for the owner's totals";
    case kXcdSitePushTail: return "pushed all-to-all of a tail parameter";
    case kXcdSitePushCost: return "pushed all-to-all of the cost";
    case kXcdSiteClosing: return "closing round";
    default: return "unknown site";
    }
}
static void xcd_capture_record(rcn_hip_ctx* c) {
    if (!c->xerr_host || c->xerr_host[0] == 0) return;
    for (int i = 0; i < kXcdRecWords; ++i) c->xrec[i] = c->xerr_host[4 + i];
    c->xrec[kXcdRecWords] = c->xerr_host[0];
    c->xrec_valid = true;
}
static std::string hex_mask(unsigned long long m) {
    char b[32];
    std::snprintf(b, sizeof b, "0x%llx", m);
    return b;
}
std::string xcd_describe(rcn_hip_ctx* c, bool with_tables) {
    if (!c->xerr_host || c->xerr_host[0] == 0) return "";
    xcd_capture_record(c);
    const unsigned* r = c->xrec;
    const unsigned site = r[0];
    const bool push_site = r[0] >= kXcdSitePushOwner && r[0] <= kXcdSitePushCost;      // (push sites: the high half carries the parameter index)
    const unsigned long long missing = (unsigned long long)r[4] | (push_site ? 0ull : (unsigned long long)r[5] << 32);
    std::string s = " [first expired wait: site " + std::to_string(site) + " (" + xcd_site_name(site) + "), worker " + std::to_string(r[1]) + " of " + std::to_string(r[11]) +
                    " on XCC " + std::to_string(r[7]) + ", step " + std::to_string((int)r[2]) + " of launch " + std::to_string(r[3]) + ", rank " + std::to_string(r[8]) + " of " +
                    std::to_string(r[9]) + ", worker blocks = blockIdx % 8 == " + std::to_string(r[10]);
    if (site == kXcdSiteClosing) s += ", arrivals seen " + std::to_string(missing) + " of " + std::to_string(r[11]);
    else if (site >= kXcdSitePushOwner && site <= kXcdSitePushCost) {
        s += ", ranks still missing " + hex_mask(missing & 0xffffffffull) + ", exchange step " + std::to_string(r[6]) + ", parameter index " + std::to_string(r[5]);
        // what THIS rank's memory holds at the awaited words now (read by the host, behind the drained stream): a word carrying the awaited
        // step means the peer's store arrived and the poll did not see it; an older step means it never arrived
        if (with_tables && c->p2p.local_buf && c->p2p.stride) {
            const size_t stride = c->p2p.stride, idx0 = r[5];
            const unsigned par = r[6] & 1u;
            const char* base = (const char*)c->p2p.local_buf + c->p2p.push_off;
            auto words = [&](size_t row, int n) {
                unsigned long long wv[4] = {0, 0, 0, 0};
                std::string t;
                if (idx0 + (size_t)n <= stride && hipMemcpy(wv, base + (row * stride + idx0) * 8, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess)
                    for (int i = 0; i < n; ++i) t += " " + std::to_string((unsigned)(wv[i] >> 32));
                return t;
            };
            const int n = site == kXcdSitePushOwner || site == kXcdSitePushMember ? 4 : 1;
            if (site == kXcdSitePushMember) s += "; steps found in this rank's totals words:" + words((size_t)2 * kP2PMaxWorld + par, n);
            else
                for (int q = 0; q < kP2PMaxWorld; ++q)
                    if ((missing >> q) & 1u) s += "; steps found in this rank's row of rank " + std::to_string(q) + ":" + words((size_t)par * kP2PMaxWorld + (size_t)q, n);
        }
    }
    else s += std::string(site == kXcdSitePlacement ? ", workers absent / elsewhere " : ", producers still behind ") + hex_mask(missing) + ", awaited tag " + std::to_string(r[6]);
    if (with_tables && c->xcdbuf.p && c->xcd_B) {
        // the tables as the failed launch left them (one word per 128-byte line): XCC answers, newest tags per producer, committed ids
        const XcdBufsT<float> xb = xcd_bufs<float>(c, c->xcd_B);       // (the flag region's place depends on the element size)
        const XcdBufsT<double> xb8 = xcd_bufs<double>(c, c->xcd_B);
        const unsigned* base = c->dtype == RCN_HIP_F64 ? xb8.flagA : xb.flagA;
        const size_t words = (size_t)(4 * kXcdWorkers + 8 + 2) * kXcdFlagStride;
        std::vector<unsigned> t(words);
        if (hipMemcpy(t.data(), base, words * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
            auto row = [&](const char* name, size_t first, int n, bool low4) {
                s += std::string("; ") + name + " =";
                for (int i = 0; i < n; ++i) s += " " + std::to_string(low4 ? (t[(first + (size_t)i) * kXcdFlagStride] & 0xfu) : t[(first + (size_t)i) * kXcdFlagStride]);
            };
            const int NW = (int)r[11] > 0 && (int)r[11] <= kXcdWorkers ? (int)r[11] : kXcdWorkers;
            row("xcc", 2 * kXcdWorkers, NW, true);
            row("flagA", 0, NW, false);
            row("flagB", kXcdWorkers, NW, false);
            row("flagD", 3 * kXcdWorkers, NW, false);
            row("flagT", 4 * kXcdWorkers, 8, false);
            s += "; decision word = " + hex_mask(t[(size_t)(4 * kXcdWorkers + 8) * kXcdFlagStride]);
        }
    }
    return s + "]";
}

bool use_xcd(rcn_hip_ctx* c, size_t B) {
    if (!xcd_supported(c->nd, B, c->esz())) return false;
    if (c->dense_path != 0 && c->dense_path != 5) return false;
    if (c->xcd_stepped_down) return false;                              // (rcn_hip_set_dense_path(ctx, 5) arms it again)
    if (c->dense_path == 0 && c->opt.xcd == 0) return false;
    if (c->xcd_probe == 0 && xcd_probe(c) != RCN_HIP_OK) return false;
    return c->xcd_probe == 1;
}

int xcd_heal(rcn_hip_ctx* c);

// the data-parallel step runs on the resident kernel: the in-kernel exchange was admitted, one hidden layer, a shard of 32 / 64 / 128 / 256
bool dp_on_xcd(rcn_hip_ctx* c, size_t B) {
    return c->p2p.on && c->p2p.push && c->dtype == RCN_HIP_F32 && c->nd.L == 2 && (size_t)xcd_bt(B) == B && use_xcd(c, B);
}

int ensure_xcd_ws(rcn_hip_ctx* c, size_t B) {
    if (!c->xerr_host) {
        HIP_TRY(c, hipHostMalloc((void**)&c->xerr_host, 256, hipHostMallocMapped));
        std::memset(c->xerr_host, 0, 256);      // [0] the sticky error word, [1] id of the newest launch all of whose workers decided to commit,
                                                // [4 .. 4 + kXcdRecWords) the record of the first wait that expired (dense_xcd.hpp: xcd_raise)
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->xerr_dev, c->xerr_host, 0));
        HIP_TRY(c, hipMalloc((void**)&c->xerrd, 256));
        HIP_TRY(c, hipMemsetAsync(c->xerrd, 0, 256, c->stream));
    }
    if (*c->xerr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, std::string(*c->xerr_host == 2 ? "train_epoch: the resident kernel's workgroups did not share one XCD in an earlier call; nothing was "
                                                                         "updated by it.  rcn_hip_set_dense_path(ctx, 2) selects the two-kernel pipeline"
                                                                       : "train_epoch: a bounded wait inside the resident kernel expired in an earlier call (is the device shared?); "
                                                                         "that call's segment was not applied.  rcn_hip_set_dense_path(ctx, 2) selects the two-kernel pipeline") +
                                            xcd_describe(c, false));
    const size_t BT = (size_t)xcd_bt(B);
    const size_t bytes = xcd_buf_bytes(c->nd, BT, c->esz());
    if (c->xcd_B != BT || c->xcdbuf.cap < bytes) {
        HIP_TRY(c, c->xcdbuf.ensure(bytes));
        HIP_TRY(c, hipMemsetAsync(c->xcdbuf.p, 0, c->xcdbuf.cap, c->stream));      // flags 0: tags start at 1; error word 0
        c->xcd_B = BT;
        c->xcd_tag = 0;
    }
    return RCN_HIP_OK;
}

P2PDesc p2p_desc(const rcn_hip_ctx* c);
static long long p2p_timeout_ticks(const rcn_hip_ctx* c);

// the workspace of one batch instantiation, carved out of c->xcdbuf
template <typename T>
static XcdBufsT<T> xcd_bufs(rcn_hip_ctx* c, size_t BT) {
    const size_t NS = BT / kP2Ts, NA = (size_t)xcd_na(c->nd);
    XcdBufsT<T> xb;
    T* f = (T*)c->xcdbuf.p;
    xb.slab = f; f += NS * NA * kP2Ts * kP2H;
    xb.d1 = f;   f += BT * kP2H;
    xb.a1 = f;   f += BT * kP2H;
    xb.d2 = f;   f += BT * kP2C;
    xb.a2 = f;   f += BT * kP2C;
    xb.d3 = f;   f += BT * kP2C;
    xb.loss = f; f += NS;
    xb.fragimg = f; f += (size_t)kP3BFrag * 64;
    unsigned* u = (unsigned*)(((uintptr_t)f + 127) & ~(uintptr_t)127);
    xb.flagA = u; u += kXcdWorkers * kXcdFlagStride;
    xb.flagB = u; u += kXcdWorkers * kXcdFlagStride;
    xb.xcc = u;   u += kXcdWorkers * kXcdFlagStride;
    xb.flagD = u; u += kXcdWorkers * kXcdFlagStride;
    xb.flagT = u; u += 8 * kXcdFlagStride;
    xb.cw = u;    u += kXcdFlagStride;          // (the 512 spare bytes of xcd_buf_bytes: the decision word's line, then 32 committed ids)
    xb.cdone = u; u += kXcdFlagStride;
    xb.phase = (long long*)u;
    xb.errd = c->xerrd;
    xb.done = c->xerr_dev + 1;
    return xb;
}

// One launch of the instantiation for batch BT.  The kernel asks for at least half a CU's LDS plus one byte so that no two of its
// workers share a CU (option "xcd_exact_lds" = 1: exactly what it uses -- two contexts' resident kernels can then be on one device).
template <typename T, int BT, bool FULL>
int xcd_launch_bt(rcn_hip_ctx* c, const T* xs, const T* ys, size_t B, size_t nb, T scale, T loss_scale, T* loss_dev, bool dp,
                  const int32_t* gperm, bool gather, const XcdBufsT<T>& xb, unsigned tag0, unsigned launch_id) {
    const NetDesc& nd = c->nd;
    size_t lds = xcd_lds_bytes<T, BT>();
    if (!c->opt.xcd_exact_lds && lds < 81 * 1024) lds = 81 * 1024;
    const long long to = c->opt.xcd_timeout_ticks;
    const int xsel = (int)c->opt.xcd_select;
#define RCN_XCD_LAUNCH(KERN, TO, DPARG)                                                                                                                   \
    do {                                                                                                                                                  \
        RCN_TRY(set_dyn_lds(c, KERN, lds));                                                                                                               \
        hipLaunchKernelGGL(KERN, dim3(8 * kXcdWorkers), dim3(kXcdThreads), lds, c->stream, nd, (T*)c->params.p, xs, ys, (int)B, (int)nb,                  \
                           pipe_slices(nd), scale, loss_scale, loss_dev, xb, tag0, c->xerr_dev, TO, DPARG, xsel, (const int*)gperm, launch_id);           \
    } while (0)
    if (dp) {
        // (the data-parallel form exists for whole instantiation sizes: a shard of 32 / 64 / 128 / 256 samples per rank, f32)
        if constexpr (FULL && sizeof(T) == 4) {
            bool clocked = false;
            if constexpr (BT == 256 || BT == 128) {
                if (c->opt.xcd_dp_phase) {          // diagnostic: the same launch with per-worker phase clocks (rcn_hip_dp_phase_us)
                    HIP_TRY(c, hipMemsetAsync(xb.phase, 0, (size_t)kXcdWorkers * 4 * sizeof(long long), c->stream));
                    RCN_XCD_LAUNCH((k_xcd_epoch<float, BT, true, true, false, false, true>), to + 2 * p2p_timeout_ticks(c),
                                   (XcdDpOn{PushDesc{p2p_desc(c), c->p2p.stride, c->p2p.push_off}, c->p2p.seq + 1, p2p_timeout_ticks(c)}));
                    clocked = true;
                }
            }
            if (!clocked)
            RCN_XCD_LAUNCH((k_xcd_epoch<float, BT, true, true>), to + 2 * p2p_timeout_ticks(c),
                           (XcdDpOn{PushDesc{p2p_desc(c), c->p2p.stride, c->p2p.push_off}, c->p2p.seq + 1, p2p_timeout_ticks(c)}));
            c->p2p.seq += (unsigned)nb;
            c->xcd_dp_used = true;
        } else return fail(c, RCN_HIP_ERR_UNSUPPORTED, "the resident kernel's data-parallel form needs an f32 context and a shard of 32, 64, 128 or 256 samples");
    } else if (nd.L == 3) {
        RCN_XCD_LAUNCH((k_xcd_epoch<T, BT, FULL, false, true>), to, XcdDpOff{});
    } else if (gather) {
        if constexpr (BT == 256 && FULL && sizeof(T) == 4) RCN_XCD_LAUNCH((k_xcd_epoch<float, 256, true, false, false, true>), to, XcdDpOff{});
        else return fail(c, RCN_HIP_ERR_UNSUPPORTED, "the gather form of the resident kernel exists for batch 256, f32 only");
    } else {
        RCN_XCD_LAUNCH((k_xcd_epoch<T, BT, FULL, false>), to, XcdDpOff{});
    }
#undef RCN_XCD_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return RCN_HIP_OK;
}

// nb consecutive steps over batches whose packed images are contiguous from xs / ys (one segment of the epoch image).
// dp: the data-parallel step -- B is this rank's shard, the update uses the global batch length, gradients meet inside the kernel.
// gather: xs / ys are the caller's X[rows][F] / Y[rows][C] as stored and gperm the order of their rows (NULL: stored order) -- the kernel
// fetches every batch's rows itself, a step ahead; else they are the packed epoch image (k_pack_epoch) and gperm is unused.
// *id_out (nullable): the launch's id, which the kernel reports in `done` once all of its workers have finished.
template <typename T>
int enqueue_xcd_steps(rcn_hip_ctx* c, const T* xs, const T* ys, size_t B, size_t nb, double eta, T* loss_dev, bool dp = false,
                      const int32_t* gperm = nullptr, bool gather = false, unsigned* id_out = nullptr) {
    const int BT = xcd_bt(B);
    const XcdBufsT<T> xb = xcd_bufs<T>(c, (size_t)BT);
    // (the tail parameters as the sample groups' operand fragments -- xb.fragimg -- are written by the kernel's own tail tiles: at its
    // start from the parameter vector, then after every update; its pads are the zeros the buffer was created with)
    const unsigned tag0 = c->xcd_tag + 1;
    // launch ids are unique in the process (the placement vote of a launch accepts only answers that carry its id: dense_xcd.hpp) and
    // increase along a context's stream (the redo journal compares them with the id the kernel reports complete)
    static std::atomic<unsigned> g_launch{0};
    unsigned id = (g_launch.fetch_add(1) + 1u) & 0x3fffffffu;
    if (id == 0) id = (g_launch.fetch_add(1) + 1u) & 0x3fffffffu;
    c->xcd_launch_id = id;
    c->xcd_launches += 1;
    // (test hooks: bit 31 -- worker 1 of this launch never becomes resident; bit 30 -- it reaches the closing round after the others gave up)
    const bool faulty = c->opt.xcd_fault_launch != 0 && (long long)c->xcd_launches == c->opt.xcd_fault_launch;
    const unsigned id_arg = id | (faulty ? (c->opt.xcd_fault_mode == 1 ? 0x40000000u : 0x80000000u) : 0u);
    const double Bg = (double)B * (dp ? (double)c->dp_world : 1.0);          // the global batch.len() of rcn.rs:214
    const T scale = (T)(eta / Bg), loss_scale = (T)(1.0 / (2.0 * Bg));
    int st = RCN_HIP_ERR_UNSUPPORTED;
    const bool full = (size_t)BT == B;
#define RCN_XCD_BT(N)                                                                                                                        \
    st = full ? xcd_launch_bt<T, N, true>(c, xs, ys, B, nb, scale, loss_scale, loss_dev, dp, gperm, gather, xb, tag0, id_arg)               \
              : xcd_launch_bt<T, N, false>(c, xs, ys, B, nb, scale, loss_scale, loss_dev, dp, gperm, gather, xb, tag0, id_arg)
    switch (BT) {
    case 32:  RCN_XCD_BT(32); break;
    case 64:  RCN_XCD_BT(64); break;
    case 128: RCN_XCD_BT(128); break;
    default:  RCN_XCD_BT(256); break;
    }
#undef RCN_XCD_BT
    RCN_TRY(st);
    c->xcd_tag += (unsigned)nb;
    if (id_out) *id_out = id;
    return RCN_HIP_OK;
}

// The gather form of the resident kernel (rows fetched by the workers themselves; f32 feature vectors whose rows are whole 16-byte
// chunks) is OFF unless RCN_HIP_XCD_GATHER=1.  Measured on MI355X (bench workload): the kernel's step takes 7.4 us this way against
// 6.45 us on the packed image -- a wave's loads retire in order, so every wait for a slab or flag word that follows the prefetch also
// waits for 256 scattered 128-byte reads, where the packed image gives it one 32 KB run -- and k_pack_epoch's gather costs only
// 0.39 us per step amortised: 7.39 vs 6.80 us per step in the bench's steady state.
constexpr size_t kXcdMaxStepsPerLaunch = 1u << 20;
static bool xcd_gather(const rcn_hip_ctx* c) {
    return c->opt.xcd_gather != 0 && c->nd.dims[0] % 4 == 0 && c->nd.L == 2;
}

// the newest source of the index rows `perm` points into (a shuffle or an upload the library performed), or none
static rcn_hip_ctx::PermSource perm_source_of(const rcn_hip_ctx* c, const int32_t* perm) {
    if (perm)
        for (auto it = c->perm_sources.rbegin(); it != c->perm_sources.rend(); ++it)
            if (perm >= it->buf && perm < it->buf + it->n * it->passes) return *it;
    return rcn_hip_ctx::PermSource{};
}
static void note_perm_source(rcn_hip_ctx* c, rcn_hip_ctx::PermSource&& src) {
    if (c->replaying) return;
    for (auto& e : c->perm_sources)
        if (e.buf == src.buf) { e = std::move(src); return; }
    if (c->perm_sources.size() >= 8) c->perm_sources.erase(c->perm_sources.begin());
    c->perm_sources.push_back(std::move(src));
}

// a whole call on the resident kernel: batches [j0, j0 + nb) of the call, packed segment by segment (or already packed)
template <typename T>
int enqueue_xcd_epoch_t(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool from_images,
                        bool prepacked, size_t j0, size_t pre_seg, bool dp) {
    const size_t G = pipe_slices(c->nd), Cc = c->nd.dims[c->nd.L];
    // single-GPU calls are journalled until their launches have been seen complete (xcd_verify): what a failed launch did not apply is
    // re-run from here on the two-kernel pipeline
    rcn_hip_ctx::RedoRec* rec = nullptr;
    if (!dp && !c->replaying && c->opt.xcd_auto_fallback) {
        // (launches the kernel has already reported complete need no record any more: xerr_host[1] is pinned memory, read for free)
        if (c->xerr_host && c->xerr_host[0] == 0) {
            const unsigned done = c->xerr_host[1];
            size_t keep = 0;
            while (keep < c->redo.size() && (c->redo[keep].launches.empty() || (int)(c->redo[keep].launches.back().id - done) <= 0)) ++keep;
            if (keep) c->redo.erase(c->redo.begin(), c->redo.begin() + keep);
        }
        c->redo.emplace_back();
        rec = &c->redo.back();
        rec->kind = prepacked ? 1 : 0;
        rec->X = X; rec->Y = Y; rec->perm = perm; rec->B = B; rec->nb = nb; rec->j0 = j0; rec->eta = eta; rec->loss_dev = loss_dev; rec->from_images = from_images;
        if (!prepacked) rec->src = perm_source_of(c, perm);
        else rec->begin = c->last_begin;
        rec->caller_rows = prepacked ? (rec->begin.perm != nullptr && rec->begin.src.kind == 0) : (perm != nullptr && rec->src.kind == 0);
    }
    auto note = [&](unsigned id, size_t k0, size_t n) { if (rec) rec->launches.push_back({id, k0, n}); };
    if (sizeof(T) == 4 && !prepacked && !from_images && !dp && xcd_gather(c)) {
        c->xg.X = (const float*)X; c->xg.Y = (const float*)Y; c->xg.perm = perm; c->xg.B = B; c->xg.nb = nb;
        // feature vectors as stored: no packed image at all -- ONE launch walks the whole call, every worker gathering its 128 bytes of
        // each row of the batch after next while it works on the current one (the bytes k_pack_epoch would read, write and hand back)
        for (size_t k = 0; k < nb;) {
            const size_t n = nb - k < kXcdMaxStepsPerLaunch ? nb - k : kXcdMaxStepsPerLaunch;
            unsigned id = 0;
            RCN_TRY(enqueue_xcd_steps<T>(c, (const T*)X, (const T*)Y, B, n, eta, loss_dev ? (T*)loss_dev + k : nullptr, dp,
                                         perm ? perm + k * B : nullptr, true, &id));
            note(id, k, n);
            if (!perm) { X = (const T*)X + n * B * c->nd.dims[0]; Y = (const T*)Y + n * B * Cc; }
            k += n;
        }
        return RCN_HIP_OK;
    }
    const size_t seg = prepacked ? pre_seg : (nb <= pack_segment(c, B) ? nb : pack_segment(c, B));
    auto slot = [&](size_t j) { return ((j / seg) % 2) * seg + j % seg; };
    for (size_t j = prepacked ? j0 : 0, end = j + nb, k = 0; j < end;) {
        const size_t in_seg = seg - j % seg, n = end - j < in_seg ? end - j : in_seg;       // up to the end of this segment of the image
        if (!prepacked) {
            const int half = (int)((j / seg) % 2);
            RCN_TRY(from_images ? launch_feat_pack<T>(c, (const uint8_t*)X, Y, perm, B, j, n, half, seg) : launch_pack<T>(c, X, Y, perm, B, j, n, half, seg));
        }
        const T* xs = (const T*)c->xpack.p + slot(j) * G * B * 16;
        const T* ys = (const T*)c->ypack.p + slot(j) * B * Cc;
        unsigned id = 0;
        RCN_TRY(enqueue_xcd_steps<T>(c, xs, ys, B, n, eta, loss_dev ? (T*)loss_dev + k : nullptr, dp, nullptr, false, &id));
        note(id, k, n);
        j += n; k += n;
    }
    return RCN_HIP_OK;                      // (the sticky error word lives in pinned host memory: current once the stream has drained)
}
int enqueue_xcd_epoch(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool from_images,
                      bool prepacked, size_t j0, size_t pre_seg, bool dp = false) {
    return c->dtype == RCN_HIP_F64 ? enqueue_xcd_epoch_t<double>(c, X, Y, perm, B, nb, eta, loss_dev, from_images, prepacked, j0, pre_seg, dp)
                                   : enqueue_xcd_epoch_t<float>(c, X, Y, perm, B, nb, eta, loss_dev, from_images, prepacked, j0, pre_seg, dp);
}

// one train_batch (rcn.rs:176-223) on device-resident data; idx selects the batch's rows (or NULL)
int enqueue_train_step(rcn_hip_ctx* c, const void* x, const void* y, const int32_t* idx, size_t B, double eta, void* loss_dev) {
    const double scale = eta / (double)B;                       // rcn.rs:214: eta / batch.len() as f64
    const double loss_scale = 1.0 / (2.0 * (double)B);
    if (use_pipe(c, B)) {
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, 1));
        return c->dtype == RCN_HIP_F64 ? enqueue_pipe_steps<double>(c, x, y, idx, B, 1, eta, loss_dev)
                                       : enqueue_pipe_steps<float>(c, x, y, idx, B, 1, eta, loss_dev);
    }
    if (c->dtype == RCN_HIP_F64) {
        RCN_TRY(launch_fwd<double>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<double>(c, true, x, idx, B, scale, nullptr, loss_dev, loss_scale));
    } else {
        RCN_TRY(launch_fwd<float>(c, true, x, y, idx, B, nullptr));
        RCN_TRY(launch_wgrad<float>(c, true, x, idx, B, scale, nullptr, loss_dev, loss_scale));
    }
    return RCN_HIP_OK;
}

int check_ctx(const rcn_hip_ctx* c) { return c ? RCN_HIP_OK : RCN_HIP_ERR_INVALID_ARG; }

// the resident kernel's sticky error word, both copies (a recovery action of the caller, the heal below, the end of a data-parallel group)
int xcd_clear_error(rcn_hip_ctx* c) {
    if (c->xerr_host) c->xerr_host[0] = 0;
    if (c->xerrd) HIP_TRY(c, hipMemsetAsync(c->xerrd, 0, 4, c->stream));
    return RCN_HIP_OK;
}

// Self-healing step-down of the single-GPU resident kernel.  Precondition: the stream is drained and the sticky word is set (a bounded
// wait expired -- typically a co-tenant holds CUs of the XCD, so the 32 workers were never resident together -- or the workers were
// not on one XCD).  Nothing a failed launch computed reached memory and every launch enqueued behind it left at once, so the
// parameter vector is the state after launch xerr_host[1]: the context steps down to the two-kernel pipeline for good, and every
// step the journal holds beyond that launch is re-run there, from the arguments its call was given (index rows the library itself
// shuffled or uploaded are re-created first; anything else the calls read is taken to be unchanged -- the contract of an
// asynchronous call whose inputs must stay untouched until a synchronise).  Reported through rcn_hip_fallbacks_taken, not as an error.
static int redo_perm(rcn_hip_ctx* c, const rcn_hip_ctx::PermSource& ps) {
    if (ps.kind == 1) return rcn_hip_shuffle_dev(c, ps.buf, ps.n, ps.passes, ps.seed);
    if (ps.kind == 2) {
        HIP_TRY(c, hipMemcpyAsync(ps.buf, ps.host.data(), ps.host.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return RCN_HIP_OK;
}
static int redo_begin(rcn_hip_ctx* c, const rcn_hip_ctx::BeginRec& b) {
    RCN_TRY(redo_perm(c, b.src));
    return b.from_images ? rcn_hip_epoch_begin_images_dev(c, (const uint8_t*)b.X, b.Y, b.perm, b.B, b.nb) : rcn_hip_epoch_begin_dev(c, b.X, b.Y, b.perm, b.B, b.nb);
}
int xcd_heal(rcn_hip_ctx* c) {
    const unsigned code = c->xerr_host[0], done = c->xerr_host[1];
    c->xlast = xcd_describe(c, true);            // (kept: rcn_hip_last_timeout / the step-down's reason stay readable after the heal)
    // cross-check of `done` (decided on one word by all workers) with the ids the workers stored behind their write-backs: a worker that
    // wrote back a launch newer than `done` would mean a torn parameter vector -- reported, never replayed over
    if (c->xcdbuf.p && c->xcd_B) {
        unsigned cd[kXcdWorkers];
        const unsigned* src = c->dtype == RCN_HIP_F64 ? xcd_bufs<double>(c, c->xcd_B).cdone : xcd_bufs<float>(c, c->xcd_B).cdone;
        HIP_TRY(c, hipMemcpy(cd, src, sizeof cd, hipMemcpyDeviceToHost));
        const int NW = xcd_workers(c->nd, (int)c->xcd_B);
        for (int w = 0; w < NW; ++w)
            if (cd[w] != 0 && (int)(cd[w] - done) > 0)
                return fail(c, RCN_HIP_ERR_HIP, "the resident kernel failed and worker " + std::to_string(w) + " wrote back launch " + std::to_string(cd[w]) +
                                                    " although the launch was not committed (newest committed: " + std::to_string(done) + "): the parameters are torn" + c->xlast);
    }
    // Index rows the caller wrote may legally have been overwritten in stream order since the call (a fresh torch.randperm per epoch):
    // steps that would be re-run on them are NOT re-run silently -- the error stays, as with xcd_auto_fallback = 0
    if (!c->opt.xcd_replay_caller_rows)
        for (const auto& r : c->redo) {
            bool open_steps = false;
            for (const auto& l : r.launches)
                if ((int)(l.id - done) > 0) open_steps = true;
            if (open_steps && r.caller_rows)
                return fail(c, RCN_HIP_ERR_HIP, "a bounded wait inside the resident one-XCD kernel expired and the steps it had not applied read index rows the caller wrote, "
                                                "which the library cannot know to be unchanged: not re-run (rows from rcn_hip_shuffle_dev or an upload of the library are; "
                                                "option xcd_replay_caller_rows = 1 re-runs these too)" + c->xlast);
        }
    RCN_TRY(xcd_clear_error(c));
    c->xcd_stepped_down = true;
    c->fallbacks_taken += 1;
    std::vector<rcn_hip_ctx::RedoRec> redo;
    redo.swap(c->redo);
    c->replaying = true;
    struct Guard { rcn_hip_ctx* c; ~Guard() { c->replaying = false; } } guard{c};
    const size_t F = (size_t)c->nd.dims[0], Cc = (size_t)c->nd.dims[c->nd.L], es = c->esz(), HW = (size_t)c->fd.H * c->fd.W;
    const bool image_was_live = c->epoch_nb != 0;
    bool image_touched = false;
    for (const auto& r : redo) {
        size_t k0 = r.nb;                        // the first step of this call no complete launch covered
        for (const auto& l : r.launches)
            if ((int)(l.id - done) > 0) { k0 = l.k0; break; }
        if (k0 >= r.nb) continue;
        void* loss = r.loss_dev ? (char*)r.loss_dev + k0 * es : nullptr;
        if (r.kind == 0) {
            RCN_TRY(redo_perm(c, r.src));
            const int32_t* pm = r.perm ? r.perm + k0 * r.B : nullptr;
            const void* Y = r.perm ? r.Y : (const void*)((const char*)r.Y + k0 * r.B * Cc * es);
            if (r.from_images) {
                const uint8_t* X = r.perm ? (const uint8_t*)r.X : (const uint8_t*)r.X + k0 * r.B * HW;
                RCN_TRY(rcn_hip_train_epoch_images_dev(c, X, Y, pm, r.B, r.nb - k0, r.eta, loss));
            } else {
                const void* X = r.perm ? r.X : (const void*)((const char*)r.X + k0 * r.B * F * es);
                RCN_TRY(rcn_hip_train_epoch_dev(c, X, Y, pm, r.B, r.nb - k0, r.eta, loss));
            }
            image_touched = true;
        } else {
            if (!r.begin.valid) return fail(c, RCN_HIP_ERR_HIP, "the resident kernel failed and the epoch image its steps ran on cannot be laid out again; what it had not applied is lost");
            RCN_TRY(redo_begin(c, r.begin));
            RCN_TRY(rcn_hip_epoch_steps_dev(c, r.j0 + k0, r.nb - k0, r.eta, loss));
            image_touched = true;
        }
    }
    // the index buffers and the epoch image end as the caller's newest calls left them
    for (const auto& ps : c->perm_sources) RCN_TRY(redo_perm(c, ps));
    if (image_touched && image_was_live && c->last_begin.valid) RCN_TRY(redo_begin(c, c->last_begin));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)code;
    return RCN_HIP_OK;
}

// The in-kernel waits (peer exchange, resident / one-launch step kernels) are bounded: a wait that expires sets a sticky
// device word, every later kernel of that family drains, and the updates of the call are only partly applied.  The word is
// copied back asynchronously at the end of each epoch call; after a stream synchronise it is current.  Every entry point
// that tells the caller "the work is complete / here are the parameters" calls this behind its synchronise.
int sticky_errors(rcn_hip_ctx* c) {
    if (c->p2p.err_host && c->p2p.err_dev) {
        if (*c->p2p.err_host != 0)
            return fail(c, RCN_HIP_ERR_HIP, "data-parallel exchange: rank " + std::to_string(c->dp_rank) + " timed out waiting for peer data (sticky word " +
                                                std::to_string(*c->p2p.err_host) + "); the last call's updates are incomplete and the replicas are no longer in step");
    }
    if (c->perr_host && *c->perr_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "a bounded wait inside the resident / one-launch step kernel expired; the last call's updates are incomplete");
    if (c->xerr_host && *c->xerr_host != 0) {
        if (!c->xcd_dp_used && c->opt.xcd_auto_fallback && !c->replaying) return xcd_heal(c);
        c->xlast = xcd_describe(c, true);
        return fail(c, RCN_HIP_ERR_HIP, std::string(*c->xerr_host == 2 ? "the resident one-XCD kernel found its workgroups on different XCDs; what it had not applied is lost"
                                                                       : (c->xcd_dp_used ? "a bounded wait inside the resident one-XCD kernel expired in a data-parallel step; the replicas are no longer in step"
                                                                                         : "a bounded wait inside the resident one-XCD kernel expired; what it had not applied is lost")) + c->xlast);
    }
    if (c->xerr_host) c->redo.clear();          // the stream is drained and nothing failed: every journalled launch is complete
    return RCN_HIP_OK;
}

// a failure of the resident kernel the host can already see: healed (or reported) before anything else is enqueued behind it
int xcd_entry_check(rcn_hip_ctx* c) {
    if (c->xerr_host && c->xerr_host[0] != 0 && !c->replaying) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return sticky_errors(c);
    }
    return RCN_HIP_OK;
}

int need_dense(rcn_hip_ctx* c) {
    if (!c->dense_err.empty()) return fail(c, RCN_HIP_ERR_SHAPE, c->dense_err);
    return RCN_HIP_OK;
}

int need_params(rcn_hip_ctx* c) {
    RCN_TRY(need_dense(c));
    if (!c->params_set) return fail(c, RCN_HIP_ERR_STATE, "parameters not set: call rcn_hip_set_params for every layer or rcn_hip_init_params first");
    return RCN_HIP_OK;
}
