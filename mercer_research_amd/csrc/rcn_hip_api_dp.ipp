// rcn_hip_api_dp.ipp -- part of the ONE translation unit rcn_hip_api.hip (included there, in this order; shares its anonymous namespace and the
// extern "C" block): C ABI: data-parallel training (RCCL, admission of the peer exchange, the in-kernel forms).
// ---------------------------------------------------------------- data-parallel training over RCCL
// The sum over samples of rcn.rs:190-205 is split over ranks: every rank computes the summed gradient of its shard of
// each global batch, ONE ncclAllReduce(sum) of the flat gradient (+ the loss in its last element) combines them over
// xGMI, and every rank applies the identical update W <- W - (eta / B_global) * sum dW (rcn.rs:214,221 with the global
// batch length), so replicas stay bit-identical without a broadcast.  The whole loop is enqueued from here on the
// context's stream -- no host round trip, no Python between steps.
#define NCCL_TRY(ctx, expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess)                                                                               \
            return fail(ctx, RCN_HIP_ERR_HIP, std::string(#expr) + ": " + rcn::Rccl::get().GetErrorString(r_)); \
    } while (0)

// ---- admission of the peer exchange -----------------------------------------------------------------------------------
// Sets up the peer-read all-reduce for a group of ranks, and keeps it only if EVERY rank could map every peer and a known-answer
// exchange came back exact on every rank; then asks the same of the in-kernel (tagged-word) form.  Every rank executes the same
// sequence of collectives whatever fails locally (a failure only lowers its vote), so a rank that cannot use xGMI peer reads makes
// the whole group stay on the previous form instead of deadlocking it.  The collectives come from a transport: RCCL on the
// communicator of rcn_hip_dp_init, or two caller-supplied callbacks (rcn_hip_dp_p2p_admit: any out-of-band channel).
//
// Outcome (identical on every rank):  0 = no peer exchange (the loop uses ncclAllReduce, or the caller's own all-reduce),
// 1 = peer exchange at kernel boundaries, 2 = peer exchange inside the gradient kernel.
//
// Fault injection for the tests, RCN_HIP_DP_FAULT="<stage>:<rank>[,<stage>:<rank>...]": the named rank behaves as if that stage had
// failed locally -- export | attach | kat (known-answer mismatch) | ll (tagged-word self-test mismatch) | llskip (the rank never
// launches its side of the tagged-word self-test, so its peers really time out).
struct P2PTransport {
    std::function<int(const void* mine, void* all, size_t bytes_per_rank)> allgather;     // host buffers, rank order
    std::function<int(int& v)> vote_min;                                                    // v <- min over ranks
};

static bool p2p_fault(const rcn_hip_ctx* c, const char* stage) {
    if (c->dp_fault.empty()) return false;
    const std::string want = std::string(stage) + ":" + std::to_string(c->dp_rank);
    const std::string& all = c->dp_fault;
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t end = all.find(',', pos);
        if (all.substr(pos, end == std::string::npos ? std::string::npos : end - pos) == want) return true;
        if (end == std::string::npos) break;
        pos = end + 1;
    }
    return false;
}

// The vote sequence itself, separated from what the stages do: `ops` is the context's device work (p2p_admission below) or a script
// (rcn_hip_dp_admission_rehearse: the same sequence over the caller's transport without any GPU, which is how the CPU test drives
// the native logic over gloo).  Every stage returns 1 (this rank is fine) or 0; every rank calls the transport the same number of times
// in the same order whatever its own stages returned.
struct AdmissionOps {
    std::function<int(char* handles)> do_export;            // export this rank's buffers -> handles
    std::function<int(const char* all_handles)> attach;      // map every peer
    std::function<int()> known_answer;                       // the kernel-boundary exchange, exact sums, no timeout
    std::function<int()> wants_fused;                        // configuration: may the exchange run inside a step kernel at all?
    std::function<int()> tagged_words;                       // the in-kernel (pull) exchange's known-answer test
    std::function<int()> wants_push;                         // configuration: f32 context?
    std::function<int()> pushed_words;                       // the pushed reduce-scatter + all-gather's known-answer test
    std::function<int()> clear_sticky;                       // after a failed in-kernel stage: 1 if the kernel-boundary form survives
};
struct AdmissionOutcome { bool on = false, fused = false, push = false; };

static int admission_protocol(int world, const AdmissionOps& ops, const P2PTransport& t, AdmissionOutcome& out) {
    char mine[kP2PHandleBytes] = {};
    int ok = ops.do_export(mine);
    std::vector<char> all((size_t)world * kP2PHandleBytes);
    int st = RCN_HIP_OK;
    do {
        if ((st = t.allgather(mine, all.data(), kP2PHandleBytes)) != RCN_HIP_OK) break;
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank export?
        if (!ok) break;
        ok = ops.attach(all.data());
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank map every peer?
        if (!ok) break;
        ok = ops.known_answer();
        if ((st = t.vote_min(ok)) != RCN_HIP_OK) break;          // did every rank see exact sums, without a timeout?
        if (!ok) break;
        out.on = true;
        // second, independent question: may the exchange also run INSIDE the gradient kernel (tagged words, no flags)?  A failed
        // wait here leaves the sticky error word set, which would disable the kernel-boundary protocol too, so it is cleared
        // (after every rank has drained: the vote synchronises) when only this stage failed.
        int okf = ops.wants_fused();
        if ((st = t.vote_min(okf)) != RCN_HIP_OK) break;         // every rank must want it (same configuration everywhere, normally)
        if (!okf) break;
        okf = ops.tagged_words();
        if ((st = t.vote_min(okf)) != RCN_HIP_OK) break;
        if (okf) out.fused = true;
        else {
            // (every rank is here -- the vote above gave all of them the same answer -- so clearing is voted too: a rank that cannot
            // clear its sticky word takes the whole group off the peer exchange, not only itself.  Found by the CPU rehearsal of this
            // sequence, tests/test_dp_gloo.py: rounds 1-2 decided this locally and the ranks could land on different forms.)
            int okc = ops.clear_sticky();
            if ((st = t.vote_min(okc)) != RCN_HIP_OK) break;
            if (!okc) out.on = false;
        }
        if (!out.on || !out.fused) break;
        // third question, asked only of a group that passed everything before it: the pushed reduce-scatter + all-gather the resident
        // kernel's data-parallel form runs (f32 contexts; remote STORES into the peers' memory and local polls, where the two forms
        // above only ever read a peer's memory)
        int okp = ops.wants_push();
        if ((st = t.vote_min(okp)) != RCN_HIP_OK) break;
        if (!okp) break;
        okp = ops.pushed_words();
        if ((st = t.vote_min(okp)) != RCN_HIP_OK) break;
        if (okp) out.push = true;
        else {
            int okc = ops.clear_sticky();
            if ((st = t.vote_min(okc)) != RCN_HIP_OK) break;
            if (!okc) { out.on = false; out.fused = false; }
        }
    } while (0);
    return st;
}

static int p2p_admission(rcn_hip_ctx* c, const P2PTransport& t) {
    const int world = c->dp_world, rank = c->dp_rank;
    AdmissionOps ops;
    ops.do_export = [&](char* h) { return (p2p_export(c, h) == RCN_HIP_OK && !p2p_fault(c, "export")) ? 1 : 0; };
    ops.attach = [&](const char* all) { return (p2p_attach(c, all, rank, world) == RCN_HIP_OK && !p2p_fault(c, "attach")) ? 1 : 0; };
    ops.known_answer = [&]() {
        unsigned bad = 0, err = 0;
        return (p2p_selftest(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "kat")) ? 1 : 0;
    };
    ops.wants_fused = [&]() { return c->opt.dp_fused ? 1 : 0; };
    ops.tagged_words = [&]() {
        unsigned bad = 0, err = 0;
        if (p2p_fault(c, "llskip")) { c->p2p.seq += 16; return 0; }          // this rank stays silent: its peers' waits expire
        return (p2p_selftest_fused(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "ll")) ? 1 : 0;
    };
    ops.wants_push = [&]() { return (c->opt.dp_fused && c->dtype == RCN_HIP_F32) ? 1 : 0; };
    ops.pushed_words = [&]() {
        unsigned bad = 0, err = 0;
        if (p2p_fault(c, "pushskip")) { c->p2p.seq += 16; return 0; }
        return (p2p_selftest_push(c, 16, &bad, &err) == RCN_HIP_OK && bad == 0 && err == 0 && !p2p_fault(c, "push")) ? 1 : 0;
    };
    ops.clear_sticky = [&]() {
        // every rank has drained (the vote synchronised them); clear the sticky word and the pinned mirror of it
        const bool ok = hipMemsetAsync(c->p2p.err_dev, 0, 4, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        if (c->p2p.err_host) *c->p2p.err_host = 0;
        return ok ? 1 : 0;
    };
    AdmissionOutcome out;
    const int st = admission_protocol(world, ops, t, out);
    c->p2p.on = out.on; c->p2p.fused = out.on && out.fused; c->p2p.push = out.on && out.push;
    if (!c->p2p.on) { const int rk = c->dp_rank, w = c->dp_world; p2p_release(c); c->dp_rank = rk; c->dp_world = w; }
    c->err.clear();                                              // a failed attempt is not an error: the loop runs on the previous form
    return st;
}

static int p2p_bootstrap_over_rccl(rcn_hip_ctx* c) {
    rcn::Rccl& r = rcn::Rccl::get();
    const int world = c->dp_world;
    DevBuf xch;
    HIP_TRY(c, xch.ensure((size_t)(world + 1) * kP2PHandleBytes + 64));
    char* d_all = (char*)xch.p;
    char* d_mine = d_all + (size_t)world * kP2PHandleBytes;
    int* d_vote = (int*)(d_mine + kP2PHandleBytes);
    P2PTransport t;
    t.vote_min = [&](int& v) -> int {
        HIP_TRY(c, hipMemcpyAsync(d_vote, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(c, r.AllReduce(d_vote, d_vote, 1, ncclInt, ncclMin, c->comm, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&v, d_vote, sizeof v, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RCN_HIP_OK;
    };
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int {
        HIP_TRY(c, hipMemcpyAsync(d_mine, mine, bytes, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(c, r.AllGather(d_mine, d_all, bytes, ncclChar, c->comm, c->stream));
        HIP_TRY(c, hipMemcpyAsync(all, d_all, (size_t)world * bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RCN_HIP_OK;
    };
    const int st = p2p_admission(c, t);
    xch.release();
    return st;
}

int rcn_hip_dp_unique_id(void* id_out) {
    if (!id_out) return RCN_HIP_ERR_INVALID_ARG;
    rcn::Rccl& r = rcn::Rccl::get();
    if (!r.ok) return RCN_HIP_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == RCN_HIP_DP_ID_BYTES, "rcn_hip.h: RCN_HIP_DP_ID_BYTES");
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return RCN_HIP_ERR_HIP;
    std::memcpy(id_out, &id, sizeof id);
    return RCN_HIP_OK;
}

int rcn_hip_dp_init(rcn_hip_ctx* c, const void* id_bytes, int rank, int world) {
    RCN_TRY(check_ctx(c));
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_init: bad id / rank / world");
    rcn::Rccl& r = rcn::Rccl::get();
    if (!r.ok) return fail(c, RCN_HIP_ERR_UNSUPPORTED, "dp_init: " + r.err);
    DevGuard g(c->device);
    if (c->comm) { HIP_TRY(c, hipStreamSynchronize(c->stream)); NCCL_TRY(c, r.CommDestroy(c->comm)); c->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    NCCL_TRY(c, r.CommInitRank(&c->comm, world, id, rank));
    c->dp_rank = rank;
    c->dp_world = world;
    p2p_release(c);
    const bool force = c->opt.dp_p2p == 2;            // 2: also at world size 1 (exercises the whole set-up path on one GPU)
    if ((world > 1 || force) && world <= kP2PMaxWorld && c->opt.dp_p2p != 0 && c->dense_err.empty()) RCN_TRY(p2p_bootstrap_over_rccl(c));
    return RCN_HIP_OK;
}

// A peer wait that expired inside the resident kernel's data-parallel form is a property of the group just torn down (a peer that
// left), not of this context's single-GPU resident path: reported by dp_finalize, then cleared with the group.
static void clear_xcd_dp_timeout(rcn_hip_ctx* c) {
    if (c->xcd_dp_used && c->xerr_host && *c->xerr_host == 1u) (void)xcd_clear_error(c);
    c->xcd_dp_used = false;
}

int rcn_hip_dp_finalize(rcn_hip_ctx* c) {
    RCN_TRY(check_ctx(c));
    DevGuard g(c->device);
    if (!c->comm) {
        if (c->p2p.exported) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const int sticky = sticky_errors(c);
            const std::string sticky_msg = c->err;
            p2p_release(c); c->dp_rank = 0; c->dp_world = 1;
            clear_xcd_dp_timeout(c);
            if (sticky != RCN_HIP_OK) return fail(c, sticky, sticky_msg);
        }
        return RCN_HIP_OK;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int sticky = sticky_errors(c);        // reported, but the group is torn down all the same
    const std::string sticky_msg = c->err;
    NCCL_TRY(c, rcn::Rccl::get().CommDestroy(c->comm));
    c->comm = nullptr;
    p2p_release(c);
    c->dp_rank = 0;
    c->dp_world = 1;
    clear_xcd_dp_timeout(c);
    if (sticky != RCN_HIP_OK) return fail(c, sticky, sticky_msg);
    return RCN_HIP_OK;
}

/* ---- the peer all-reduce without RCCL: explicit handle exchange (what rcn_hip_dp_init does internally over RCCL) ---- */
int rcn_hip_dp_p2p_export(rcn_hip_ctx* c, void* handles_out) {
    RCN_TRY(check_ctx(c));
    if (!handles_out) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_export: NULL pointer");
    static_assert(kP2PHandleBytes == RCN_HIP_DP_P2P_HANDLE_BYTES, "rcn_hip.h: RCN_HIP_DP_P2P_HANDLE_BYTES");
    DevGuard g(c->device);
    return p2p_export(c, handles_out);
}

int rcn_hip_dp_p2p_attach(rcn_hip_ctx* c, const void* all_handles, int rank, int world) {
    RCN_TRY(check_ctx(c));
    if (!all_handles) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_attach: NULL pointer");
    DevGuard g(c->device);
    RCN_TRY(p2p_attach(c, all_handles, rank, world));
    c->p2p.on = world > 1;
    return RCN_HIP_OK;
}

int rcn_hip_dp_p2p_selftest(rcn_hip_ctx* c, int iters, unsigned* mismatches, unsigned* timed_out) {
    RCN_TRY(check_ctx(c));
    if (!mismatches || !timed_out || iters < 1) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_selftest: bad arguments");
    DevGuard g(c->device);
    RCN_TRY(p2p_selftest(c, iters, mismatches, timed_out));
    if (*mismatches || *timed_out) { c->p2p.on = false; return RCN_HIP_OK; }
    // the in-kernel form of the exchange, same verdict rule (the caller's ranks see the same result and decide alike)
    unsigned bad2 = 0, to2 = 0;
    if (c->opt.dp_fused) {
        RCN_TRY(p2p_selftest_fused(c, iters, &bad2, &to2));
        c->p2p.fused = bad2 == 0 && to2 == 0;
    }
    *mismatches += bad2;
    *timed_out |= to2;
    if (c->opt.dp_fused && c->dtype == RCN_HIP_F32 && bad2 == 0 && to2 == 0) {
        unsigned bad3 = 0, to3 = 0;
        RCN_TRY(p2p_selftest_push(c, iters, &bad3, &to3));
        c->p2p.push = bad3 == 0 && to3 == 0;
        *mismatches += bad3;
        *timed_out |= to3;
    }
    return RCN_HIP_OK;
}

int rcn_hip_dp_p2p_admit(rcn_hip_ctx* c, int rank, int world, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min, void* user) {
    RCN_TRY(check_ctx(c));
    if (!allgather || !vote_min || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world)
        return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_p2p_admit: bad callbacks / rank / world (1..8 ranks)");
    RCN_TRY(need_dense(c));
    DevGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    p2p_release(c);
    c->dp_rank = rank;
    c->dp_world = world;
    P2PTransport t;
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int {
        return allgather(user, mine, all, bytes) == 0 ? RCN_HIP_OK : fail(c, RCN_HIP_ERR_HIP, "dp_p2p_admit: the caller's allgather failed");
    };
    t.vote_min = [&](int& v) -> int {
        return vote_min(user, &v) == 0 ? RCN_HIP_OK : fail(c, RCN_HIP_ERR_HIP, "dp_p2p_admit: the caller's vote failed");
    };
    return p2p_admission(c, t);
}

int rcn_hip_dp_admission_rehearse(int rank, int world, const char* faults, rcn_hip_allgather_fn allgather, rcn_hip_vote_min_fn vote_min, void* user,
                                   int* form_out, int* resident_out) {
    if (!allgather || !vote_min || !form_out || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world) return RCN_HIP_ERR_INVALID_ARG;
    const std::string all_faults = faults ? faults : "";
    auto faulty = [&](const char* stage) {
        const std::string want = std::string(stage) + ":" + std::to_string(rank);
        size_t pos = 0;
        while (pos <= all_faults.size()) {
            const size_t end = all_faults.find(',', pos);
            if (all_faults.substr(pos, end == std::string::npos ? std::string::npos : end - pos) == want) return true;
            if (end == std::string::npos) break;
            pos = end + 1;
        }
        return false;
    };
    AdmissionOps ops;
    ops.do_export = [&](char* h) { std::memset(h, 0, kP2PHandleBytes); h[0] = (char)(rank + 1); return faulty("export") ? 0 : 1; };
    ops.attach = [&](const char* all) {
        for (int r = 0; r < world; ++r)
            if (all[(size_t)r * kP2PHandleBytes] != (char)(r + 1)) return 0;          // the transport delivered every rank's bytes, in rank order
        return faulty("attach") ? 0 : 1;
    };
    ops.known_answer = [&]() { return faulty("kat") ? 0 : 1; };
    ops.wants_fused = [&]() { return faulty("nofused") ? 0 : 1; };
    ops.tagged_words = [&]() { return (faulty("ll") || faulty("llskip")) ? 0 : 1; };
    ops.wants_push = [&]() { return faulty("f64") ? 0 : 1; };
    ops.pushed_words = [&]() { return (faulty("push") || faulty("pushskip")) ? 0 : 1; };
    ops.clear_sticky = [&]() { return faulty("clear") ? 0 : 1; };
    P2PTransport t;
    t.allgather = [&](const void* mine, void* all, size_t bytes) -> int { return allgather(user, mine, all, bytes) == 0 ? RCN_HIP_OK : RCN_HIP_ERR_HIP; };
    t.vote_min = [&](int& v) -> int { return vote_min(user, &v) == 0 ? RCN_HIP_OK : RCN_HIP_ERR_HIP; };
    AdmissionOutcome out;
    const int st = admission_protocol(world, ops, t, out);
    *form_out = out.on ? (out.fused ? 2 : 1) : 0;
    if (resident_out) *resident_out = (out.on && out.push) ? 1 : 0;
    return st;
}

int rcn_hip_dp_p2p_active(const rcn_hip_ctx* c) { return c && c->p2p.on ? (c->p2p.fused ? 2 : 1) : 0; }

int rcn_hip_train_epoch_gathers(rcn_hip_ctx* c, size_t B) {
    if (!c) return 0;
    DevGuard g(c->device);
    return use_xcd(c, B) && xcd_gather(c) ? 1 : 0;
}

int rcn_hip_dp_phase_us(rcn_hip_ctx* c, double* out, size_t cap) {
    RCN_TRY(check_ctx(c));
    if (!out || cap < 8) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_phase_us: out must hold 8 doubles");
    if (!c->xcdbuf.p || (c->xcd_B != 256 && c->xcd_B != 128) || c->dtype != RCN_HIP_F32)
        return fail(c, RCN_HIP_ERR_STATE, "dp_phase_us: no clocked data-parallel launch at a shard of 256 or 128 has run (option xcd_dp_phase)");
    DevGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    long long ph[kXcdWorkers * 4];
    HIP_TRY(c, hipMemcpy(ph, xcd_bufs<float>(c, c->xcd_B).phase, sizeof ph, hipMemcpyDeviceToHost));
    double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}, step = 0, steps = 0;
    int cnt[4] = {0, 0, 0, 0}, nstep = 0;
    for (int w = 0; w < kXcdWorkers; ++w) {
        const long long nb = ph[w * 4 + 2], role = ph[w * 4 + 3];
        if (nb <= 0) continue;
        step += (double)ph[w * 4 + 1] * 0.01 / (double)nb; ++nstep; steps = (double)nb;
        if (role < 1 || role > 3) continue;
        const double us = (double)ph[w * 4 + 0] * 0.01 / (double)nb;              // 100 MHz ticks -> us per step
        sum[role] += us; ++cnt[role];
        if (us > mx[role]) mx[role] = us;
    }
    if (nstep == 0) return fail(c, RCN_HIP_ERR_STATE, "dp_phase_us: no clocked data-parallel launch has run since the option was set");
    for (int r = 1; r <= 3; ++r) { out[2 * (r - 1)] = cnt[r] ? sum[r] / cnt[r] : 0.0; out[2 * (r - 1) + 1] = mx[r]; }
    out[6] = step / nstep;
    out[7] = steps;
    return RCN_HIP_OK;
}

int rcn_hip_last_timeout(const rcn_hip_ctx* c, uint32_t* words, size_t cap) {
    if (!c || !words || !c->xrec_valid) return 0;
    const size_t n = cap < (size_t)kXcdRecWords + 1 ? cap : (size_t)kXcdRecWords + 1;
    for (size_t i = 0; i < n; ++i) words[i] = c->xrec[i];
    return (int)n;
}
const char* rcn_hip_last_timeout_text(const rcn_hip_ctx* c) { return c ? c->xlast.c_str() : ""; }

int rcn_hip_train_epoch_resident(rcn_hip_ctx* c, size_t B) {
    if (!c || B == 0) return 0;
    DevGuard g(c->device);
    return need_dense(c) == RCN_HIP_OK && use_pipe(c, B) && use_xcd(c, B) ? 1 : 0;
}

int rcn_hip_dp_resident(rcn_hip_ctx* c, size_t B_shard) {
    if (!c) return 0;
    DevGuard g(c->device);
    return dp_on_xcd(c, B_shard) ? 1 : 0;
}

int rcn_hip_dp_epoch_steps_dev(rcn_hip_ctx* c, size_t first_batch, size_t n_batches, double eta, void* loss_dev) {
    RCN_TRY(check_ctx(c));
    if (!c->comm && !c->p2p.on) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_epoch_steps: rcn_hip_dp_init was not called");
    if (c->p2p.on && *c->p2p.err_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "dp_epoch_steps: the peer exchange timed out in an earlier call; the replicas are no longer in step");
    if (c->epoch_nb == 0) return fail(c, RCN_HIP_ERR_STATE, "dp_epoch_steps: no epoch begun (rcn_hip_epoch_begin_dev), or another training call has re-packed the image since");
    if (first_batch > c->epoch_nb || n_batches > c->epoch_nb - first_batch) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_epoch_steps: batches beyond the begun epoch");
    if (n_batches == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    const size_t B = c->epoch_B, seg = c->epoch_seg;
    DevGuard g(c->device);
    if (!(dp_on_xcd(c, B)))
        return fail(c, RCN_HIP_ERR_UNSUPPORTED, "dp_epoch_steps: only where the data-parallel step runs on the resident kernel (rcn_hip_dp_resident); "
                                                "rcn_hip_dp_train_epoch_dev packs and runs its batches itself on every form");
    RCN_TRY(ensure_dense_ws(c, B));
    RCN_TRY(ensure_pipe_ws(c, B));
    RCN_TRY(ensure_xcd_ws(c, B));
    RCN_TRY(enqueue_xcd_epoch(c, nullptr, nullptr, nullptr, B, n_batches, eta, loss_dev, false, true, first_batch, seg, true));
    HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    return RCN_HIP_OK;
}

int rcn_hip_dp_world(const rcn_hip_ctx* c) { return c ? c->dp_world : 0; }
int rcn_hip_dp_rank(const rcn_hip_ctx* c) { return c ? c->dp_rank : -1; }

int rcn_hip_dp_broadcast_params(rcn_hip_ctx* c, int root) {
    RCN_TRY(check_ctx(c));
    if (!c->comm) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_broadcast_params: rcn_hip_dp_init was not called");
    if (root < 0 || root >= c->dp_world) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_broadcast_params: bad root");
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    NCCL_TRY(c, rcn::Rccl::get().Broadcast(c->params.p, c->params.p, (size_t)c->nd.P, c->dtype == RCN_HIP_F64 ? ncclDouble : ncclFloat, root,
                                           c->comm, c->stream));
    return RCN_HIP_OK;
}

static int dp_epoch_impl(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta, void* loss_dev, bool launch) {
    RCN_TRY(check_ctx(c));
    if (!c->comm && !c->p2p.on) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: rcn_hip_dp_init was not called");
    if (c->p2p.on && *c->p2p.err_host != 0)
        return fail(c, RCN_HIP_ERR_HIP, "dp_train_epoch: the peer all-reduce timed out waiting for rank " + std::to_string((int)*c->p2p.err_host - 1) +
                                            "'s peers in an earlier call; the replicas are no longer in step");
    if (!X || !Y) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: NULL pointer");
    if (B == 0 || B > 0x7fffffffULL / 2) return fail(c, RCN_HIP_ERR_INVALID_ARG, "dp_train_epoch: shard batch size must be in 1..2^30");
    if (nb == 0) return RCN_HIP_OK;
    RCN_TRY(need_params(c));
    DevGuard g(c->device);
    RCN_TRY(ensure_dense_ws(c, B));
    const size_t es = c->esz(), P = (size_t)c->nd.P, F = c->nd.dims[0], Cc = c->nd.dims[c->nd.L];
    HIP_TRY(c, ws_ensure(c, c->grad, (P + 1) * es));          // [gradient | loss]: one all-reduce carries both
    rcn::Rccl& r = rcn::Rccl::get();
    const double Bg = (double)B * (double)c->dp_world;          // the global batch.len() of rcn.rs:214
    const double scale = eta / Bg, loss_scale = 1.0 / (2.0 * Bg);
    char* gbuf = (char*)c->grad.p;
    void* lslot = gbuf + P * es;
    const bool f64 = c->dtype == RCN_HIP_F64;
    if (dp_on_xcd(c, B)) {
        // the resident one-XCD kernel with the exchange between its gradient MFMAs and its update (dense_xcd.hpp, DP = true): one
        // launch per segment of the epoch image, nothing to capture
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, nb));
        RCN_TRY(ensure_xcd_ws(c, B));
        if (!launch) return RCN_HIP_OK;
        RCN_TRY(enqueue_xcd_epoch(c, X, Y, perm, B, nb, eta, loss_dev, false, false, 0, 0, true));
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));
        return RCN_HIP_OK;
    }
    if (c->p2p.on && c->dense_path != 1 && p2_supported(c->nd, B)) {
        // the lean pipeline with the exchange inside its third kernel (dense_p2_dp.hpp)
        RCN_TRY(ensure_pipe_ws(c, B));
        RCN_TRY(ensure_pack_ws(c, B, nb));
        // captured once per (pointers, B, n_batches, eta) and replayed: three launches per step would otherwise be bound by the
        // host's launch rate (~6 us each), not by the GPU.  Sequence numbers inside the graph are offsets from a device word.
        if (!c->opt.dp_graph) {
            if (!launch) return RCN_HIP_OK;
            RCN_TRY(f64 ? enqueue_pipe_steps_dp<double>(c, X, Y, perm, B, nb, eta, loss_dev, false, c->p2p.fused)
                        : enqueue_pipe_steps_dp<float>(c, X, Y, perm, B, nb, eta, loss_dev, false, c->p2p.fused));
        } else {
            const EpochKey key{X, Y, perm, B, nb, eta, loss_dev};
            auto it = c->dp_graphs.find(key);
            if (it == c->dp_graphs.end()) {
                hipGraph_t graph = nullptr;
                HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                const int st = f64 ? enqueue_pipe_steps_dp<double>(c, X, Y, perm, B, nb, eta, loss_dev, true, c->p2p.fused)
                                   : enqueue_pipe_steps_dp<float>(c, X, Y, perm, B, nb, eta, loss_dev, true, c->p2p.fused);
                hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (st != RCN_HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
                HIP_TRY(c, e);
                hipGraphExec_t exec = nullptr;
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                HIP_TRY(c, e);
                if (c->dp_graphs.size() >= 16) drop_graphs(c);
                it = c->dp_graphs.emplace(key, exec).first;
            }
            if (!launch) return RCN_HIP_OK;                   // rcn_hip_dp_prepare_epoch_dev: instantiated, not run
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, c->stream, c->p2p.err_dev + 16, c->p2p.seq);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipGraphLaunch(it->second, c->stream));
            c->p2p.seq += (unsigned)nb;
        }
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));   // read at the next call
        return RCN_HIP_OK;
    }
    if (!launch) return RCN_HIP_OK;                           // nothing to prepare on the eager paths
    if (c->p2p.on) {
        // gradient kernels write straight into this rank's exported slot; ONE kernel then waits for the peers' flags, reads
        // all `world` slots over xGMI, adds them in rank order and applies the update (dp_p2p.hpp)
        for (size_t j = 0; j < nb; ++j) {
            const void* xb = perm ? X : (const char*)X + j * B * F * es;
            const void* yb = perm ? Y : (const char*)Y + j * B * Cc * es;
            const int32_t* ib = perm ? perm + j * B : nullptr;
            char* slot = (char*)c->p2p.local_buf + (size_t)((c->p2p.seq + 1) & 1u) * c->p2p.stride * es;
            void* lj = loss_dev ? (char*)loss_dev + j * es : nullptr;
            if (f64) {
                RCN_TRY(launch_fwd<double>(c, true, xb, yb, ib, B, nullptr));
                RCN_TRY(launch_wgrad<double>(c, false, xb, ib, B, 0.0, slot, slot + P * es, loss_scale));
                RCN_TRY(p2p_step<double>(c, 0, scale, lj, p2p_timeout_ticks(c)));
            } else {
                RCN_TRY(launch_fwd<float>(c, true, xb, yb, ib, B, nullptr));
                RCN_TRY(launch_wgrad<float>(c, false, xb, ib, B, 0.0, slot, slot + P * es, loss_scale));
                RCN_TRY(p2p_step<float>(c, 0, scale, lj, p2p_timeout_ticks(c)));
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->p2p.err_host, c->p2p.err_dev, 4, hipMemcpyDeviceToHost, c->stream));   // read at the next call
        return RCN_HIP_OK;
    }
    for (size_t j = 0; j < nb; ++j) {
        const void* xb = perm ? X : (const char*)X + j * B * F * es;
        const void* yb = perm ? Y : (const char*)Y + j * B * Cc * es;
        const int32_t* ib = perm ? perm + j * B : nullptr;
        if (f64) {
            RCN_TRY(launch_fwd<double>(c, true, xb, yb, ib, B, nullptr));
            RCN_TRY(launch_wgrad<double>(c, false, xb, ib, B, 0.0, gbuf, lslot, loss_scale));
        } else {
            RCN_TRY(launch_fwd<float>(c, true, xb, yb, ib, B, nullptr));
            RCN_TRY(launch_wgrad<float>(c, false, xb, ib, B, 0.0, gbuf, lslot, loss_scale));
        }
        NCCL_TRY(c, r.AllReduce(gbuf, gbuf, P + 1, f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, c->stream));
        if (f64)
            hipLaunchKernelGGL((k_apply_gradient<double>), dim3(grid_for((int)P, 256)), dim3(256), 0, c->stream, (double*)c->params.p,
                               (const double*)gbuf, scale, (int)P);
        else
            hipLaunchKernelGGL((k_apply_gradient<float>), dim3(grid_for((int)P, 256)), dim3(256), 0, c->stream, (float*)c->params.p,
                               (const float*)gbuf, (float)scale, (int)P);
        HIP_TRY(c, hipGetLastError());
        if (loss_dev) HIP_TRY(c, hipMemcpyAsync((char*)loss_dev + j * es, lslot, es, hipMemcpyDeviceToDevice, c->stream));
    }
    return RCN_HIP_OK;
}

int rcn_hip_dp_train_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta,
                               void* loss_dev) {
    return dp_epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, true);
}

int rcn_hip_dp_prepare_epoch_dev(rcn_hip_ctx* c, const void* X, const void* Y, const int32_t* perm, size_t B, size_t nb, double eta,
                                 void* loss_dev) {
    return dp_epoch_impl(c, X, Y, perm, B, nb, eta, loss_dev, false);
}
