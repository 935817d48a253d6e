"""GPU tests added in round 2 (run with -m gpu on an MI355X), all through the C ABI:

* the >= 200-step loss-curve parity check on the BENCH workload (BASELINE.json north_star "at matched loss curve");
* rcn_hip_epoch_begin_dev / rcn_hip_epoch_steps_dev == rcn_hip_train_epoch_dev bit for bit;
* regressions for the round-1 advisor findings (stale image-epoch graphs after a scale change, sticky in-kernel timeouts
  surfacing from synchronize / get_params, graphs dropped when a workspace is released);
* the statistics of rcn_hip_init_params (rcn.rs:500-523: every W then b of a layer ~ N(0,1), un-scaled).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.rcn_oracle import DEFAULT_LAYERS, one_hot, synthetic_images, synthetic_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BENCH_B, BENCH_N, BENCH_ETA, BENCH_DIMS = 256, 16384, 3.0, [784, 30, 10]


@pytest.fixture(scope="module")
def amd():
    import mercer_research_amd as m
    return m


@pytest.fixture(scope="module")
def bench_workload(oracle):
    """bench.py's workload on the CPU side: 16 384 synthetic pictures (seed 1234) -> oracle features -> standardised; N(0,1)
    parameters of seed 42 (un-scaled, rcn.rs:500-523)."""
    imgs, labels = synthetic_images(BENCH_N, seed=1234)
    feats = oracle.features(imgs, DEFAULT_LAYERS)
    m, s = oracle.gen_scales(feats)
    X, Y = oracle.standardize(feats, m, s), one_hot(labels)
    ws, bs = synthetic_params(BENCH_DIMS, seed=42)
    return imgs, labels, X, Y, ws, bs


def _oracle_curve(oracle, X, Y, ws, bs, perms, nb, B, eta):
    h = oracle.net(ws, bs)
    costs = []
    for p in perms:
        for j in range(nb):
            idx = p[j * B:(j + 1) * B]
            xb, yb = np.ascontiguousarray(X[idx]), np.ascontiguousarray(Y[idx])
            costs.append(oracle.lib.rcn_o_train_batch(C.byref(h.net), xb.ctypes.data_as(C.POINTER(C.c_double)), yb.ctypes.data_as(C.POINTER(C.c_double)), B, eta))
    return np.array(costs), h.weights(), h.biases()


# f32 envelope of the curve.  Measured on MI355X (printed by the test): bench net 784-30-10 -- 4.1e-7 over the first 32 steps, 2.3e-5
# worst over all 256, 1.4e-7 on the mean; the reference's test net 784-10-10-10 -- 8.0e-8 / 8.0e-8 / 8.3e-10.  eta = 3 on un-scaled
# N(0,1) parameters saturates the sigmoids, so f32 rounding differences do grow along the trajectory.  Asserted: 1e-5 over the first
# 32 steps, ten times the measurement (2.5e-4 <= SURVEY 8(c)'s 1e-3) at every step, 1e-5 on the mean (round 2: 1e-3 / 5e-2 / 5e-3).
F32_TIGHT_STEPS, F32_TIGHT_RTOL = 32, 1e-5
F32_CURVE_RTOL, F32_MEAN_RTOL = 2.5e-4, 1e-5


@pytest.mark.parametrize("dtype", [1, 0], ids=["f64", "f32"])
def test_loss_curve_256_steps_of_the_reference_test_net_matches_the_cpu_restatement(amd, oracle, bench_workload, dtype):
    """The same 256-step curve for the reference's own test net, two hidden layers of ten (rcn.rs:558,577): in the f32 context these
    steps run on the resident kernel's two-hidden-layer instantiation, in the f64 context on the generic pipeline."""
    _loss_curve_case(amd, oracle, bench_workload, dtype, [784, 10, 10, 10])


@pytest.mark.parametrize("dtype", [1, 0], ids=["f64", "f32"])
def test_loss_curve_256_steps_on_the_bench_workload_matches_the_cpu_restatement(amd, oracle, bench_workload, dtype):
    """256 consecutive train_batch steps of bench.py's workload (B = 256, eta = 3.0, synthetic_params(seed = 42), 16 384 synthetic
    pictures, four device-shuffled epochs through rcn_hip_epoch_begin_dev / rcn_hip_epoch_steps_dev -- the loop bench.py times) with
    the per-step quadratic cost recorded, against oracle/rcn_oracle.c's train_batch (rcn.rs:176-223, f64) on the identical
    batches in the identical order.  f64 context: <= 1e-9 relative at every step and on the final parameters."""
    _loss_curve_case(amd, oracle, bench_workload, dtype, BENCH_DIMS)


def _loss_curve_case(amd, oracle, bench_workload, dtype, dims):
    import torch
    from mercer_research_amd.device import DeviceRCN
    imgs, labels, X, Y, ws, bs = bench_workload
    if dims != BENCH_DIMS:
        ws, bs = synthetic_params(dims, seed=42)
    B, nb, steps = BENCH_B, BENCH_N // BENCH_B, 256
    d = DeviceRCN(dtype=dtype, feedforward_cfg=dims[1:-1])
    d.set_params(ws, bs)
    imgs_d, labels_d = d.to_device(imgs), d.to_device(labels)
    Xd, Yd = d.load_data(imgs_d, labels_d)                      # HIP features + gen_scales + standardise
    assert np.allclose(Xd.double().cpu().numpy(), X, rtol=1e-5 if dtype == 0 else 1e-10, atol=1e-6 if dtype == 0 else 1e-12)   # (x - mean) / sd cancels near the mean
    perm = torch.empty(BENCH_N, dtype=torch.int32, device=d.device)
    loss = d.empty(steps)
    perms = []
    for e in range(steps // nb):
        d.shuffle(perm, BENCH_N, 1, seed=0xC0FFEE + e)
        d.synchronize()
        perms.append(perm.cpu().numpy().astype(np.int64))
        assert np.array_equal(np.sort(perms[-1]), np.arange(BENCH_N))
        d.epoch_begin(Xd, Yd, perm, B, nb)
        d.epoch_steps(0, 40, BENCH_ETA, loss[e * nb:])           # an epoch walked in two pieces, as a session interrupted mid-epoch
        d.epoch_steps(40, nb - 40, BENCH_ETA, loss[e * nb + 40:])
    gw, gb = d.get_params()
    gpu = loss.double().cpu().numpy()
    d.rcn.close()
    cpu, rw, rb = _oracle_curve(oracle, X, Y, ws, bs, perms, nb, B, BENCH_ETA)
    assert np.all(np.isfinite(gpu)) and np.all(np.isfinite(cpu))
    rel = np.abs(gpu - cpu) / np.abs(cpu)
    if dtype == 1:
        assert rel.max() <= 1e-9, (rel.max(), int(rel.argmax()))
        for a, b in zip(gw + gb, rw + rb):
            assert np.all(np.abs(a - b) <= 1e-9 * np.abs(b) + 1e-10)
    else:
        print(f"MEASURED f32 loss curve {dims}: first {F32_TIGHT_STEPS} steps {rel[:F32_TIGHT_STEPS].max():.3e}, whole curve {rel.max():.3e}, mean {abs(gpu.mean() - cpu.mean()) / cpu.mean():.3e}")
        assert rel[:F32_TIGHT_STEPS].max() <= F32_TIGHT_RTOL, (rel[:F32_TIGHT_STEPS].max(), int(rel[:F32_TIGHT_STEPS].argmax()))
        assert rel.max() <= F32_CURVE_RTOL, (rel.max(), int(rel.argmax()))
        assert abs(gpu.mean() - cpu.mean()) <= F32_MEAN_RTOL * cpu.mean()
    # the curve is a training curve: the cost at the end is below the cost at the start on both sides (the bench net; the ten-unit
    # test net at eta = 3 on un-scaled parameters need not descend -- then both sides must agree that it does not)
    assert (gpu[-16:].mean() < gpu[:16].mean()) == (cpu[-16:].mean() < cpu[:16].mean())
    if dims == BENCH_DIMS:
        assert cpu[-16:].mean() < cpu[:16].mean()


@pytest.mark.parametrize("dtype", [0, 1], ids=["f32", "f64"])
def test_epoch_begin_steps_equals_train_epoch_bit_for_bit(amd, dtype, monkeypatch):
    """rcn_hip_epoch_begin_dev + any split of rcn_hip_epoch_steps_dev == rcn_hip_train_epoch_dev over the same batches: same
    kernels on the same image, so the parameters and the recorded costs are identical bit for bit -- including an epoch that
    spans both segments of the image, and from u8 pictures (rcn_hip_epoch_begin_images_dev)."""
    import torch
    from mercer_research_amd.device import DeviceRCN
    monkeypatch.setenv("RCN_HIP_PACK_SEGMENT_BYTES", str(5 * 49 * 256 * 16 * (8 if dtype == 1 else 4)))     # 5 batches per segment
    B, nb, N = 256, 9, 2560
    imgs, labels = synthetic_images(N, seed=31)
    ws, bs = synthetic_params(BENCH_DIMS, seed=3)
    ws = [w * 0.1 for w in ws]
    perm_h = np.random.default_rng(8).permutation(N).astype(np.int32)
    out = {}
    for form in ("train_epoch", "begin_steps", "begin_steps_images", "train_epoch_images"):
        d = DeviceRCN(dtype=dtype)
        d.set_params(ws, bs)
        dev = d.to_device(imgs)
        raw = d.features(dev)
        d.gen_scales(raw)
        Xd = d.features(dev, standardize=True)
        Yd = d.to_device(one_hot(labels), d.tdtype)
        perm = d.to_device(perm_h)
        loss = d.empty(nb)
        if form == "train_epoch":
            d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss)
        elif form == "train_epoch_images":
            d.train_epoch_images(dev, Yd, perm, B, nb, 3.0, loss)
        else:
            d.epoch_begin(dev if form.endswith("images") else Xd, Yd, perm, B, nb)
            for j0, n in ((0, 1), (1, 3), (4, 0), (4, 5)):           # crosses the segment boundary at batch 5 inside the last piece
                d.epoch_steps(j0, n, 3.0, loss[j0:] if n else None)
            d.epoch_steps(4, 5, 3.0, None, prepare_only=True)        # instantiating again is a no-op
        gw, gb = d.get_params()
        out[form] = (gw + gb, loss.cpu().numpy().copy())
        if form == "begin_steps":
            with pytest.raises(amd.RcnHipError):
                d.epoch_steps(8, 2, 3.0, None)                       # beyond the begun epoch
            d.train_batch(Xd[:B], Yd[:B], 3.0, None)                 # re-packs the image: the begun epoch is over
            with pytest.raises(amd.RcnHipError):
                d.epoch_steps(0, 1, 3.0, None)
        d.rcn.close()
    ref = out["train_epoch"]
    for form in ("begin_steps", "begin_steps_images", "train_epoch_images"):
        for a, b in zip(out[form][0], ref[0]):
            assert np.array_equal(a, b), form
        assert np.array_equal(out[form][1], ref[1]), form


def test_epoch_begin_is_refused_where_the_pipeline_does_not_run(amd):
    from mercer_research_amd.device import DeviceRCN
    ws, bs = synthetic_params(BENCH_DIMS, seed=3)
    d = DeviceRCN(dtype=0)
    d.set_params(ws, bs)
    d.set_dense_path(1)                                              # sample-tile kernels: no epoch image
    X = d.empty(512, 784).zero_()
    Y = d.empty(512, 10).zero_()
    with pytest.raises(amd.RcnPanic):
        d.epoch_begin(X, Y, None, 256, 2)
    with pytest.raises(amd.RcnHipError):
        d.epoch_steps(0, 1, 3.0, None)
    d.rcn.close()


# ------------------------------------------------------------------------------------------------ advisor regressions

def test_image_epoch_graph_is_not_replayed_with_a_stale_scale_set(amd):
    """the graphs of rcn_hip_train_epoch_images_dev bake (mean, sd, reciprocal, kernel variant) into the captured
    feature launch.  prepare/capture -> set_scale(other) -> replay with the same arguments must standardise with the NEW scale_set,
    i.e. equal features_dev(standardize = 1) under the new scale followed by train_epoch_dev."""
    from mercer_research_amd.device import DeviceRCN
    B, nb, N = 256, 3, 768
    imgs, labels = synthetic_images(N, seed=77)
    ws, bs = synthetic_params(BENCH_DIMS, seed=5)
    ws = [w * 0.1 for w in ws]
    for dtype in (0, 1):
        res = []
        for images_path in (True, False):
            d = DeviceRCN(dtype=dtype)
            d.set_params(ws, bs)
            dev = d.to_device(imgs)
            Y = d.to_device(one_hot(labels), d.tdtype)
            d.rcn.scale_set = (20.0, 50.0)
            if images_path:
                d.train_epoch_images(dev, Y, None, B, nb, 3.0, None)          # captures with (20, 50) and runs once
                d.set_params(ws, bs)
                d.rcn.scale_set = (35.5, 81.25)                                  # load_data(test) overwrites scale_set (rcn.rs:136-137, 406)
                d.train_epoch_images(dev, Y, None, B, nb, 3.0, None)          # same arguments: must NOT replay the old standardisation
            else:
                d.rcn.scale_set = (35.5, 81.25)
                X = d.features(dev, standardize=True)
                d.train_epoch(X, Y, None, B, nb, 3.0, None)
            gw, gb = d.get_params()
            res.append(gw + gb)
            d.rcn.close()
        for a, b in zip(*res):
            assert np.array_equal(a, b)


def test_gen_scales_also_invalidates_image_epoch_graphs(amd):
    from mercer_research_amd.device import DeviceRCN
    B, nb, N = 256, 2, 512
    imgs, labels = synthetic_images(N, seed=78)
    imgs2, _ = synthetic_images(N, seed=79)
    ws, bs = synthetic_params(BENCH_DIMS, seed=5)
    ws = [w * 0.1 for w in ws]
    res = []
    for images_path in (True, False):
        d = DeviceRCN(dtype=0)
        d.set_params(ws, bs)
        dev, dev2 = d.to_device(imgs), d.to_device((imgs2 // 2).astype(np.uint8))
        Y = d.to_device(one_hot(labels), d.tdtype)
        d.gen_scales(d.features(dev))
        if images_path:
            d.train_epoch_images(dev, Y, None, B, nb, 3.0, None)
            d.set_params(ws, bs)
        d.gen_scales(d.features(dev2))                                        # a different data set's statistics
        if images_path:
            d.train_epoch_images(dev, Y, None, B, nb, 3.0, None)
        else:
            d.train_epoch(d.features(dev, standardize=True), Y, None, B, nb, 3.0, None)
        gw, gb = d.get_params()
        res.append(gw + gb)
        d.rcn.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def _spawn_ranks(tmp_path, world, dtype, case_name, env_extra, dims=(784, 30, 10), Bs=256, nb=2, seed=17):
    rng = np.random.default_rng(31)
    dims = list(dims)
    Xs = [np.maximum(rng.standard_normal((Bs * nb, dims[0])), 0.0) for _ in range(world)]
    Ys = [one_hot(rng.integers(0, dims[-1], Bs * nb), dims[-1]) for _ in range(world)]
    np.savez(tmp_path / "case.npz", dims=dims, Bs=Bs, nb=nb, seed=seed, **{f"X{r}": Xs[r] for r in range(world)}, **{f"Y{r}": Ys[r] for r in range(world)})
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "_p2p_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(dtype), str(tmp_path), case_name], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        logs.append(out.decode(errors="replace")[-3000:])
    assert all(pr.returncode == 0 for pr in procs), "\n----\n".join(logs)
    return Xs, Ys, logs


@pytest.mark.parametrize("xcd", ["0", "1"], ids=["two-kernel-pipeline", "resident-one-xcd-kernel"])
def test_sticky_timeout_of_the_peer_exchange_surfaces_from_synchronize_and_get_params(amd, tmp_path, xcd):
    """a peer wait that times out in the LAST call used to go unnoticed (the word was only looked at by the next
    epoch call).  Two ranks on this GPU; rank 1 stops after one step, rank 0's second step waits (20 ms bound) for data that never
    comes: rcn_hip_synchronize, rcn_hip_get_params and rcn_hip_dp_finalize must each report it."""
    _, _, logs = _spawn_ranks(tmp_path, 2, 0, "sticky", {"RCN_HIP_XCD": xcd})
    assert "STICKY synchronize=error get_params=error finalize=error" in logs[0], logs[0]
    assert f"RESIDENT {xcd}" in logs[0], logs[0]                    # ... on the step kernel this case is about


# (fault, the form every rank must land on).  No fault: the in-kernel exchange.  A rank that cannot export / map its peers / gets a
# wrong known-answer sum: nobody uses the peer exchange (0).  A rank whose tagged-word self-test fails, or that stays silent in it
# so that its peer's waits really expire: the kernel-boundary exchange (1) on every rank, with the sticky error word cleared.
@pytest.mark.parametrize("fault,expect", [("", 2), ("export:1", 0), ("attach:0", 0), ("kat:1", 0), ("ll:1", 1), ("llskip:0", 1), ("kat:0,ll:1", 0)],
                         ids=["no-fault", "export-fails-on-rank1", "peer-map-fails-on-rank0", "known-answer-mismatch-on-rank1",
                              "tagged-word-selftest-mismatch-on-rank1", "tagged-word-selftest-timeout", "two-faults"])
def test_admission_votes_land_every_rank_on_the_same_form_and_the_epoch_equals_the_oracle(amd, oracle, tmp_path, fault, expect):
    """The admission logic of rcn_hip_dp_init (one procedure, p2p_admission in csrc/rcn_hip_api_dp.ipp, run here over gloo by
    rcn_hip_dp_p2p_admit) under injected faults: every rank executes the same votes, lands on the same form, and two epochs of the
    sharded loop on that form equal the oracle's train_batch on the concatenated global batches (f64, 1e-11)."""
    env = {"RCN_HIP_DP_FAULT": fault} if fault else {}
    if "llskip" in fault:
        env["RCN_HIP_DP_TIMEOUT_TICKS"] = "5000000"                # 50 ms per expired wait instead of 1 s
    world, Bs, nb, dims = 2, 256, 2, [784, 30, 10]
    Xs, Ys, logs = _spawn_ranks(tmp_path, world, 1, "admit", env, dims=dims, Bs=Bs, nb=nb)
    outs = [np.load(tmp_path / f"out{r}.npz") for r in range(world)]
    assert [int(o["active"]) for o in outs] == [expect] * world, (fault, [int(o["active"]) for o in outs], logs)
    for k in ("w0", "w1", "b0", "b1", "loss"):
        assert np.array_equal(outs[0][k], outs[1][k]), k           # replicas bit-identical on every form
    ws, bs = synthetic_params(dims, seed=17)
    rw, rb, costs = [w * 0.1 for w in ws], bs, []
    for ep in range(2):
        for j in range(nb):
            xb = np.concatenate([X[j * Bs:(j + 1) * Bs] for X in Xs])
            yb = np.concatenate([Y[j * Bs:(j + 1) * Bs] for Y in Ys])
            rw, rb, c = oracle.train_batch(rw, rb, xb, yb, 3.0)
            if ep == 0:
                costs.append(c)
    for a, b in zip([outs[0]["w0"], outs[0]["w1"], outs[0]["b0"], outs[0]["b1"]], [rw[0], rw[1], rb[0], rb[1]]):
        assert np.all(np.abs(a - b) <= 1e-11 * np.abs(b) + 1e-12)
    np.testing.assert_allclose(outs[0]["loss"], costs, rtol=1e-10)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_workspace_growth_failure_path_leaves_no_dangling_graphs(amd):
    """DevBuf::ensure frees the old block before it allocates; the cached graphs must be dropped whenever the old
    block is released.  Exercised through the success path (the failure path shares the code): grow every workspace between
    replays of different cached shapes and compare with fresh contexts."""
    from mercer_research_amd.device import DeviceRCN
    ws, bs = synthetic_params(BENCH_DIMS, seed=6)
    ws = [w * 0.1 for w in ws]
    rng = np.random.default_rng(0)
    X, Y = rng.random((4096, 784)), one_hot(rng.integers(0, 10, 4096))
    d = DeviceRCN(dtype=1)
    d.set_params(ws, bs)
    Xd, Yd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype)
    seq = [(256, 2), (512, 3), (256, 2), (1024, 4), (512, 3), (256, 2)]
    for B, nb in seq:
        d.train_epoch(Xd, Yd, None, B, nb, 3.0, None)
    got = d.get_params()
    d.rcn.close()
    r = DeviceRCN(dtype=1)
    r.set_params(ws, bs)
    Xd, Yd = r.to_device(X, r.tdtype), r.to_device(Y, r.tdtype)
    for B, nb in seq:
        for j in range(nb):
            r.train_batch(Xd[j * B:(j + 1) * B], Yd[j * B:(j + 1) * B], 3.0, None)
    ref = r.get_params()
    r.rcn.close()
    for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
        assert np.all(np.abs(a - b) <= 1e-11 * np.abs(b) + 1e-12)


# ------------------------------------------------------------------------------------------------ init statistics

def test_init_params_draws_unscaled_standard_normals_w_then_b_per_layer(amd):
    """rcn_hip_init_params = load_weights_and_bias (rcn.rs:425-457) with get_weight_matrix / get_bias_vector (rcn.rs:500-523): every
    entry ~ N(0,1), NOT scaled by 1/sqrt(fan-in) (the scaling is commented out at rcn.rs:509).  Moments over the 23 520 entries of
    W_0: |mean| < 4 sigma/sqrt(n), variance within 4 standard errors of 1, skew ~ 0, excess kurtosis ~ 0, a KS test against N(0,1);
    the same seed gives the same stream, different seeds differ, seed 0 is non-deterministic; and the stream is consumed W_l then
    b_l, layer by layer (rcn.rs:446-455)."""
    from scipy import stats
    r = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=amd.F64)
    r.load_weights_and_bias(12345)
    w, b = r.get_params()
    allv = np.concatenate([w[0].ravel(), b[0], w[1].ravel(), b[1]])
    n = allv.size
    assert n == 23860
    assert abs(allv.mean()) < 4 / np.sqrt(n)
    assert abs(allv.var() - 1.0) < 4 * np.sqrt(2.0 / n)
    assert abs(stats.skew(allv)) < 4 * np.sqrt(6.0 / n)
    assert abs(stats.kurtosis(allv)) < 4 * np.sqrt(24.0 / n)
    assert stats.kstest(allv, "norm").pvalue > 1e-3
    # un-scaled: a 1/sqrt(784) scaling would give W_0 a standard deviation of 0.036
    assert 0.97 < w[0].std() < 1.03 and 0.9 < w[1].std() < 1.1
    # each layer's W and b pass on their own (a generator that re-seeded per tensor, or filled b from W's tail, would not)
    assert stats.kstest(w[0].ravel(), "norm").pvalue > 1e-3 and stats.kstest(w[1].ravel(), "norm").pvalue > 1e-3
    assert abs(np.corrcoef(w[0].ravel()[:300], w[1].ravel())[0, 1]) < 0.25
    # determinism per seed
    r.load_weights_and_bias(12345)
    w2, b2 = r.get_params()
    assert all(np.array_equal(x, y) for x, y in zip(w + b, w2 + b2))
    r.load_weights_and_bias(12346)
    w3, _ = r.get_params()
    assert not np.array_equal(w[0], w3[0])
    r.load_weights_and_bias(0)
    wa, _ = r.get_params()
    r.load_weights_and_bias(0)
    wb, _ = r.get_params()
    assert not np.array_equal(wa[0], wb[0])            # seed 0: thread_rng-like, not reproducible
    # draw order: the stream fills W_0 (column-major), then b_0, then W_1, then b_1 -- the f32 context rounds the same stream
    r32 = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=amd.F32)
    r32.load_weights_and_bias(12345)
    w32, b32 = r32.get_params()
    for x, y in zip(w + b, w32 + b32):
        assert np.array_equal(x.astype(np.float32).astype(np.float64), y)
    r.close()
    r32.close()


# ------------------------------------------------------------------------------------------------ resident data sets

@pytest.mark.parametrize("dtype", [1, 0], ids=["f64", "f32"])
def test_resident_set_training_flow_equals_the_oracle_loop(amd, oracle, dtype):
    """rcn_hip_load_data / rcn_hip_train_set_epoch / rcn_hip_evaluate_set -- what a host-language RCN::train calls (rust/rcn-hip,
    csrc/host/rcn.hpp, rcn.py: train_arrays) -- against the oracle's restatement of rcn.rs:126-167 on the same pictures, the same
    shuffles and the same initial parameters: scale_set ends up holding the TEST statistics (rcn.rs:134-137), chunks_exact drops the
    tail (300 samples, batch 32 -> 9 steps), the accepted counts per epoch agree."""
    B, epochs = 32, 3
    imgs, labels = synthetic_images(300, seed=3)
    timgs, tlabels = synthetic_images(120, seed=4)
    ws, bs = synthetic_params(BENCH_DIMS, seed=21)
    ws = [w * 0.05 for w in ws]
    r = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=dtype)
    r.set_params(ws, bs)
    assert r.load_set(0, imgs, labels) == 300 and r.load_set(1, timgs, tlabels) == 120
    # oracle side
    f, tf = oracle.features(imgs, DEFAULT_LAYERS), oracle.features(timgs, DEFAULT_LAYERS)
    X = oracle.standardize(f, *oracle.gen_scales(f))
    tm, ts = oracle.gen_scales(tf)
    TX = oracle.standardize(tf, tm, ts)
    Y, TY = one_hot(labels), one_hot(tlabels)
    np.testing.assert_allclose(r.scale_set, (tm, ts), rtol=1e-12)          # the TEST set's statistics
    rng = np.random.default_rng(99)
    rw, rb = ws, bs
    tol = 1e-10 if dtype == 1 else 2e-4
    for e in range(epochs):
        order = rng.permutation(300).astype(np.int32)
        loss = r.train_set_epoch(0, B, 3.0, perm=order, want_loss=True)
        assert loss.shape == (300 // B,)
        costs = []
        for j in range(300 // B):
            sel = order[j * B:(j + 1) * B]
            rw, rb, c = oracle.train_batch(rw, rb, X[sel], Y[sel], 3.0)
            costs.append(c)
        np.testing.assert_allclose(loss, costs, rtol=1e-9 if dtype == 1 else 2e-3)
        out = oracle.classify_test(rw, rb, TX)
        want = sum(oracle.eval_accept(out[i], TY[i]) for i in range(len(TX)))
        got = r.evaluate_set(1)
        assert got == want if dtype == 1 else abs(got - want) <= 2, (e, got, want)
    gw, gb = r.get_params()
    for a, b in zip(gw + gb, rw + rb):
        assert np.all(np.abs(a - b) <= tol * np.abs(b) + tol * 0.1)
    # a device-side shuffle instead of a host order: a permutation, deterministic per seed
    l1 = r.train_set_epoch(0, B, 0.0, seed=7, want_loss=True)
    l2 = r.train_set_epoch(0, B, 0.0, seed=7, want_loss=True)
    l3 = r.train_set_epoch(0, B, 0.0, seed=8, want_loss=True)
    assert np.array_equal(l1, l2) and not np.array_equal(l1, l3)
    # error behaviour: unloaded slot, labels beyond the classes, batch size 0
    r2 = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=dtype)
    r2.set_params(ws, bs)
    with pytest.raises(amd.RcnHipError):
        r2.evaluate_set(0)
    with pytest.raises(amd.RcnPanic):
        r2.load_set(0, imgs[:4], np.array([0, 1, 10, 2]))
    r2.load_set(0, imgs[:40], labels[:40])
    with pytest.raises(amd.RcnHipError):
        r2.train_set_epoch(0, 0, 3.0)
    assert r2.train_set_epoch(0, 64, 3.0, seed=1, want_loss=True).shape == (0,)      # fewer samples than one batch: no step (rcn.rs:147)
    r.close(); r2.close()


def test_train_arrays_runs_on_resident_sets_and_prints_the_reference_line(amd, oracle):
    imgs, labels = synthetic_images(200, seed=5)
    timgs, tlabels = synthetic_images(80, seed=6)
    r = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=amd.F64)
    lines = []
    acc = r.train_arrays(imgs, labels, timgs, tlabels, 10, 2, 3.0, rng=np.random.default_rng(1), log=lines.append)
    assert len(acc) == 2 and all(0 <= a <= 80 for a in acc)
    import re
    assert all(re.fullmatch(r"Epoch \d+: \d+/80 \[\d+\.\d\d%\]", l) for l in lines), lines      # rcn.rs:158-164
    r.close()


# ------------------------------------------------------------------------------------------------ resident one-XCD kernel

def _xcd_or_skip(d):
    import mercer_research_amd as amd
    try:
        d.set_dense_path(5)
    except amd.RcnHipError as e:
        pytest.skip(f"the resident one-XCD kernel does not apply on this device: {e}")


def test_resident_one_xcd_kernel_is_the_reference_loop(amd, oracle, monkeypatch):
    """dense path 5 (csrc/dense_xcd.hpp): ONE resident kernel per epoch segment, its 32 workgroups on one XCD, slab / deltas handed over
    through that XCD's L2.  Must be the reference's sequential train_batch loop (rcn.rs:147-149, 176-223): parameters and per-step costs
    against the oracle (f32 tolerances of the two-kernel pipeline) and against the two-kernel pipeline itself; shuffled and stored
    order, several launches per call (3 batches per segment of the epoch image), a second call continuing from the first, from u8
    pictures, and a call of ONE step (nothing to prefetch, no next forward)."""
    from mercer_research_amd.device import DeviceRCN
    monkeypatch.setenv("RCN_HIP_PACK_SEGMENT_BYTES", str(3 * 49 * 256 * 16 * 4))
    B, nb, N = 256, 8, 2304
    imgs, labels = synthetic_images(N, seed=41)
    ws, bs = synthetic_params(BENCH_DIMS, seed=14)
    ws = [w * 0.1 for w in ws]
    perm = np.random.default_rng(5).permutation(N).astype(np.int32)
    got = {}
    for path in (5, 2):
        d = DeviceRCN(dtype=0)
        if path == 5:
            _xcd_or_skip(d)
        else:
            d.set_dense_path(2)
        d.set_params(ws, bs)
        dev = d.to_device(imgs)
        d.gen_scales(d.features(dev))
        X = d.features(dev, standardize=True)
        Y = d.to_device(one_hot(labels), d.tdtype)
        permd = d.to_device(perm)
        loss = d.empty(nb)
        d.train_epoch(X, Y, permd, B, nb, 3.0, loss)                 # 3 + 3 + 2 steps: three launches of the resident kernel
        d.synchronize()                                              # (the copies below run on torch's stream, not the context's)
        l1 = loss.cpu().numpy().copy()
        d.train_epoch(X, Y, None, B, 2, 3.0, None)                   # stored order, continues from the first call
        d.train_epoch_images(dev, Y, permd, B, 4, 3.0, loss)         # straight from the pictures
        d.synchronize()
        l2 = loss.cpu().numpy()[:4].copy()
        d.train_epoch(X, Y, permd[5 * B:], B, 1, 3.0, loss)          # a call of one step
        d.synchronize()
        l3 = float(loss.cpu().numpy()[0])
        got[path] = (sum(d.get_params(), []), l1, l2, l3, X.double().cpu().numpy())
        d.rcn.close()
    # against the oracle's loop on the same (device-standardised) features
    Xh, Yh = got[5][4], one_hot(labels)
    rw, rb, c1, c2 = ws, bs, [], []
    for j in range(nb):
        sel = perm[j * B:(j + 1) * B]
        rw, rb, c = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
        c1.append(c)
    for j in range(2):
        rw, rb, _ = oracle.train_batch(rw, rb, Xh[j * B:(j + 1) * B], Yh[j * B:(j + 1) * B], 3.0)
    for j in range(4):
        sel = perm[j * B:(j + 1) * B]
        rw, rb, c = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
        c2.append(c)
    sel = perm[5 * B:6 * B]
    rw, rb, c3 = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
    for path in (5, 2):
        np.testing.assert_allclose(got[path][1], c1, rtol=1e-3)
        np.testing.assert_allclose(got[path][2], c2, rtol=2e-3)
        assert abs(got[path][3] - c3) <= 2e-3 * c3
        for a, b in zip(got[path][0], rw + rb):
            assert np.all(np.abs(a - b) <= 5e-4 * np.abs(b) + 5e-5), path   # 15 chained f32 steps
    for a, b in zip(got[5][0], got[2][0]):
        assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-5)               # the two forms differ only in summation grouping


def test_resident_one_xcd_kernel_first_step_tight_and_reproducible(amd, oracle):
    """One step from identical parameters holds the one-step f32 tolerance of SURVEY §8(c) (1e-5 relative + 1e-6), N(0,1) un-scaled
    parameters included, and two runs of the same epoch give the same bits (fixed summation orders; nothing depends on which
    workgroup arrives first)."""
    from mercer_research_amd.device import DeviceRCN
    B, N = 256, 1024
    rng = np.random.default_rng(2)
    X, Y = np.maximum(rng.standard_normal((N, 784)), 0.0), one_hot(rng.integers(0, 10, N))
    for wscale in (0.1, 1.0):
        ws, bs = synthetic_params(BENCH_DIMS, seed=8)
        ws = [w * wscale for w in ws]
        runs = []
        for rep in range(2):
            d = DeviceRCN(dtype=0)
            _xcd_or_skip(d)
            d.set_params(ws, bs)
            Xd, Yd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype)
            loss = d.empty(4)
            d.train_epoch(Xd, Yd, None, B, 1, 3.0, loss)
            p1 = sum(d.get_params(), [])
            d.train_epoch(Xd[B:], Yd[B:], None, B, 3, 3.0, loss[1:])
            d.synchronize()
            runs.append((p1, sum(d.get_params(), []), loss.cpu().numpy().copy()))
            d.rcn.close()
        for a, b in zip(runs[0][1], runs[1][1]):
            assert np.array_equal(a, b)
        assert np.array_equal(runs[0][2], runs[1][2])
        Xf = X.astype(np.float32).astype(np.float64)
        rw, rb, c = oracle.train_batch(ws, bs, Xf[:B], Y[:B], 3.0)
        tol = (1e-5, 1e-6) if wscale < 1 else (1e-4, 1e-5)          # saturated sigmoids (un-scaled init): as test_unscaled_n01_init_default_net
        for a, b in zip(runs[0][0], rw + rb):
            assert np.all(np.abs(a - b) <= tol[0] * np.abs(b) + tol[1])
        assert abs(runs[0][2][0] - c) <= 1e-5 * c if wscale < 1 else abs(runs[0][2][0] - c) <= 1e-4 * c


def test_resident_kernel_is_what_auto_selects_for_the_bench_shape_and_not_for_others(amd):
    from mercer_research_amd.device import DeviceRCN
    d = DeviceRCN(dtype=0)
    _xcd_or_skip(d)
    d.rcn.close()
    d64 = DeviceRCN(dtype=1)
    d64.set_dense_path(5)                                            # f64 context (round 4): the kernel's f64 instantiations, batches of 1..128
    assert d64.train_epoch_resident(128) and d64.train_epoch_resident(256) and not d64.train_epoch_resident(257)
    d64.rcn.close()
    d3 = DeviceRCN(dtype=0, feedforward_cfg=[10, 10])
    d3.set_dense_path(5)                                             # two hidden layers <= 32, <= 16: the kernel's second instantiation
    d3.rcn.close()
    for cfg in ([10, 10, 10], [40], [30, 20]):                       # three hidden layers; hidden > 32; second hidden > 16
        dn = DeviceRCN(dtype=0, feedforward_cfg=cfg)
        with pytest.raises(amd.RcnHipError):
            dn.set_dense_path(5)
        dn.rcn.close()


def test_resident_kernel_data_parallel_form_at_world_one_is_the_single_gpu_kernel_bit_for_bit(amd, monkeypatch):
    """The DP instantiation of the resident kernel (k_xcd_epoch<true>: publish / gather between the gradient MFMAs and the update)
    with a group of ONE rank: the gathered sum is the rank's own partial, so parameters and per-step costs must equal the
    single-GPU instantiation bit for bit -- every line of the exchange except the peer polls runs (bootstrap over RCCL forced on,
    as in test_peer_allreduce_bootstrap_over_rccl_world1), several launches per call, and a second call carries the sequence on."""
    from mercer_research_amd.device import DeviceRCN
    monkeypatch.setenv("RCN_HIP_PACK_SEGMENT_BYTES", str(3 * 49 * 256 * 16 * 4))
    monkeypatch.setenv("RCN_HIP_DP_P2P", "2")
    B, nb, N = 256, 7, 2048
    rng = np.random.default_rng(12)
    X = np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32)
    Y = one_hot(rng.integers(0, 10, N)).astype(np.float32)
    ws, bs = synthetic_params(BENCH_DIMS, seed=3)
    ws = [w * 0.1 for w in ws]
    perm = np.random.default_rng(6).permutation(N).astype(np.int32)
    got = {}
    for form in ("dp", "single"):
        d = DeviceRCN(dtype=0)
        _xcd_or_skip(d)
        d.set_params(ws, bs)
        Xd, Yd, permd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(perm)
        loss, loss2 = d.empty(nb), d.empty(2)
        if form == "dp":
            try:
                assert d.dp_init() == (0, 1)
            except amd.RcnHipError as e:
                pytest.skip(f"no RCCL bootstrap on this box: {e}")
            assert d.dp_p2p_mode() == 2 and d.dp_resident(B)
            d.dp_train_epoch(Xd, Yd, permd, B, nb, 3.0, loss)
            d.dp_train_epoch(Xd, Yd, None, B, 2, 3.0, loss2)
        else:
            d.train_epoch(Xd, Yd, permd, B, nb, 3.0, loss)
            d.train_epoch(Xd, Yd, None, B, 2, 3.0, loss2)
        d.synchronize()
        got[form] = (sum(d.get_params(), []), loss.cpu().numpy().copy(), loss2.cpu().numpy().copy())
        if form == "dp":
            d.dp_finalize()
        d.rcn.close()
    assert np.array_equal(got["dp"][1], got["single"][1]) and np.array_equal(got["dp"][2], got["single"][2]), (got["dp"][1] - got["single"][1], got["dp"][2] - got["single"][2])
    for a, b in zip(got["dp"][0], got["single"][0]):
        assert np.array_equal(a, b), float(np.abs(a - b).max())


_GATHER_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import mercer_research_amd as amd
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
case = np.load(sys.argv[2])
d = DeviceRCN(dtype=0)
try:
    d.set_dense_path(5)
except amd.RcnHipError:
    np.savez(sys.argv[3], skipped=1); sys.exit(0)
ws, bs = synthetic_params([784, 30, 10], seed=8)
d.set_params([w * 0.1 for w in ws], bs)
X, Y, perm = d.to_device(case["X"], d.tdtype), d.to_device(case["Y"], d.tdtype), d.to_device(case["perm"])
B, nb = 256, int(case["nb"])
loss, loss2 = d.empty(nb), d.empty(3)
d.train_epoch(X, Y, perm, B, nb, 3.0, loss)          # shuffled order
d.train_epoch(X, Y, None, B, 3, 3.0, loss2)          # stored order, continues
d.train_epoch(X, Y, perm[B:], B, 1, 3.0, None)       # a call of one step
d.synchronize()
gw, gb = d.get_params()
np.savez(sys.argv[3], skipped=0, gathers=int(d.train_epoch_gathers(B)), loss=loss.cpu().numpy(), loss2=loss2.cpu().numpy(),
         **{f"p{i}": p for i, p in enumerate(gw + gb)})
"""


def test_resident_kernel_gather_form_equals_the_packed_image_bit_for_bit(amd, tmp_path):
    """RCN_HIP_XCD_GATHER=1 (opt-in; measured slower, csrc/rcn_hip_api_xcd.ipp): the resident kernel fetches every batch's rows itself --
    no k_pack_epoch, one launch per call.  Same loads, same arithmetic: parameters and per-step costs must equal the default
    (packed image) form bit for bit, in shuffled and stored order and for a one-step call.  The form is chosen per process
    (environment), hence two child processes."""
    rng = np.random.default_rng(77)
    N, nb = 2048, 7
    np.savez(tmp_path / "case.npz", X=np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32),
             Y=one_hot(rng.integers(0, 10, N)).astype(np.float32), perm=rng.permutation(N).astype(np.int32), nb=nb)
    outs = {}
    for g in ("0", "1"):
        out = tmp_path / f"out{g}.npz"
        pr = subprocess.run([sys.executable, "-c", _GATHER_SCRIPT, ROOT, str(tmp_path / "case.npz"), str(out)],
                            env=dict(os.environ, RCN_HIP_XCD_GATHER=g), capture_output=True, timeout=240)
        assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
        outs[g] = np.load(out)
        if int(outs[g]["skipped"]):
            pytest.skip("the resident one-XCD kernel does not apply on this device")
    assert int(outs["0"]["gathers"]) == 0 and int(outs["1"]["gathers"]) == 1
    for k in outs["0"].files:
        if k not in ("gathers", "skipped"):
            assert np.array_equal(outs["0"][k], outs["1"][k]), k


@pytest.mark.parametrize("hidden", [[10, 10], [30, 12], [32, 16], [7, 3]], ids=["reference-test-net-784-10-10-10", "784-30-12-10", "784-32-16-10", "784-7-3-10"])
def test_resident_kernel_with_two_hidden_layers_is_the_reference_loop(amd, oracle, monkeypatch, hidden):
    """k_xcd_epoch<false, true> (csrc/dense_xcd.hpp): the resident kernel for TWO hidden layers -- the reference's own test net is
    784-10-10-10 (rcn.rs:558,577).  The sample group runs one more small layer forward and backward, the tail tiles cover [W_2 | b_2];
    against the oracle's sequential train_batch (f32 tolerances of the resident kernel) and against the generic two-kernel pipeline:
    shuffled and stored order, several launches per call, a call of one step, per-step costs."""
    from mercer_research_amd.device import DeviceRCN
    monkeypatch.setenv("RCN_HIP_PACK_SEGMENT_BYTES", str(3 * 49 * 256 * 16 * 4))
    dims = [784] + hidden + [10]
    B, nb, N = 256, 7, 2048
    rng = np.random.default_rng(21)
    X = np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32)
    labels = rng.integers(0, 10, N)
    Y = one_hot(labels).astype(np.float32)
    ws, bs = synthetic_params(dims, seed=33)
    ws = [w * 0.1 for w in ws]
    perm = np.random.default_rng(9).permutation(N).astype(np.int32)
    got = {}
    for path in (5, 2):
        d = DeviceRCN(dtype=0, feedforward_cfg=hidden)
        if path == 5:
            _xcd_or_skip(d)
        else:
            d.set_dense_path(2)
        d.set_params(ws, bs)
        Xd, Yd, permd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(perm)
        loss, loss1 = d.empty(nb), d.empty(1)
        d.train_epoch(Xd, Yd, permd, B, nb, 3.0, loss)               # 3 + 3 + 1 steps: three launches
        d.train_epoch(Xd, Yd, None, B, 2, 3.0, None)                 # stored order, continues
        d.train_epoch(Xd, Yd, permd[5 * B:], B, 1, 3.0, loss1)       # a call of one step
        d.synchronize()
        got[path] = (sum(d.get_params(), []), loss.cpu().numpy().copy(), float(loss1.cpu().numpy()[0]))
        d.rcn.close()
    Xh, Yh = X.astype(np.float64), Y.astype(np.float64)
    rw, rb, c1 = ws, bs, []
    for j in range(nb):
        sel = perm[j * B:(j + 1) * B]
        rw, rb, c = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
        c1.append(c)
    for j in range(2):
        rw, rb, _ = oracle.train_batch(rw, rb, Xh[j * B:(j + 1) * B], Yh[j * B:(j + 1) * B], 3.0)
    sel = perm[5 * B:6 * B]
    rw, rb, c3 = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
    for path in (5, 2):
        np.testing.assert_allclose(got[path][1], c1, rtol=1e-3)
        assert abs(got[path][2] - c3) <= 2e-3 * c3
        for a, b in zip(got[path][0], rw + rb):
            assert np.all(np.abs(a - b) <= 5e-4 * np.abs(b) + 5e-5), path   # 10 chained f32 steps
    for a, b in zip(got[5][0], got[2][0]):
        assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-5)


def test_resident_kernel_is_what_auto_selects_for_the_reference_test_net(amd):
    """784-10-10-10 at B = 256 in the f32 context lands on the resident kernel by default (rcn_hip_time_kernels_dev reports no
    first / second kernel), and the step is faster than the generic two-kernel pipeline it replaces there."""
    from mercer_research_amd.device import DeviceRCN
    import torch
    res = {}
    for path in (0, 2):
        d = DeviceRCN(dtype=0, feedforward_cfg=[10, 10])
        ws, bs = synthetic_params([784, 10, 10, 10], seed=5)
        d.set_params(ws, bs)
        if path:
            d.set_dense_path(path)
        with torch.cuda.stream(d.stream):
            X = torch.rand(4096, 784, device=d.device)
            Y = torch.zeros(4096, 10, device=d.device); Y[:, 1] = 1
        d.synchronize()
        for _ in range(2):
            d.train_epoch(X, Y, None, 256, 16, 3.0, None)
        d.synchronize()
        res[path] = d.time_kernels(X[:256], Y[:256], reps=128)
        d.rcn.close()
    if res[0][0] != 0.0:
        pytest.skip("the resident one-XCD kernel does not apply on this device")
    assert res[0][2] < res[2][2], res


def test_resident_set_flow_of_the_reference_test_net_at_batch_256(amd, oracle):
    """RCN::train's flow (rcn.rs:126-167: load both sets, shuffle, chunks_exact, train_batch, classify_test) for the reference's own
    test net -- two hidden layers of ten (rcn.rs:558,577) -- at the batch size of BASELINE's metric, f32: the steps run on the resident
    kernel's two-hidden-layer instantiation (where the device has it) and must follow the oracle's restatement on the same pictures,
    shuffles and initial parameters: per-step costs, final parameters, accepted counts."""
    B, epochs, N = 256, 2, 600
    dims = [784, 10, 10, 10]
    imgs, labels = synthetic_images(N, seed=13)
    timgs, tlabels = synthetic_images(160, seed=14)
    ws, bs = synthetic_params(dims, seed=22)
    ws = [w * 0.05 for w in ws]
    r = amd.RCN(10, amd.default_convpool(), [10, 10], input_shape=(28, 28), dtype=0)
    r.set_params(ws, bs)
    assert r.load_set(0, imgs, labels) == N and r.load_set(1, timgs, tlabels) == 160
    f, tf = oracle.features(imgs, DEFAULT_LAYERS), oracle.features(timgs, DEFAULT_LAYERS)
    tm, ts = oracle.gen_scales(tf)
    X = oracle.standardize(f, *oracle.gen_scales(f))
    TX = oracle.standardize(tf, tm, ts)
    Y, TY = one_hot(labels), one_hot(tlabels)
    rng = np.random.default_rng(5)
    rw, rb = ws, bs
    for e in range(epochs):
        order = rng.permutation(N).astype(np.int32)
        loss = r.train_set_epoch(0, B, 3.0, perm=order, want_loss=True)
        assert loss.shape == (N // B,)                                   # chunks_exact: 600 -> two batches, the tail is dropped
        costs = []
        for j in range(N // B):
            sel = order[j * B:(j + 1) * B]
            rw, rb, c = oracle.train_batch(rw, rb, X[sel], Y[sel], 3.0)
            costs.append(c)
        np.testing.assert_allclose(loss, costs, rtol=2e-3)
        out = oracle.classify_test(rw, rb, TX)
        want = sum(oracle.eval_accept(out[i], TY[i]) for i in range(len(TX)))
        assert abs(r.evaluate_set(1) - want) <= 2, e
    gw, gb = r.get_params()
    for a, b in zip(gw + gb, rw + rb):
        assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-5)
    r.close()


def test_resident_kernel_many_segments_of_the_epoch_image(amd, oracle, monkeypatch):
    """A call of many segments of the epoch image (2 batches each here, 17 batches: nine launches): the halves of the image are reused
    every second segment -- a pack that ran over a half still being read, or steps on a half not yet packed, would train on the
    wrong batches.  Per-step costs and parameters against the oracle's sequential loop, twice in a row, then once more from u8
    pictures (the fused feature + pack kernel) for the same costs bit for bit."""
    from mercer_research_amd.device import DeviceRCN
    monkeypatch.setenv("RCN_HIP_PACK_SEGMENT_BYTES", str(2 * 49 * 256 * 16 * 4))
    B, nb, N = 256, 17, 4608
    imgs, labels = synthetic_images(N, seed=61)
    ws, bs = synthetic_params(BENCH_DIMS, seed=15)
    ws = [w * 0.1 for w in ws]
    perm = np.random.default_rng(8).permutation(N).astype(np.int32)
    d = DeviceRCN(dtype=0)
    _xcd_or_skip(d)
    d.set_dense_path(0)
    d.set_params(ws, bs)
    dev = d.to_device(imgs)
    d.gen_scales(d.features(dev))
    X = d.features(dev, standardize=True)
    Y = d.to_device(one_hot(labels), d.tdtype)
    permd = d.to_device(perm)
    loss, loss2, loss3 = d.empty(nb), d.empty(nb), d.empty(nb)
    d.train_epoch(X, Y, permd, B, nb, 3.0, loss)
    d.train_epoch(X, Y, None, B, nb, 3.0, loss2)
    d.synchronize()
    p_after_two = sum(d.get_params(), [])
    d.set_params(ws, bs)
    d.train_epoch_images(dev, Y, permd, B, nb, 3.0, loss3)
    d.synchronize()
    Xh, Yh = X.double().cpu().numpy(), one_hot(labels)
    rw, rb, c1, c2 = ws, bs, [], []
    for j in range(nb):
        sel = perm[j * B:(j + 1) * B]
        rw, rb, c = oracle.train_batch(rw, rb, Xh[sel], Yh[sel], 3.0)
        c1.append(c)
    for j in range(nb):
        rw, rb, c = oracle.train_batch(rw, rb, Xh[j * B:(j + 1) * B], Yh[j * B:(j + 1) * B], 3.0)
        c2.append(c)
    np.testing.assert_allclose(loss.cpu().numpy(), c1, rtol=2e-3)
    np.testing.assert_allclose(loss2.cpu().numpy(), c2, rtol=5e-3)
    for a, b in zip(p_after_two, rw + rb):
        assert np.all(np.abs(a - b) <= 1e-3 * np.abs(b) + 1e-4)            # 34 chained f32 steps
    assert np.array_equal(loss3.cpu().numpy(), loss.cpu().numpy())          # the in-line pack from pictures: the same batches, bit for bit
    d.rcn.close()
