"""GPU tests added in round 4 (run with -m gpu on an MI355X), all through the C ABI:

* the resident one-XCD kernel in the reference's OWN arithmetic type (f64: rcn.rs:28,31,49) at its own operating points -- batch_size
  10 (rcn/src/main.rs:36-37, rcn.rs:581), 32 (BASELINE configs[0]) -- and at every other batch of 1..256 samples, both reference nets,
  against the CPU restatement at the f64 tolerance (1e-11 per step);
* the 256-step f64 loss curve at B = 10 and B = 32 on the resident kernel (<= 1e-9 per step);
* the self-healing step-down in f64.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.rcn_oracle import DEFAULT_LAYERS, one_hot, synthetic_images, synthetic_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32, F64 = 0, 1


@pytest.fixture(scope="module")
def amd():
    import mercer_research_amd as m
    return m


def _xcd_or_skip(d):
    import mercer_research_amd as amd
    try:
        d.set_dense_path(5)
    except amd.RcnHipError as e:
        pytest.skip(f"the resident one-XCD kernel does not apply on this device: {e}")


def _oracle_steps(oracle, ws, bs, X, Y, perm, B, nb, eta):
    rw, rb, costs = ws, bs, []
    for j in range(nb):
        sel = perm[j * B:(j + 1) * B]
        rw, rb, c = oracle.train_batch(rw, rb, X[sel], Y[sel], eta)
        costs.append(c)
    return rw, rb, np.array(costs)


def test_f64_context_reaches_the_resident_kernel_by_default(amd):
    """The drop-in crate opens an F64 context (rust/rcn-hip/src/rcn.rs) and trains at batch_size 10: that context, with no option set,
    must land on the resident kernel for batches of 1..256 and on the pipeline above."""
    from mercer_research_amd.device import DeviceRCN
    d = DeviceRCN(dtype=F64)
    _xcd_or_skip(d)
    d.set_dense_path(0)
    for B in (1, 10, 32, 64, 100, 128, 129, 200, 256):
        assert d.train_epoch_resident(B), B
    for B in (257, 512):
        assert not d.train_epoch_resident(B), B
    d.rcn.close()
    d = DeviceRCN(dtype=F32)
    for B in (1, 10, 32, 128, 200, 256):
        assert d.train_epoch_resident(B), B
    assert not d.train_epoch_resident(257)
    d.rcn.close()
    d = DeviceRCN(dtype=F64, feedforward_cfg=[40])                       # outside the kernel's shape class
    assert not d.train_epoch_resident(10)
    d.rcn.close()


@pytest.mark.parametrize("hidden", [[30], [10, 10]], ids=["784-30-10", "reference-test-net-784-10-10-10"])
@pytest.mark.parametrize("B", [10, 32, 1, 8, 33, 64, 100, 128, 200, 255, 256])
def test_resident_kernel_f64_at_any_batch_of_1_to_256_is_the_reference_loop(amd, oracle, hidden, B):
    """train_batch (rcn.rs:176-223, 260-314; sigmoid rcn.rs:478-492) in f64 on the resident kernel's v_mfma_f64_16x16x4_f64
    instantiations for 32 / 64 / 128 / 256 samples (the last with ONE batch buffer and delta_1 staged in two halves: LDS): one step from identical parameters within 1e-11 of the CPU restatement, six chained steps
    in a shuffled order within 1e-10, and the same calls on the sample-tile kernels (dense path 1)."""
    from mercer_research_amd.device import DeviceRCN
    dims = [784] + hidden + [10]
    N, nb = 2048, 6
    rng = np.random.default_rng(300 + B)
    X = np.maximum(rng.standard_normal((N, 784)), 0.0)
    Y = one_hot(rng.integers(0, 10, N))
    ws, bs = synthetic_params(dims, seed=21)
    ws = [w * 0.1 for w in ws]
    perm = rng.permutation(N).astype(np.int32)
    got = {}
    for path in (5, 1):
        d = DeviceRCN(dtype=F64, feedforward_cfg=hidden)
        if path == 5:
            _xcd_or_skip(d)
            assert d.train_epoch_resident(B)
        else:
            d.set_dense_path(1)
            assert not d.train_epoch_resident(B)
        d.set_params(ws, bs)
        Xd, Yd, pd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(perm)
        loss = d.empty(nb + 1)
        d.train_epoch(Xd, Yd, None, B, 1, 3.0, loss)                   # one step, stored order
        p1 = sum(d.get_params(), [])
        d.train_epoch(Xd, Yd, pd, B, nb, 3.0, loss[1:])                 # six more, shuffled
        d.synchronize()
        got[path] = (p1, sum(d.get_params(), []), loss.cpu().numpy().astype(np.float64))
        assert d.fallbacks_taken() == 0
        d.rcn.close()
    rw, rb, c0 = oracle.train_batch(ws, bs, X[:B], Y[:B], 3.0)
    for path in (5, 1):
        for a, b in zip(got[path][0], rw + rb):
            assert np.all(np.abs(a - b) <= 1e-11 * np.abs(b) + 1e-12), (path, float(np.abs(a - b).max()))
        assert abs(got[path][2][0] - c0) <= 1e-11 * c0
    rw, rb, cs = _oracle_steps(oracle, rw, rb, X, Y, perm, B, nb, 3.0)
    for path in (5, 1):
        np.testing.assert_allclose(got[path][2][1:], cs, rtol=1e-10)
        for a, b in zip(got[path][1], rw + rb):
            assert np.all(np.abs(a - b) <= 1e-10 * np.abs(b) + 1e-11), (path, float(np.abs(a - b).max()))


@pytest.mark.parametrize("B", [10, 32], ids=["B10-reference-default", "B32-baseline-config0"])
def test_f64_loss_curve_256_steps_on_the_resident_kernel(amd, oracle, B):
    """256 consecutive steps in the reference's own type at its own batch sizes, from the resident u8 pictures through load_data, on the
    resident kernel against oracle/rcn_oracle.c on the identical batches: the cost within 1e-9 relative at EVERY step, the final
    parameters within 1e-8 (the f64 bar of DESIGN section 5)."""
    import torch
    from mercer_research_amd.device import DeviceRCN
    N, steps, eta = 8192, 256, 3.0
    imgs, labels = synthetic_images(N, seed=77)
    feats = oracle.features(imgs, DEFAULT_LAYERS)
    m, s = oracle.gen_scales(feats)
    X, Y = oracle.standardize(feats, m, s), one_hot(labels)
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    d = DeviceRCN(dtype=F64)
    _xcd_or_skip(d)
    d.set_dense_path(0)                                                 # the default selection, as the drop-in crate gets it
    assert d.train_epoch_resident(B)
    d.set_params(ws, bs)
    Xd, Yd = d.load_data(d.to_device(imgs), d.to_device(labels))
    perm = torch.empty(N, dtype=torch.int32, device=d.device)
    loss = d.empty(steps)
    d.shuffle(perm, N, 1, seed=0xABCD)
    d.synchronize()
    p = perm.cpu().numpy().astype(np.int64)
    d.epoch_begin(Xd, Yd, perm, B, steps)
    d.epoch_steps(0, 100, eta, loss)
    d.epoch_steps(100, steps - 100, eta, loss[100:])
    gw, gb = d.get_params()
    gpu = loss.double().cpu().numpy()
    assert d.fallbacks_taken() == 0
    d.rcn.close()
    h = oracle.net(ws, bs)
    cpu = np.zeros(steps)
    for j in range(steps):
        idx = p[j * B:(j + 1) * B]
        xb, yb = np.ascontiguousarray(X[idx]), np.ascontiguousarray(Y[idx])
        cpu[j] = oracle.lib.rcn_o_train_batch(C.byref(h.net), xb.ctypes.data_as(C.POINTER(C.c_double)), yb.ctypes.data_as(C.POINTER(C.c_double)), B, eta)
    rel = np.abs(gpu - cpu) / np.abs(cpu)
    assert rel.max() <= 1e-9, (rel.max(), int(rel.argmax()))
    for a, b in zip(gw + gb, h.weights() + h.biases()):
        assert np.all(np.abs(a - b) <= 1e-8 * np.abs(b) + 1e-9), float(np.abs(a - b).max())


@pytest.mark.steps_down
@pytest.mark.parametrize("fail_launch", [1, 2])
def test_f64_resident_kernel_steps_down_by_itself(amd, oracle, fail_launch):
    """The forced expiry of tests/test_gpu_round3.py in the f64 context at B = 32: the library re-runs what the failed launch had not applied
    on the two-kernel pipeline; the epochs equal the oracle at the f64 tolerance and the counter reads 1."""
    import torch
    from mercer_research_amd.device import DeviceRCN
    B, nb, N = 32, 8, 512
    imgs, labels = synthetic_images(N, seed=9)
    ws, bs = synthetic_params([784, 30, 10], seed=3)
    ws = [w * 0.1 for w in ws]
    d = DeviceRCN(dtype=F64)
    _xcd_or_skip(d)
    d.set_dense_path(0)
    d.set_option("xcd_timeout_ticks", 300000)
    d.set_option("xcd_fault_launch", fail_launch)
    d.set_params(ws, bs)
    Xd, Yd = d.load_data(d.to_device(imgs), d.to_device(labels))
    Xh, Yh = Xd.double().cpu().numpy(), one_hot(labels)
    perm = torch.empty(N, dtype=torch.int32, device=d.device)
    loss = d.empty(2 * nb)
    perms = []
    for e in range(2):
        d.shuffle(perm, N, 1, seed=900 + e)
        d.synchronize()
        perms.append(perm.cpu().numpy().astype(np.int64))
    assert d.fallbacks_taken() == 0
    for e in range(2):                                                  # nothing synchronises in here
        d.shuffle(perm, N, 1, seed=900 + e)
        d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss[e * nb:])
    d.synchronize()                                                     # heals
    assert d.fallbacks_taken() == 1
    assert not d.train_epoch_resident(B)
    got = sum(d.get_params(), [])
    costs = loss.double().cpu().numpy()
    rw, rb, cs = ws, bs, []
    for e in range(2):
        rw, rb, c = _oracle_steps(oracle, rw, rb, Xh, Yh, perms[e], B, nb, 3.0)
        cs.extend(c)
    np.testing.assert_allclose(costs, cs, rtol=1e-9)
    for a, b in zip(got, rw + rb):
        assert np.all(np.abs(a - b) <= 1e-9 * np.abs(b) + 1e-10), float(np.abs(a - b).max())
    d.rcn.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# an expired wait says where; the closing round is one decision (torn write-back)

def _faulty_context(amd, mode, auto, B=256, N=1024, seed=3, dtype=F32):
    from mercer_research_amd.device import DeviceRCN
    d = DeviceRCN(dtype=dtype)
    _xcd_or_skip(d)
    d.set_dense_path(0)
    d.set_option("xcd_timeout_ticks", 300000)                          # 3 ms per expired wait
    d.set_option("xcd_auto_fallback", auto)
    d.set_option("xcd_fault_mode", mode)
    d.set_option("xcd_fault_launch", 1)
    ws, bs = synthetic_params([784, 30, 10], seed=seed)
    ws = [w * 0.1 for w in ws]
    d.set_params(ws, bs)
    return d, ws, bs


def test_an_expired_wait_names_its_site_worker_and_producer(amd):
    """The first worker that gives up writes ONE record (site, worker, step, launch, who was missing) and the host adds the workspace's
    tables: a launch whose worker 1 never becomes resident (test hook) fails at step 0 on a wait for producer 1 -- the slab flag a
    sample group polls or the delta flag a feature worker / tail tile polls, whichever clock runs out first -- and says so."""
    B, nb, N = 256, 2, 512
    d, ws, bs = _faulty_context(amd, 0, 0)
    rng = np.random.default_rng(0)
    X, Y = np.maximum(rng.standard_normal((N, 784)), 0.0), one_hot(rng.integers(0, 10, N))
    Xd, Yd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype)
    assert d.last_timeout() is None
    d.train_epoch(Xd, Yd, None, B, nb, 3.0, None)
    with pytest.raises(amd.RcnHipError) as ei:
        d.synchronize()
    rec = d.last_timeout()
    assert rec is not None and rec["code"] == 1
    assert rec["site"] in (3, 4), rec                                   # slab flag or delta flag
    assert rec["missing"] == 0b10, rec                                  # producer 1 (feature worker 1 / sample group 1) never delivered
    assert rec["step"] == 0 and rec["worker"] != 1 and rec["workers"] == 32 and rec["world"] == 1 and rec["xsel"] == 0
    msg = str(ei.value)
    assert "first expired wait: site %d" % rec["site"] in msg and "producers still behind 0x2" in msg, msg
    assert "xcc =" in msg and "flagA =" in msg and "flagB =" in msg and "decision word" in msg, msg
    xcc = msg.split("xcc =")[1].split(";")[0].split()
    assert len(xcc) == 32 and len(set(xcc)) == 1, xcc                  # the table shows all 32 workers on one XCC
    d.rcn.close()


@pytest.mark.steps_down
def test_a_worker_that_reaches_the_closing_round_late_cannot_tear_the_parameters(amd, oracle):
    """every worker used to wait for all closing flags under its OWN clock, so a worker that arrived after another had
    given up saw all flags set and wrote its slice -- a torn parameter vector under a launch the host then took for complete.  The
    closing round is now one decision on one word (arrivals counted, poisoned by the first worker that gives up): with worker 1 held
    back past everyone's time-out (test hook, mode 1) NO worker writes, the record names the closing round with 31 of 32 arrivals,
    and the healed epochs equal the oracle."""
    import torch
    B, nb, N = 256, 3, 1024
    d, ws, bs = _faulty_context(amd, 1, 0)
    imgs, labels = synthetic_images(N, seed=9)
    Xd, Yd = d.load_data(d.to_device(imgs), d.to_device(labels))
    Xh, Yh = Xd.double().cpu().numpy(), one_hot(labels)
    perm = torch.empty(N, dtype=torch.int32, device=d.device)
    d.shuffle(perm, N, 1, seed=41)
    d.synchronize()
    p = perm.cpu().numpy().astype(np.int64)
    loss = d.empty(nb)
    d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss)
    with pytest.raises(amd.RcnHipError) as ei:
        d.synchronize()
    rec = d.last_timeout()
    assert rec["site"] == 9 and rec["missing"] == 31 and rec["workers"] == 32 and rec["worker"] != 1 and rec["step"] == nb - 1, rec
    assert "closing round" in str(ei.value) and "arrivals seen 31 of 32" in str(ei.value)
    d.set_dense_path(2)                                                 # the recovery the message names: clears the word
    d.synchronize()
    got = sum(d.get_params(), [])
    for a, b in zip(got, ws + bs):
        assert np.array_equal(a, b.astype(np.float32).astype(np.float64)), "a worker wrote its slice although the launch was not committed"
    d.rcn.close()
    # the same fault with the self-healing step-down on: the epoch equals the oracle, one step-down
    d, ws, bs = _faulty_context(amd, 1, 1)
    Xd, Yd = d.load_data(d.to_device(imgs), d.to_device(labels))
    perm = torch.empty(N, dtype=torch.int32, device=d.device)
    d.shuffle(perm, N, 1, seed=41)
    d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss)
    d.synchronize()
    assert d.fallbacks_taken() == 1 and d.last_timeout()["site"] == 9
    rw, rb, cs = _oracle_steps(oracle, ws, bs, Xh, Yh, p, B, nb, 3.0)
    np.testing.assert_allclose(loss.double().cpu().numpy(), cs, rtol=2e-3)
    for a, b in zip(sum(d.get_params(), []), rw + rb):
        assert np.all(np.abs(a - b) <= 5e-4 * np.abs(b) + 5e-5), float(np.abs(a - b).max())
    d.rcn.close()


@pytest.mark.steps_down
def test_steps_on_index_rows_the_caller_wrote_are_not_replayed_silently(amd, oracle):
    """the step-down journals raw pointers; index rows the CALLER wrote may legally have been overwritten in stream order
    since the call.  Such steps are not re-run silently: the error stays and says why; rows from rcn_hip_shuffle_dev (re-drawn from
    their seed) are re-run as before, and option xcd_replay_caller_rows = 1 opts in."""
    B, nb, N = 256, 2, 512
    rng = np.random.default_rng(5)
    X, Y = np.maximum(rng.standard_normal((N, 784)), 0.0), one_hot(rng.integers(0, 10, N))
    order = rng.permutation(N).astype(np.int32)
    d, ws, bs = _faulty_context(amd, 0, 1)
    Xd, Yd, pd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(order)
    d.train_epoch(Xd, Yd, pd, B, nb, 3.0, None)
    with pytest.raises(amd.RcnHipError) as ei:
        d.synchronize()
    assert "index rows the caller wrote" in str(ei.value) and d.fallbacks_taken() == 0
    d.rcn.close()
    d, ws, bs = _faulty_context(amd, 0, 1)
    d.set_option("xcd_replay_caller_rows", 1)
    Xd, Yd, pd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(order)
    d.train_epoch(Xd, Yd, pd, B, nb, 3.0, None)
    d.synchronize()
    assert d.fallbacks_taken() == 1
    rw, rb, _ = _oracle_steps(oracle, ws, bs, X.astype(np.float32).astype(np.float64), Y, order.astype(np.int64), B, nb, 3.0)
    for a, b in zip(sum(d.get_params(), []), rw + rb):
        assert np.all(np.abs(a - b) <= 5e-4 * np.abs(b) + 5e-5), float(np.abs(a - b).max())
    d.rcn.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# the resident data-parallel form at the BENCH's shard size between processes

@pytest.mark.parametrize("world,Bs", [(2, 256), (2, 128)], ids=["2-ranks-shard-256-the-bench-shard", "2-ranks-shard-128"])
def test_resident_data_parallel_form_at_the_bench_shard_between_processes(amd, oracle, tmp_path, world, Bs):
    """bench.py --gpus N runs 256 images per rank: k_xcd_epoch<float, 256, true, true> -- 32 workers, 25 owners of slice pairs, 148 KB of
    LDS each.  Between processes on ONE device that shard is opt-in (RCN_TEST_DP_SHARD256_ON_ONE_GPU=1): a rank's 32 workers fill their
    XCD, the peer's launch has 32 blocks bound for that XCD which cannot be placed, and the dispatcher does not run past them -- the two
    kernels run one after the other and every wait expires.  Measured, not guessed (profiles/r4_dp_shard256_two_process_probe.txt: 10 of
    20 runs expired; the two ranks fail exactly one time-out apart and the early one finds the late one's words in its memory right after
    giving up).  What the case would add over the shards that DO run here is little: the exchange's indices (woff0, wvalid, the tail
    tiles' tp, the cost's P) do not depend on the batch instantiation -- shards of 128 (default-on here), 64 and 32 push the same words.
    Two processes, rank r's workers on PHYSICAL XCD r: replicas bit-identical; two epochs equal the oracle on the concatenated batches."""
    if Bs == 256 and os.environ.get("RCN_TEST_DP_SHARD256_ON_ONE_GPU") != "1":
        pytest.skip("two resident kernels of 32 workers each on ONE device serialise (profiles/r4_dp_shard256_two_process_probe.txt); RCN_TEST_DP_SHARD256_ON_ONE_GPU=1 runs it")
    from test_gpu_round3 import _oracle_global_epochs, _spawn_ranks
    nb, dims = 3, (784, 30, 10)
    env = {"RCN_HIP_XCD_TIMEOUT_TICKS": "400000000", "RCN_HIP_DP_TIMEOUT_TICKS": "400000000"}
    Xs, Ys, logs = _spawn_ranks(tmp_path, world, 0, "default", env, dims=dims, Bs=Bs, nb=nb)
    outs = [np.load(tmp_path / f"out{r}.npz") for r in range(world)]
    for o in outs:
        assert int(o["bad"]) == 0 and int(o["timed_out"]) == 0 and int(o["active"]) == 2, logs
        assert int(o["resident"]) == 1, "the data-parallel epoch did not run on the resident kernel"
    for r in range(1, world):
        for k in ("w0", "w1", "b0", "b1", "loss"):
            assert np.array_equal(outs[0][k], outs[r][k]), (r, k)
    rw, rb, costs = _oracle_global_epochs(oracle, list(dims), Xs, Ys, Bs, nb)
    for a, b in zip([outs[0]["w0"], outs[0]["w1"], outs[0]["b0"], outs[0]["b1"]], [rw[0], rw[1], rb[0], rb[1]]):
        assert np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-5), float(np.abs(a - b).max())
    np.testing.assert_allclose(outs[0]["loss"], costs, rtol=1e-4)
    if Bs in (128, 256):
        # the phase clocks of a diagnostic launch (rcn_hip_dp_phase_us): owners and members waited a finite, positive time per step, the
        # launch ran nb steps, and with two ranks both roles exist on each (worker w's owner is rank w % 2)
        for o in outs:
            ph = o["phase"]
            assert ph[7] == nb and np.all(np.isfinite(ph)) and ph[0] > 0 and ph[2] > 0 and ph[4] > 0 and ph[6] > ph[1] * 0.5, ph


# ---------------------------------------------------------------------------------------------------------------------------------
# Track X: kernel-selection knobs are per net

def test_trackx_options_are_per_net_and_the_plan_of_a_net_is_its_own(monkeypatch):
    """RCN_HIPX_BF16_1CB / _PIPE / RCN_HIPX_HALO were read into function-local statics at first use: frozen process-wide, and rcn_hipx_plan
    reported whatever the first call had latched.  Now fields of the net (rcn_hipx_set_option), seeded from the environment at creation:
    two nets of ONE process run different kernels for the same layer, each net's plan (rcn_hipx_plan_net) names the kernels that net
    runs, and the two forms of the same arithmetic agree."""
    import torch
    from mercer_research_amd.convnet import ConvNet, ConvNetError
    in_shape, layers, B = (16, 16, 3), (("conv", 32), ("conv", 32), ("pool",), ("dense_relu", 64), ("dense", 10)), 32
    a = ConvNet(in_shape, layers, B)
    monkeypatch.setenv("RCN_HIPX_BF16_1CB", "0")                        # seeds only nets created from here on
    b = ConvNet(in_shape, layers, B)
    monkeypatch.delenv("RCN_HIPX_BF16_1CB")
    assert a.get_option("bf16_1cb") == 1 and b.get_option("bf16_1cb") == 0
    with pytest.raises(ConvNetError):
        a.set_option("no_such_option", 1)
    with pytest.raises(ConvNetError):
        a.set_option("halo", 2)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    outs = []
    for net in (a, b):
        net.init_params(7)
        net.set_precision("bf16")
        plan = net.plan_of_this_net(B)
        outs.append(plan)
        xd, yd = net.to_device(x), net.to_device(y)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        for _ in range(3):
            net.train_step(xd, yd, 0.01, loss)
        net.synchronize()
        outs.append((float(loss.item()), net.get_params()))
    assert "k_conv3x3_halo_bf16_1cb" in outs[0] and "k_conv3x3_halo_bf16_1cb" not in outs[2] and "k_conv3x3_halo_bf16p" in outs[2], (outs[0], outs[2])
    assert abs(outs[1][0] - outs[3][0]) <= 1e-3 * abs(outs[1][0])
    assert np.allclose(outs[1][1], outs[3][1], rtol=2e-2, atol=2e-3)
    b.set_option("bf16_1cb", 1)                                         # settable afterwards, per net; the plan follows
    assert "k_conv3x3_halo_bf16_1cb" in b.plan_of_this_net(B)
    a.close(); b.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16_stored"])
def test_trackx_bucketed_gradients_are_the_gradients_bit_for_bit(precision):
    """rcn_hipx_gradients_begin_dev / _bucket_dev: the backward pass as a resumable walk that stops after every bucket of layers and runs
    that bucket's slab reduction, so that a data-parallel step can all-reduce the slice while the layers below still run.  Same kernels,
    same sums: for every bucket size the flat gradient and the loss equal rcn_hipx_gradients_dev's bit for bit, the slices arrive from
    the top of the buffer down and cover it exactly once, and a slice is already final when its callback runs."""
    import torch
    from mercer_research_amd.convnet import ConvNet
    in_shape, layers, B = (16, 16, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 128), ("dense", 10)), 64
    net = ConvNet(in_shape, layers, B)
    net.init_params(5)
    net.set_precision(precision)
    rng = np.random.default_rng(11)
    x = net.to_device(rng.standard_normal((B,) + in_shape).astype(np.float32))
    y = net.to_device(rng.integers(0, 10, B).astype(np.int32))
    loss0 = torch.zeros(1, dtype=torch.float32, device=net.device)
    ref = net.gradients(x, y, None, loss0)
    net.synchronize()
    ref_h, l0 = ref.cpu().numpy().copy(), float(loss0.item())
    for min_bytes in (0, 64 << 10, 1 << 30):
        grad = torch.full((net.n_padded,), float("nan"), dtype=torch.float32, device=net.device)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        seen = []

        def on_bucket(piece, k, n):
            net.synchronize()                                           # (a test may; a step records an event instead)
            off = piece.storage_offset()
            seen.append((k, n, off, piece.numel()))
            assert np.array_equal(piece.cpu().numpy(), ref_h[off:off + piece.numel()]), (min_bytes, k)
        net.gradients_bucketed(x, y, grad, loss, min_bytes, on_bucket)
        net.synchronize()
        assert np.array_equal(grad.cpu().numpy(), ref_h) and float(loss.item()) == l0
        assert [s[0] for s in seen] == list(range(seen[0][1]))
        assert seen[-1][2] == 0 and seen[0][2] + seen[0][3] == net.n_padded
        assert all(seen[i][2] == seen[i + 1][2] + seen[i + 1][3] for i in range(len(seen) - 1))
        assert (len(seen) == 1) if min_bytes == 1 << 30 else (len(seen) == 4 if min_bytes == 0 else len(seen) >= 2)
    net.close()


def _layer_slices(in_shape, layers):
    """(kind, w slice, b slice) of every layer with parameters in the logical flat layout (W[K][Cout] then b[Cout], layers back to back)."""
    H, W, Cc = in_shape
    off, out, flat = 0, [], False
    for l in layers:
        if l[0] == "conv":
            k, o = 9 * Cc, l[1]
            Cc = o
        elif l[0] == "pool":
            H, W = H // 2, W // 2
            continue
        else:
            k, o = (Cc if flat else H * W * Cc), l[1]
            Cc, flat = o, True
        out.append((l[0], slice(off, off + k * o), slice(off + k * o, off + k * o + o)))
        off += k * o + o
    return out


@pytest.mark.parametrize("shape", ["pool_pairs", "conv_conv"])
def test_trackx_bf16_storage_rounds_what_bf16_mode_rounds(shape):
    """RCN_HIPX_BF16_STORED: the convolutional stage's activations and gradients live in HBM as bf16.  Every consumer
    of those tensors rounds them to bf16 on the way into LDS in bf16 mode anyway, so storing them rounded changes WHERE the rounding
    happens, not what is rounded: logits and loss equal bf16 mode's bit for bit, and so does the weight gradient of every layer but the
    first (whose fp32 kernel now reads a rounded dZ); bias gradients (summed in fp32 from dZ) and the first layer's weights agree to bf16
    resolution.  Then ten training steps in each mode end at the same loss to 2 %."""
    import torch
    from mercer_research_amd.convnet import ConvNet
    if shape == "pool_pairs":
        in_shape, layers, B = (16, 16, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 128), ("dense", 10)), 64
    else:
        in_shape, layers, B = (32, 32, 1), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("conv", 128), ("pool",), ("dense", 10)), 128    # (a batch at which bf16 mode does not split K either: same sums)
    rng = np.random.default_rng(17)
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    res = {}
    for mode in ("bf16", "bf16_stored"):
        net = ConvNet(in_shape, layers, B)
        net.init_params(9)
        net.set_precision(mode)
        plan = net.plan_of_this_net(B)
        assert ("stored as bf16" in plan) == (mode == "bf16_stored"), plan
        xd, yd = net.to_device(x), net.to_device(y)
        logits_d = net.forward(xd)
        net.synchronize()
        logits = logits_d.cpu().numpy().copy()
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        g = net.gradients(xd, yd, None, loss)
        net.synchronize()
        grad, l0 = net.unpad(g), float(loss.item())
        for _ in range(10):
            net.train_step(xd, yd, 0.02, loss)
        net.synchronize()
        res[mode] = (logits, l0, grad, float(loss.item()))
        net.close()
    a, b = res["bf16"], res["bf16_stored"]
    assert np.array_equal(a[0], b[0]) and a[1] == b[1], (a[1], b[1])
    assert np.isfinite(b[2]).all() and np.abs(b[2]).max() > 0
    for li, (kind, ws, bs) in enumerate(_layer_slices(in_shape, layers)):
        if li > 0:
            assert np.array_equal(a[2][ws], b[2][ws]), (li, kind, float(np.abs(a[2][ws] - b[2][ws]).max()))
        else:
            assert np.linalg.norm(a[2][ws] - b[2][ws]) <= 1e-2 * np.linalg.norm(a[2][ws]), (li, kind)
        assert np.linalg.norm(a[2][bs] - b[2][bs]) <= 1e-2 * np.linalg.norm(a[2][bs]) + 1e-7, (li, kind)
    assert b[3] < b[1] and abs(a[3] - b[3]) <= 2e-2 * abs(a[3]), (a[1], a[3], b[3])


@pytest.mark.parametrize("in_shape,layers,B", [
    ((16, 16, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 32), ("dense", 10)), 5),
    ((32, 32, 1), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense", 7)), 3),
    ((16, 32, 3), (("conv", 64), ("pool",), ("conv", 128), ("conv", 128), ("pool",), ("dense_relu", 128), ("dense_relu", 32), ("dense", 3)), 9),
    # 24-pixel-wide maps: the first layer's kernels take 8 x 8 blocks of TWO images side by side (an odd batch leaves the last block half empty)
    ((24, 24, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 64), ("dense", 10)), 5),
    ((24, 24, 1), (("conv", 32), ("conv", 32), ("pool",), ("dense", 10)), 6),
])
def test_trackx_bf16_storage_matches_the_oracle_with_the_storage_rounding_mirrored(in_shape, layers, B):
    """RCN_HIPX_BF16_STORED against oracle/convnet_oracle.py evaluated with the same operand rounding AND the same storage rounding
    (stored=True: maps rounded where they are written, gradients with respect to maps rounded where the input-gradient kernel writes
    them): logits, loss, every gradient and two training steps within the tolerances of the bf16-operand test (5e-3 of each tensor's
    scale; 2e-2 after two steps)."""
    import torch
    from mercer_research_amd.convnet import ConvNet
    from oracle import convnet_oracle as co
    rng = np.random.default_rng(B + 3)
    net = ConvNet(in_shape, layers, B)
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    net.set_params(co.flatten(ws, bs).astype(np.float32))
    x = rng.standard_normal((B,) + in_shape)
    y = rng.integers(0, layers[-1][1], B).astype(np.int32)
    xd, yd = net.to_device(x.astype(np.float32)), net.to_device(y)
    x64 = x.astype(np.float32).astype(np.float64)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(x64, y, w32, b32, layers, operand="bf16", stored=True)
    net.set_precision("bf16_stored")

    def close(a, b, rtol):
        scale = max(1e-3, float(np.abs(b).max()))                      # (tests/test_gpu_convnet.py _close: of each tensor's scale)
        assert np.abs(np.asarray(a, dtype=np.float64) - b).max() <= rtol * scale + 1e-6, (float(np.abs(np.asarray(a, dtype=np.float64) - b).max()), scale)
    logits = net.forward(xd)
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    close(logits.cpu().numpy(), logits_ref, 5e-3)
    assert abs(loss.item() - loss_ref) <= 5e-3 * max(1.0, loss_ref)
    got, ref = net.unpad(grad), co.flatten(gws, gbs)
    for kind, wsl, bsl in _layer_slices(in_shape, layers):
        close(got[wsl], ref[wsl], 5e-3)
        close(got[bsl], ref[bsl], 5e-3)
    net.train_step(xd, yd, 0.05, loss)
    net.train_step(xd, yd, 0.05, loss)
    net.synchronize()
    nw, nb, _ = co.sgd_step(x64, y, w32, b32, layers, 0.05, operand="bf16", stored=True)
    nw, nb, _ = co.sgd_step(x64, y, nw, nb, layers, 0.05, operand="bf16", stored=True)
    close(net.get_params(), co.flatten(nw, nb), 2e-2)
    net.close()


def test_trackx_bf16_storage_16x16_pixel_blocks_compute_the_same_bits():
    """Option "bf16_rows16": k_conv3x3_halo_bf16p<..., MG = 2> -- a wave computes two 32-pixel row groups against the same staged weights.
    Same operands, same accumulation order per output: logits, loss, gradients and three training steps equal the 8-row form's bit for
    bit, and the plan names the form for the layers whose height it covers without extra padding."""
    import torch
    from mercer_research_amd.convnet import ConvNet
    in_shape, layers, B = (32, 32, 3), (("conv", 32), ("conv", 64), ("pool",), ("conv", 64), ("conv", 128), ("pool",), ("dense_relu", 64), ("dense", 10)), 128
    rng = np.random.default_rng(23)
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    res = []
    for rows16 in (0, 1):
        net = ConvNet(in_shape, layers, B)
        net.init_params(4)
        net.set_option("bf16_rows16", rows16)
        net.set_precision("bf16_stored")
        plan = net.plan_of_this_net(B)
        assert ("16 x 16 pixel blocks" in plan) == bool(rows16), plan
        xd, yd = net.to_device(x), net.to_device(y)
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        g = net.gradients(xd, yd, None, loss)
        net.synchronize()
        out = [logits.cpu().numpy().copy(), float(loss.item()), g.cpu().numpy().copy()]
        for _ in range(3):
            net.train_step(xd, yd, 0.002, loss)
        net.synchronize()
        out += [net.get_params(), float(loss.item())]
        res.append(out)
        net.close()
    a, b = res
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
    assert np.isfinite(b[3]).all() and np.isfinite(b[4])


def test_trackx_bf16_storage_partial_batches_and_mode_switches():
    """A net built for 32 pictures steps batches of 7 and 32 in bf16-storage mode (in this mode a 3 x 3 layer's kernels do not depend on
    the batch size), then is switched to bf16 and fp32 mode and back: every forward pass agrees with the oracle evaluated in that mode on
    the parameters the net has at that point."""
    import torch
    from mercer_research_amd.convnet import ConvNet
    from oracle import convnet_oracle as co
    in_shape, layers, MB = (16, 16, 3), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 64), ("dense", 10)), 32
    rng = np.random.default_rng(31)
    net = ConvNet(in_shape, layers, MB)
    net.init_params(3)
    net.set_precision("bf16_stored")
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    for B in (7, 32, 7):
        x = rng.standard_normal((B,) + in_shape).astype(np.float32)
        y = rng.integers(0, 10, B).astype(np.int32)
        net.train_step(net.to_device(x), net.to_device(y), 0.01, loss)
    net.synchronize()
    assert np.isfinite(loss.item())
    x = rng.standard_normal((7,) + in_shape).astype(np.float32)
    xd = net.to_device(x)
    ws, bs = co.unflatten(net.get_params().astype(np.float64), in_shape, layers)
    for mode, kw, tol in (("bf16_stored", dict(operand="bf16", stored=True), 5e-3), ("bf16", dict(operand="bf16"), 5e-3), ("fp32", dict(), 2e-4),
                          ("bf16_stored", dict(operand="bf16", stored=True), 5e-3)):
        net.set_precision(mode)
        lg = net.forward(xd)
        net.synchronize()
        ref = co.forward(x.astype(np.float64), ws, bs, layers, **kw)
        assert np.abs(lg.cpu().numpy() - ref).max() <= tol * max(1e-3, np.abs(ref).max()) + 1e-6, mode
    net.close()


def test_trackx_bf16_storage_refuses_a_net_it_does_not_cover():
    """rcn_hipx_set_precision walks the net's plan first: a net with a layer that no bf16-tensor kernel runs (here: the LDS-tiled kernels
    switched off) gets -3 with the reason, and stays in the mode it was in."""
    from mercer_research_amd.convnet import ConvNet, ConvNetError
    in_shape, layers, B = (16, 16, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense", 10)), 16
    net = ConvNet(in_shape, layers, B)
    net.init_params(1)
    net.set_precision("bf16")
    net.set_option("halo", 0)
    with pytest.raises(ConvNetError, match="bf16 storage"):
        net.set_precision("bf16_stored")
    assert "stored as bf16" not in net.plan_of_this_net(B)
    net.set_option("halo", 1)
    net.set_precision("bf16_stored")
    assert "stored as bf16" in net.plan_of_this_net(B)
    net.close()


@pytest.mark.parametrize("B", [10, 200])
def test_f64_epoch_from_images_on_the_resident_kernel_equals_features_then_train(amd, oracle, B):
    """The end-to-end form (u8 pictures -> features -> standardise -> packed epoch image in one kernel per segment, then the steps) in the
    f64 context on the resident kernel (k_features_cpcp_packed<double> feeding k_xcd_epoch<double>): bit-identical to
    rcn_hip_features_dev(standardize) + rcn_hip_train_epoch_dev, and both within 1e-10 of the oracle's loop on the oracle's features."""
    from mercer_research_amd.device import DeviceRCN
    nb, N = 5, 1200
    imgs, labels = synthetic_images(N, seed=23)
    ws, bs = synthetic_params([784, 30, 10], seed=9)
    ws = [w * 0.1 for w in ws]
    order = np.random.default_rng(4).permutation(N).astype(np.int32)
    res = []
    for images_path in (True, False):
        d = DeviceRCN(dtype=F64)
        _xcd_or_skip(d)
        d.set_dense_path(0)
        assert d.train_epoch_resident(B)
        d.set_params(ws, bs)
        dev = d.to_device(imgs)
        d.gen_scales(d.features(dev))                                   # sets scale_set
        Y = d.to_device(one_hot(labels), d.tdtype)
        perm = d.to_device(order)
        loss = d.empty(nb)
        if images_path:
            d.train_epoch_images(dev, Y, perm, B, nb, 3.0, loss)
        else:
            d.train_epoch(d.features(dev, standardize=True), Y, perm, B, nb, 3.0, loss)
        gw, gb = d.get_params()
        res.append((gw + gb, loss.cpu().numpy().copy()))
        assert d.fallbacks_taken() == 0
        d.rcn.close()
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][1], res[1][1])
    f = oracle.features(imgs, DEFAULT_LAYERS)
    X = oracle.standardize(f, *oracle.gen_scales(f))
    rw, rb, cs = _oracle_steps(oracle, ws, bs, X, one_hot(labels), order.astype(np.int64), B, nb, 3.0)
    np.testing.assert_allclose(res[0][1], cs, rtol=1e-10)
    for a, b in zip(res[0][0], rw + rb):
        assert np.all(np.abs(a - b) <= 1e-10 * np.abs(b) + 1e-11), float(np.abs(a - b).max())


def test_the_drop_in_call_sequence_in_f64_at_batch_size_10_runs_on_the_resident_kernel(amd, oracle):
    """What rust/rcn-hip's RCN::train does (rcn.rs:126-167 over rcn_hip_load_data / rcn_hip_train_set_epoch / rcn_hip_evaluate_set) in the
    F64 context that crate opens, at the reference's own batch_size 10 (rcn/src/main.rs:36-37): the steps run on the resident kernel
    (rcn_hip_train_epoch_resident) and follow the oracle's restatement -- per-step costs, accepted counts, final parameters."""
    B, epochs = 10, 2
    imgs, labels = synthetic_images(205, seed=3)                        # chunks_exact(10) drops the last five (rcn.rs:147)
    timgs, tlabels = synthetic_images(80, seed=4)
    ws, bs = synthetic_params([784, 30, 10], seed=21)
    ws = [w * 0.05 for w in ws]
    r = amd.RCN(10, amd.default_convpool(), [30], input_shape=(28, 28), dtype=amd.F64)
    r.set_params(ws, bs)
    r.load_set(0, imgs, labels); r.load_set(1, timgs, tlabels)
    if not r._lib.rcn_hip_train_epoch_resident(r._ctx, B):
        r.close()
        pytest.skip("the resident one-XCD kernel does not apply on this device")
    f, tf = oracle.features(imgs, DEFAULT_LAYERS), oracle.features(timgs, DEFAULT_LAYERS)
    X = oracle.standardize(f, *oracle.gen_scales(f))
    TX = oracle.standardize(tf, *oracle.gen_scales(tf))
    Y, TY = one_hot(labels), one_hot(tlabels)
    rng = np.random.default_rng(99)
    rw, rb = ws, bs
    for e in range(epochs):
        order = rng.permutation(205).astype(np.int32)
        loss = r.train_set_epoch(0, B, 3.0, perm=order, want_loss=True)
        assert loss.shape == (20,)
        rw, rb, cs = _oracle_steps(oracle, rw, rb, X, Y, order.astype(np.int64), B, 20, 3.0)
        np.testing.assert_allclose(loss, cs, rtol=1e-9)
        out = oracle.classify_test(rw, rb, TX)
        assert r.evaluate_set(1) == sum(oracle.eval_accept(out[i], TY[i]) for i in range(len(TX)))
    gw, gb = r.get_params()
    for a, b in zip(gw + gb, rw + rb):
        assert np.all(np.abs(a - b) <= 1e-10 * np.abs(b) + 1e-11)
    assert r.fallbacks_taken() == 0
    r.close()


def test_bench_n2_flow_rehearsed_on_one_gpu():
    """`bench.py --gpus 2 --rehearse-on-one-gpu`: the script's N > 1 flow -- admission, rehearsal of the two-kernel form, the voted trial of
    the resident form, the timed loop (barrier, max over ranks), the replica check, the clocked diagnostic epoch -- executed by two real
    processes on this box's GPU (gloo votes, shards of 128, rank r's workers on physical XCD r).  A diagnostic of the FLOW the driver
    would start on a node; its numbers mean nothing and the line says so."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "64", "--warmup", "16"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    c = j["config"]
    assert j["n_gpus"] == 2 and j["steps"] == 64 and c["parallelism"] == "dp2" and "one_gpu_rehearsal" in c
    assert c["replicas_identical"] is True and c["dp_fallbacks_taken"] == [] and c["dp_timeout_site"] is None
    assert c["dp_form_trial"]["kept"] in ("resident kernel", "two-kernel pipeline") and c["dp_form_trial"]["resident_healthy"] is True
    if c["dp_form_trial"]["kept"] == "resident kernel":
        ph = c["dp_phase_us"]
        assert ph and ph["steps"] == 128 and ph["owner_wait_mean"] > 0 and ph["member_wait_mean"] > 0 and c["dp_rs_wait_us"] == ph["owner_wait_mean"]
    assert j["value"] > 0 and np.isfinite(c["final_cost_rank0"])
