"""Track X at WORKLOAD size (BASELINE.json configs[2], [3], [4]) -- the shapes bench_convnet.py times, not toy shapes.

No reference counterpart exists (include/rcn_hipx.h); at these sizes the f64 oracle (oracle/convnet_oracle.py) is affordable only on
a few images, so each config is held by: the oracle on a sub-batch at full resolution, and size-independent properties on the full
batch -- gradient additivity over a split of the batch, data-parallel halves == the full step, hipGraph replay == eager bit for bit.
Tolerances as in test_gpu_convnet.py: fp32 MFMA |d| <= 2e-4 * scale + 1e-6; bf16 operands 5e-3 * scale."""
import os

import numpy as np
import pytest

from oracle import convnet_oracle as co

pytestmark = pytest.mark.gpu

CIFAR = ((32, 32, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("conv", 128), ("pool",), ("dense_relu", 256), ("dense", 10)), 512)
SYNTH224 = ((224, 224, 3), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("conv", 64), ("pool",), ("conv", 128), ("conv", 128), ("pool",),
                            ("conv", 256), ("conv", 256), ("pool",), ("dense", 10)), 128)
MNIST = ((28, 28, 1), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 128), ("dense", 10)), 4096)


def _net(in_shape, layers, B):
    from mercer_research_amd.convnet import ConvNet
    return ConvNet(in_shape, layers, B)


def _close(a, b, rtol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1e-3, float(np.abs(b).max()))
    assert np.abs(a - b).max() <= rtol * scale + 1e-6, (float(np.abs(a - b).max()), scale)


def _params(rng, in_shape, layers):
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    return ws, bs


def _properties(net, x, y, flat, lr, tol, bitwise_replay=True):
    """gradient additivity over a split, DP halves == full step, graph replay == eager, on the full batch held by (x, y)"""
    import torch
    B = x.shape[0]
    h = B // 2
    xd, yd = net.to_device(x), net.to_device(y)
    net.set_params(flat)
    with torch.cuda.stream(net.stream):
        lf, l1, l2 = (torch.zeros(1, dtype=torch.float32, device=net.device) for _ in range(3))
        g_full = net.gradients(xd, yd, loss=lf).clone()
        g1 = net.gradients(xd[:h], yd[:h], loss=l1).clone()
        g2 = net.gradients(xd[h:], yd[h:], loss=l2).clone()
        gm = (g1 + g2) * 0.5
        gs = g1 + g2
    net.synchronize()
    # gradients of the MEAN loss: the full batch's is the mean of the halves'
    _close(net.unpad(g_full), net.unpad(gm), rtol=tol)
    assert abs(lf.item() - 0.5 * (l1.item() + l2.item())) <= tol * max(1.0, abs(lf.item()))
    # data-parallel halves (what bench_convnet.py --gpus 2 does with an all-reduce in between) == one full step
    with torch.cuda.stream(net.stream):
        net.apply(gs, lr / 2)
    net.synchronize()
    p_dp = net.get_params()
    net.set_params(flat)
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)                       # first call for these arguments: eager + capture
    net.synchronize()
    p_eager, l_eager = net.get_params(), loss.item()
    _close(p_dp, p_eager, rtol=tol)
    net.set_params(flat)
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)                       # same arguments: the cached hipGraph replays
    net.synchronize()
    p_replay = net.get_params()
    if bitwise_replay:
        assert np.array_equal(p_replay, p_eager) and loss.item() == l_eager      # same kernels, fixed reduction orders
    else:
        _close(p_replay, p_eager, rtol=tol)
    assert np.isfinite(p_eager).all() and np.isfinite(l_eager)
    return l_eager


def test_cifar_b512_fp32_workload_size():
    """configs[2]: CIFAR-10 shape, 3 conv + 2 dense, batch 512, fp32 MFMA."""
    import torch
    in_shape, layers, B = CIFAR
    rng = np.random.default_rng(512)
    ws, bs = _params(rng, in_shape, layers)
    flat = co.flatten(ws, bs)
    net = _net(in_shape, layers, B)
    assert flat.size == net.n_logical
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    # the oracle on a 4-image sub-batch at the full 32x32 resolution: logits, loss, every gradient
    net.set_params(flat)
    xs, ys = x[:4], y[:4]
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(xs.astype(np.float64), ys, w32, b32, layers)
    with torch.cuda.stream(net.stream):
        xd, yd = net.to_device(xs), net.to_device(ys)
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref)
    assert abs(loss.item() - loss_ref) <= 2e-4 * max(1.0, loss_ref)
    _close(net.unpad(grad), co.flatten(gws, gbs))
    # the full 512-image batch: logits of the first 4 images do not depend on the batch they ride in
    with torch.cuda.stream(net.stream):
        logits_full = net.forward(net.to_device(x))
    net.synchronize()
    _close(logits_full.cpu().numpy()[:4], logits_ref)
    _properties(net, x, y, flat, 0.02, 2e-4)
    assert abs(net.step_flops(B) / 1e9 - 32.4) < 0.5             # the figure bench_convnet.py prices the step at
    net.close()


def test_synth224_8conv_workload_size():
    """configs[3]: synthetic 224x224x3, 8 conv layers, 128 images per GPU (global 1024 on 8).  Oracle at B = 2 on the full
    224x224 resolution (logits, loss, gradients); split / replay properties at B = 128."""
    import torch
    in_shape, layers, B = SYNTH224
    rng = np.random.default_rng(224)
    ws, bs = _params(rng, in_shape, layers)
    flat = co.flatten(ws, bs)
    net = _net(in_shape, layers, B)
    net.set_params(flat)
    xs = rng.standard_normal((2,) + in_shape).astype(np.float32)
    ys = rng.integers(0, 10, 2).astype(np.int32)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(xs.astype(np.float64), ys, w32, b32, layers)
    with torch.cuda.stream(net.stream):
        xd, yd = net.to_device(xs), net.to_device(ys)
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref)
    assert abs(loss.item() - loss_ref) <= 2e-4 * max(1.0, loss_ref)
    _close(net.unpad(grad), co.flatten(gws, gbs))
    del xd, yd, grad
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    _properties(net, x, y, flat, 1e-6, 2e-4)
    # the same batch in bf16 mode (the mode the 8.05 ms figure is quoted in): properties within the bf16 tolerance
    net.set_precision("bf16")
    _properties(net, x, y, flat, 1e-6, 5e-3)
    net.close()


def test_mnist_bf16_b4096_hipgraph_step_workload_size():
    """configs[4]: MNIST shape, bf16 MFMA path, batch 4096, hipGraph-captured train step: replay == eager bit for bit,
    additivity and the data-parallel halves within the bf16 tolerance, and the loss against the bf16-operand oracle on a
    16-image sub-batch."""
    import torch
    in_shape, layers, B = MNIST
    rng = np.random.default_rng(4096)
    ws, bs = _params(rng, in_shape, layers)
    flat = co.flatten(ws, bs)
    net = _net(in_shape, layers, B)
    net.set_precision("bf16")
    net.set_params(flat)
    xs = rng.standard_normal((16,) + in_shape).astype(np.float32)
    ys = rng.integers(0, 10, 16).astype(np.int32)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(xs.astype(np.float64), ys, w32, b32, layers, operand="bf16")
    with torch.cuda.stream(net.stream):
        xd, yd = net.to_device(xs), net.to_device(ys)
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref, rtol=5e-3)
    assert abs(loss.item() - loss_ref) <= 5e-3 * max(1.0, loss_ref)
    _close(net.unpad(grad), co.flatten(gws, gbs), rtol=5e-3)
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    l0 = _properties(net, x, y, flat, 0.05, 5e-3)
    # and it trains: 20 replayed steps on a fixed batch lower the loss
    net.set_params(flat)
    xd, yd = net.to_device(x), net.to_device(y)
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    for _ in range(20):
        with torch.cuda.stream(net.stream):
            net.train_step(xd, yd, 0.05, loss)
    net.synchronize()
    assert np.isfinite(loss.item()) and loss.item() < l0
    net.close()


def test_data_parallel_step_replays_as_one_graph_with_the_collective_inside():
    """bench_convnet.py's data-parallel step (shard gradients -> ONE all-reduce over RCCL -> apply) is captured as a hipGraph per batch
    buffer, the collective inside it, and replayed; --force-dp runs it on this box's one GPU as a group of one.  The line says which form
    ran, and the captured form trains like the eager one (same loss after the same steps: same kernels, same order)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("1", "0"):
        env = dict(os.environ, MASTER_PORT=str(29600 + int(mode)), HSA_ENABLE_IPC_MODE_LEGACY="0")
        pr = subprocess.run([sys.executable, os.path.join(root, "bench_convnet.py"), "--config", "mnist", "--steps", "12", "--warmup", "4", "--force-dp", "--dp-graph", mode],
                            capture_output=True, text=True, timeout=240, env=env)
        assert pr.returncode == 0, pr.stderr[-2000:]
        out[mode] = json.loads(pr.stdout.strip().splitlines()[-1])
    assert out["1"]["data_parallel_step"] == "hipGraph" and out["0"]["data_parallel_step"] == "eager"
    assert out["1"]["final_loss"] == out["0"]["final_loss"]
