"""GPU tests of the Track-X trainable convolution network (include/rcn_hipx.h) against its f64 oracle.
No reference counterpart exists ("parity unpinned"); tolerance: fp32 MFMA vs f64, |d| <= 2e-4 * scale + 1e-6."""
import numpy as np
import pytest

from oracle import convnet_oracle as co

pytestmark = pytest.mark.gpu


def _net(in_shape, layers, B):
    from mercer_research_amd.convnet import ConvNet
    return ConvNet(in_shape, layers, B)


def _close(a, b, rtol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1e-3, float(np.abs(b).max()))
    assert np.abs(a - b).max() <= rtol * scale + 1e-6, (float(np.abs(a - b).max()), scale)


@pytest.mark.parametrize("in_shape,layers,B", [
    ((8, 8, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 32), ("dense", 10)), 5),
    ((6, 6, 1), (("conv", 32), ("conv", 32), ("pool",), ("dense", 7)), 3),
    ((4, 8, 3), (("conv", 64), ("pool",), ("dense_relu", 64), ("dense_relu", 32), ("dense", 3)), 130),
    ((8, 8, 32), (("conv", 32), ("pool",), ("dense", 10)), 4),
    # the layer stack of the synthetic 224x224x3 config (BASELINE configs[3]) at 16x16: 8 conv layers, pool after each pair
    ((16, 16, 3), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("conv", 64), ("pool",), ("conv", 128), ("conv", 128), ("pool",),
                   ("conv", 256), ("conv", 256), ("pool",), ("dense", 10)), 3),
])
def test_forward_gradients_and_step_match_oracle(in_shape, layers, B):
    rng = np.random.default_rng(B)
    net = _net(in_shape, layers, B)
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    flat = co.flatten(ws, bs)
    assert flat.size == net.n_logical
    net.set_params(flat)
    assert np.array_equal(net.get_params(), flat.astype(np.float32))
    x = rng.standard_normal((B,) + in_shape)
    y = rng.integers(0, layers[-1][1], B).astype(np.int32)
    xd, yd = net.to_device(x.astype(np.float32)), net.to_device(y)
    x64 = x.astype(np.float32).astype(np.float64)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(x64, y, w32, b32, layers)
    net.synchronize()
    import torch
    with torch.cuda.stream(net.stream):
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref)
    assert abs(loss.item() - loss_ref) <= 2e-4 * max(1.0, loss_ref)
    _close(net.unpad(grad), co.flatten(gws, gbs))
    # one SGD step (eager first call, then the cached graph for the second)
    lr = 0.05
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)
    net.synchronize()
    nw, nb, _ = co.sgd_step(x64, y, w32, b32, layers, lr)
    _close(net.get_params(), co.flatten(nw, nb))
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)            # graph replay
    net.synchronize()
    nw2, nb2, l2 = co.sgd_step(x64, y, nw, nb, layers, lr)
    _close(net.get_params(), co.flatten(nw2, nb2), rtol=4e-4)
    assert abs(loss.item() - l2) <= 4e-4 * max(1.0, l2)
    # data-parallel halves: apply(gradients) == train_step
    net.set_params(flat)
    with torch.cuda.stream(net.stream):
        g = net.gradients(xd, yd)
        net.apply(g, lr)
    net.synchronize()
    _close(net.get_params(), co.flatten(nw, nb))


@pytest.mark.parametrize("in_shape,layers,B", [
    ((8, 8, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 32), ("dense", 10)), 5),
    ((6, 6, 1), (("conv", 32), ("conv", 32), ("pool",), ("dense", 7)), 3),
    ((4, 8, 3), (("conv", 128), ("pool",), ("dense_relu", 128), ("dense_relu", 32), ("dense", 3)), 130),
    ((16, 16, 3), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("conv", 64), ("pool",), ("conv", 128), ("conv", 128), ("pool",),
                   ("conv", 256), ("conv", 256), ("pool",), ("dense", 10)), 3),
])
def test_bf16_mfma_path_matches_bf16_operand_oracle(in_shape, layers, B):
    """rcn_hipx_set_precision(BF16): GEMM operands of forward, dgrad and wgrad rounded to bf16 (RNE), fp32 accumulation and
    storage, fp32 bias gradients and update; the first layer (9 * Cin <= 32) and the fused classifier head stay fp32 (the oracle's
    operand="bf16" mirrors that rule by shape).  Checked against the oracle evaluated with the SAME operand rounding (oracle.round_bf16; f64
    products and sums): 5e-3 of each tensor's scale -- what is left is accumulation precision and the rare operand that
    rounds the other way because its fp32 value differs in the last bit.  Against the unrounded f64 oracle the logits sit
    within 3e-2 (the price of 8-bit mantissas), and switching back to fp32 restores the 2e-4 agreement."""
    import torch
    rng = np.random.default_rng(B + 1)
    net = _net(in_shape, layers, B)
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    flat = co.flatten(ws, bs)
    net.set_params(flat)
    x = rng.standard_normal((B,) + in_shape)
    y = rng.integers(0, layers[-1][1], B).astype(np.int32)
    xd, yd = net.to_device(x.astype(np.float32)), net.to_device(y)
    x64 = x.astype(np.float32).astype(np.float64)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(x64, y, w32, b32, layers, operand="bf16")
    logits_f64 = co.forward(x64, w32, b32, layers)
    net.set_precision("bf16")
    with torch.cuda.stream(net.stream):
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref, rtol=5e-3)
    _close(logits.cpu().numpy(), logits_f64, rtol=3e-2)
    assert abs(loss.item() - loss_ref) <= 5e-3 * max(1.0, loss_ref)
    _close(net.unpad(grad), co.flatten(gws, gbs), rtol=5e-3)
    # it really is a different arithmetic: not bit-equal to the fp32 path, which still holds its own tolerance
    net.set_precision("fp32")
    with torch.cuda.stream(net.stream):
        logits32 = net.forward(xd)
    net.synchronize()
    _close(logits32.cpu().numpy(), logits_f64)
    assert not np.array_equal(logits32.cpu().numpy(), logits.cpu().numpy())
    # two training steps in bf16 mode (eager, then the cached graph) follow the bf16-operand oracle's steps
    net.set_precision("bf16")
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, 0.05, loss)
        net.train_step(xd, yd, 0.05, loss)
    net.synchronize()
    nw, nb, _ = co.sgd_step(x64, y, w32, b32, layers, 0.05, operand="bf16")
    nw, nb, _ = co.sgd_step(x64, y, nw, nb, layers, 0.05, operand="bf16")
    # second step: operands that round the other way after step one move a few weights by a bf16 ulp of their gradient; 2e-2 of scale
    _close(net.get_params(), co.flatten(nw, nb), rtol=2e-2)


def test_cached_step_graphs_survive_scratch_growth():
    """Steps at batch 4 (graph cached), then at batch 48 (larger scratch buffers: they move), then batch 4 again: the graph
    cached for the first shape pointed into the old scratch and must not be replayed as is.  Five steps against the oracle."""
    import torch
    in_shape = (8, 8, 3)
    layers = (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 32), ("dense", 10))
    rng = np.random.default_rng(44)
    net = _net(in_shape, layers, 48)
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    net.set_params(co.flatten(ws, bs))
    w = [a.astype(np.float32).astype(np.float64) for a in ws]
    b = [a.astype(np.float32).astype(np.float64) for a in bs]
    batches = {}
    for B in (4, 48):
        x = rng.standard_normal((B,) + in_shape).astype(np.float32)
        y = rng.integers(0, 10, B).astype(np.int32)
        batches[B] = (x, y, net.to_device(x), net.to_device(y))
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    for B in (4, 4, 48, 4, 48):
        x, y, xd, yd = batches[B]
        with torch.cuda.stream(net.stream):
            net.train_step(xd, yd, 0.05, loss)
        w, b, _ = co.sgd_step(x.astype(np.float64), y, w, b, layers, 0.05)
    net.synchronize()
    _close(net.get_params(), co.flatten(w, b), rtol=1e-3)


def test_training_reduces_loss_on_cifar_shape():
    """CIFAR-10 shape net of SURVEY.md §8(d): 32x32x3, conv 3->32, pool, 32->64, pool, 64->128, pool -> 2048 -> 256 -> 10."""
    import torch
    layers = (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("conv", 128), ("pool",), ("dense_relu", 256), ("dense", 10))
    B = 64
    net = _net((32, 32, 3), layers, B)
    net.init_params(3)
    rng = np.random.default_rng(0)
    protos = rng.standard_normal((10, 32, 32, 3)).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.int32)
    x = protos[y] + 0.3 * rng.standard_normal((B, 32, 32, 3)).astype(np.float32)
    xd, yd = net.to_device(x), net.to_device(y)
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    net.synchronize()
    losses = []
    for _ in range(30):
        with torch.cuda.stream(net.stream):
            net.train_step(xd, yd, 0.02, loss)
        net.synchronize()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], losses[::5]
    assert net.step_flops(B) > 0


def test_unsupported_shapes_are_rejected():
    from mercer_research_amd.convnet import ConvNetError
    with pytest.raises(ConvNetError):
        _net((8, 8, 3), (("conv", 30), ("dense", 10)), 4)            # channels not a multiple of 32
    with pytest.raises(ConvNetError):
        _net((7, 8, 3), (("conv", 32), ("pool",), ("dense", 10)), 4)   # odd height under the pool
    with pytest.raises(ConvNetError):
        _net((8, 8, 3), (("conv", 32), ("pool",), ("dense_relu", 32)), 4)   # no logits layer
