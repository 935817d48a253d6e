"""GPU tests of the fp32 LDS-tiled Track-X kernels (csrc/convnet_halo.hpp) against the f64 oracle, with the tiling mode forced to
"lds" so that the small shapes an oracle can afford reach them (in "auto" mode small layers take the split-K implicit GEMM).
Shapes are chosen for the kernels' corner cases: both block geometries (8 x 16 of one image, 8 x 8 of two images with an odd
batch), maps that do not fill their blocks, one / two input-channel blocks, 32- and 64-wide column blocks, the fused pool epilogue,
pooled-resolution gradients into the input-gradient and weight-gradient kernels, the first layer with 1 and 3 channels.
No reference counterpart ("parity unpinned"); tolerance as tests/test_gpu_convnet.py: |d| <= 2e-4 * scale + 1e-6."""
import numpy as np
import pytest

from oracle import convnet_oracle as co

pytestmark = pytest.mark.gpu

NETS = [
    # 8 x 16 blocks, maps 12 x 20 and 6 x 10 (partly empty blocks), first layer RGB + fused pool, 32 -> 64 + fused pool
    ((12, 20, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense", 10)), 3),
    # 8 x 8 blocks of two images, odd batch; 1-channel first layer WITHOUT a pool behind it, conv-conv-pool (gate epilogue + pooled input
    # gradient), a 4 x 4 map
    ((8, 8, 1), (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense", 7)), 5),
    # 32-channel input: no first-layer kernel; 96 output channels = three 32-wide column blocks; two input-channel blocks behind it
    ((16, 16, 32), (("conv", 64), ("conv", 96), ("pool",), ("conv", 32), ("dense_relu", 32), ("dense", 10)), 2),
    # first layer with two column blocks on an 8-wide map, then 64 -> 32
    ((24, 8, 3), (("conv", 64), ("pool",), ("conv", 32), ("pool",), ("dense", 4)), 4),
    # a batch of 64 (two workgroups of the fused head, one more hidden dense layer in front of it)
    ((8, 8, 3), (("conv", 32), ("pool",), ("dense_relu", 64), ("dense_relu", 32), ("dense", 10)), 64),
    # the MNIST-shape first layer: ONE channel with the pool right behind it (weight gradient on the 16-row MFMA from a pooled-resolution
    # gradient), 64 filters = two column blocks, a 28 x 12 map (blocks that are not full in either direction), several blocks per chunk
    ((28, 12, 1), (("conv", 64), ("pool",), ("conv", 32), ("pool",), ("dense", 10)), 6),
]


def _close(a, b, rtol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1e-3, float(np.abs(b).max()))
    assert np.abs(a - b).max() <= rtol * scale + 1e-6, (float(np.abs(a - b).max()), scale)


def _setup(in_shape, layers, B, tiling):
    import torch
    from mercer_research_amd.convnet import ConvNet
    rng = np.random.default_rng(B + in_shape[0])
    net = ConvNet(in_shape, layers, B)
    net.set_tiling(tiling)
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    flat = co.flatten(ws, bs)
    net.set_params(flat)
    x = rng.standard_normal((B,) + in_shape).astype(np.float32)
    y = rng.integers(0, layers[-1][1], B).astype(np.int32)
    w32 = [w.astype(np.float32).astype(np.float64) for w in ws]
    b32 = [b.astype(np.float32).astype(np.float64) for b in bs]
    return torch, net, flat, x, y, w32, b32


@pytest.mark.parametrize("in_shape,layers,B", NETS)
def test_lds_tiled_kernels_match_oracle(in_shape, layers, B):
    torch, net, flat, x, y, w32, b32 = _setup(in_shape, layers, B, "lds")
    xd, yd = net.to_device(x), net.to_device(y)
    x64 = x.astype(np.float64)
    loss_ref, logits_ref, gws, gbs = co.loss_and_grads(x64, y, w32, b32, layers)
    with torch.cuda.stream(net.stream):
        logits = net.forward(xd)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        grad = net.gradients(xd, yd, loss=loss)
    net.synchronize()
    _close(logits.cpu().numpy(), logits_ref)
    assert abs(loss.item() - loss_ref) <= 2e-4 * max(1.0, loss_ref)
    g1 = net.unpad(grad)
    _close(g1, co.flatten(gws, gbs))
    # fixed summation orders: a second evaluation is bit-identical
    with torch.cuda.stream(net.stream):
        grad2 = net.gradients(xd, yd)
    net.synchronize()
    assert np.array_equal(net.unpad(grad2), g1)
    # one SGD step eagerly, the second as a replayed graph
    lr = 0.05
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)
    net.synchronize()
    nw, nb, _ = co.sgd_step(x64, y, w32, b32, layers, lr)
    _close(net.get_params(), co.flatten(nw, nb))
    with torch.cuda.stream(net.stream):
        net.train_step(xd, yd, lr, loss)
    net.synchronize()
    nw2, nb2, l2 = co.sgd_step(x64, y, nw, nb, layers, lr)
    _close(net.get_params(), co.flatten(nw2, nb2), rtol=4e-4)
    assert abs(loss.item() - l2) <= 4e-4 * max(1.0, l2)


@pytest.mark.parametrize("in_shape,layers,B", NETS[:3])
def test_tiling_modes_agree(in_shape, layers, B):
    """The three modes are three routes to the same numbers: the implicit-GEMM kernels and the LDS-tiled ones differ only in the
    order of the fp32 sums (and agree on every pooling arg-max here: the logits would show a flipped window).  The third net ends in
    ReLU-dense -> logits: outside "gemm" mode its classifier head is ONE launch (k_head_f32), in "gemm" mode five."""
    out = {}
    for mode in ("gemm", "auto", "lds"):
        torch, net, flat, x, y, _, _ = _setup(in_shape, layers, B, mode)
        xd, yd = net.to_device(x), net.to_device(y)
        with torch.cuda.stream(net.stream):
            logits = net.forward(xd)
            grad = net.gradients(xd, yd)
        net.synchronize()
        out[mode] = (logits.cpu().numpy(), net.unpad(grad))
        net.close()
    for mode in ("auto", "lds"):
        _close(out[mode][0], out["gemm"][0], rtol=1e-5)
        _close(out[mode][1], out["gemm"][1], rtol=2e-5)


def test_set_tiling_rejects_unknown_modes():
    from mercer_research_amd.convnet import ConvNet, ConvNetError
    net = ConvNet((8, 8, 3), (("conv", 32), ("pool",), ("dense", 10)), 2)
    with pytest.raises(ConvNetError):
        net._ck(net.lib.rcn_hipx_set_tiling(net.net, 7))


@pytest.mark.parametrize("in_shape,layers,B", [NETS[0], NETS[2]])
def test_backward_overlap_changes_nothing_but_the_schedule(in_shape, layers, B):
    """The weight gradients of the backward pass run on a second stream beside the input-gradient chain (rcn_hipx_set_overlap):
    same kernels, same summation orders -- parameters after three steps (one eager, two replayed graphs) are bit-identical."""
    out = []
    for on in (True, False):
        torch, net, flat, x, y, _, _ = _setup(in_shape, layers, B, "lds")
        net.set_overlap(on)
        xd, yd = net.to_device(x), net.to_device(y)
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        with torch.cuda.stream(net.stream):
            for _ in range(3):
                net.train_step(xd, yd, 0.05, loss)
        net.synchronize()
        out.append((net.get_params(), loss.item()))
        net.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]
