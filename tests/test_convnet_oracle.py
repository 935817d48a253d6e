"""CPU test of the Track-X oracle (oracle/convnet_oracle.py): its analytic gradients against central finite differences.
This is the ONLY pin the trainable-conv path can have -- the reference has no trainable convolution."""
import numpy as np

from oracle import convnet_oracle as co

LAYERS = (("conv", 4), ("pool",), ("conv", 6), ("pool",), ("dense_relu", 8), ("dense", 5))


def test_gradients_match_finite_differences():
    rng = np.random.default_rng(0)
    in_shape = (8, 8, 3)
    x = rng.standard_normal((3,) + in_shape)
    y = rng.integers(0, 5, 3)
    ws = [rng.standard_normal(k) * 0.3 for k, _ in co.param_shapes(in_shape, LAYERS)]
    bs = [rng.standard_normal(n) * 0.1 for _, n in co.param_shapes(in_shape, LAYERS)]
    loss, logits, gws, gbs = co.loss_and_grads(x, y, ws, bs, LAYERS)
    assert logits.shape == (3, 5) and loss > 0
    for li in range(len(ws)):
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in ws[li].shape)
            num = co.numeric_grad(x, y, ws, bs, LAYERS, li, idx)
            assert abs(num - gws[li][idx]) <= 1e-6 * max(1.0, abs(num)), (li, idx, num, gws[li][idx])
        j = int(rng.integers(0, bs[li].size))
        num = co.numeric_grad(x, y, ws, bs, LAYERS, li, j, bias=True)
        assert abs(num - gbs[li][j]) <= 1e-6 * max(1.0, abs(num))


def test_flatten_roundtrip_and_step():
    rng = np.random.default_rng(1)
    in_shape = (4, 4, 2)
    layers = (("conv", 3), ("pool",), ("dense", 4))
    shapes = co.param_shapes(in_shape, layers)
    flat = rng.standard_normal(sum(k * c + n for (k, c), n in shapes))
    ws, bs = co.unflatten(flat, in_shape, layers)
    assert np.array_equal(co.flatten(ws, bs), flat)
    x = rng.standard_normal((2,) + in_shape); y = np.array([1, 3])
    l0 = co.loss_and_grads(x, y, ws, bs, layers)[0]
    for _ in range(20):
        ws, bs, _ = co.sgd_step(x, y, ws, bs, layers, 0.2)
    assert co.loss_and_grads(x, y, ws, bs, layers)[0] < l0


def test_round_bf16_is_round_to_nearest_even():
    """oracle.round_bf16 models v_cvt_pk_bf16_f32: 8 significant bits, ties to even, sign preserved, exact values fixed."""
    a = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, -3.14159, 0.0, 256.0, 257.0, 259.0])
    want = np.array([1.0, 1.0, 1.0 + 2.0 ** -6, 1.0 + 2.0 ** -7, -3.140625, 0.0, 256.0, 256.0, 260.0])
    assert np.array_equal(co.round_bf16(a), want)
    r = np.random.default_rng(0).standard_normal(1000)
    q = co.round_bf16(r)
    assert np.all(np.abs(q - r) <= np.abs(r) * 2.0 ** -8) and np.array_equal(co.round_bf16(q), q)


def test_stored_rounding_is_where_the_maps_and_their_gradients_are_written():
    """stored=True (the mirror of RCN_HIPX_BF16_STORED): the convolutional stage's maps and the gradients with respect to them are bf16
    values; with bf16 operands that changes nothing in the forward pass (every consumer rounds the same values again: idempotent) and
    nothing in a weight gradient whose GEMM rounds dZ anyway -- only the first layer's (exact products) and the bias gradients move, by
    bf16 resolution."""
    rng = np.random.default_rng(5)
    in_shape = (8, 8, 3)
    layers = (("conv", 32), ("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 32), ("dense", 10))
    shapes = co.param_shapes(in_shape, layers)
    ws = [rng.standard_normal(k) * np.sqrt(2.0 / k[0]) for k, _ in shapes]
    bs = [rng.standard_normal(n) * 0.1 for _, n in shapes]
    x = rng.standard_normal((4,) + in_shape)
    y = rng.integers(0, 10, 4)
    l0, lg0, gw0, gb0 = co.loss_and_grads(x, y, ws, bs, layers, operand="bf16")
    l1, lg1, gw1, gb1 = co.loss_and_grads(x, y, ws, bs, layers, operand="bf16", stored=True)
    assert l0 == l1 and np.array_equal(lg0, lg1)
    for i in range(1, len(ws)):
        assert np.array_equal(gw0[i], gw1[i]), i
    assert not np.array_equal(gw0[0], gw1[0]) and np.linalg.norm(gw0[0] - gw1[0]) <= 1e-2 * np.linalg.norm(gw0[0])
    for i in range(3):
        assert np.linalg.norm(gb0[i] - gb1[i]) <= 1e-2 * np.linalg.norm(gb0[i])
    # with exact (f64) operands the storage rounding is the only rounding there is: visible, and of bf16 size
    l2 = co.loss_and_grads(x, y, ws, bs, layers)[0]
    l3 = co.loss_and_grads(x, y, ws, bs, layers, stored=True)[0]
    assert l2 != l3 and abs(l2 - l3) <= 2e-2 * abs(l2)
