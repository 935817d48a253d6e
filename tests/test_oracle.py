"""CPU tests of the oracle itself (no GPU): reference KATs, cross-restatement agreement,
hand-derivable quirk cases, finite-difference gradient check, golden vectors."""
import os

import numpy as np
import pytest

from oracle.rcn_oracle import (DEFAULT_LAYERS, LAYER_CONV, LAYER_POOL, OP_BOTTOM, OP_LEFT, OP_RIGHT, OP_TOP, PAD_NONE,
                               PAD_SAME, POOL_AVERAGE, POOL_MAX, OracleError, np_batch_gradient, np_convolve_2d,
                               np_convolve_2d_separated, np_flatten_feature_set, np_gen_scales, np_numeric_gradient,
                               np_pool_2d, np_sobel_separated, np_standardize, np_train_batch, one_hot,
                               synthetic_images, synthetic_params)


# ------------------------------------------------------------------ reference KATs (the only pins the reference has)

def test_kat_convolve_2d_padding_same(oracle):
    """utils/kernel.rs:436-441: 30x30 0..900 row-major (x) 3x3 identity under Same == itself."""
    m = np.arange(900, dtype=np.float64).reshape(30, 30)
    k = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float64)
    assert np.array_equal(oracle.convolve_2d(m, k, PAD_SAME), m)
    assert np.array_equal(np_convolve_2d(m, k, PAD_SAME), m)


def test_kat_verify_separated_sobels(oracle):
    """utils/kernel.rs:402-417: col (x) row of every separated pair == the 3x3 constant (:56-59)."""
    expect = {
        OP_TOP: [[1, 2, 1], [0, 0, 0], [-1, -2, -1]],
        OP_BOTTOM: [[-1, -2, -1], [0, 0, 0], [1, 2, 1]],
        OP_LEFT: [[1, 0, -1], [2, 0, -2], [1, 0, -1]],
        OP_RIGHT: [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]],
    }
    for op, full in expect.items():
        c, r = oracle.sobel_separated(op)
        assert np.array_equal(c @ r, np.array(full, dtype=np.float64))
        assert np.array_equal(oracle.sobel_full(op), np.array(full, dtype=np.float64))
        c2, r2 = np_sobel_separated(op)
        assert np.array_equal(c2 @ r2, np.array(full, dtype=np.float64))


def test_kat_validate_padding_calc():
    """utils/kernel.rs:421-432 (shape arithmetic only)."""
    m, k = (28, 28), (3, 3)
    reg = (m[0] - k[0] + 1, m[1] - k[1] + 1)
    pad = (m[0] - reg[0], m[1] - reg[1])
    assert pad[0] < k[0] and pad[1] < k[1]


# ------------------------------------------------------------------ two restatements agree

@pytest.mark.parametrize("shape", [(3, 3), (5, 6), (7, 3), (14, 14), (28, 28), (9, 31)])
def test_c_vs_numpy_operators(oracle, shape):
    rng = np.random.default_rng(sum(shape))
    m = rng.integers(0, 256, shape).astype(np.float64)
    for pad in (PAD_NONE, PAD_SAME):
        for op in range(4):
            assert np.array_equal(oracle.convolve_2d_separated(m, op, pad), np_convolve_2d_separated(m, op, pad))
        for ks in ((3, 3), (1, 3), (3, 1), (1, 1)):
            k = rng.standard_normal(ks)
            np.testing.assert_allclose(oracle.convolve_2d(m, k, pad), np_convolve_2d(m, k, pad), rtol=1e-13, atol=1e-10)
        assert np.array_equal(oracle.pool_2d(m, pad, POOL_MAX), np_pool_2d(m, pad, POOL_MAX))


def test_c_vs_numpy_features_and_scales(oracle):
    imgs, _ = synthetic_images(5, seed=3)
    f = oracle.features(imgs, DEFAULT_LAYERS)
    f2 = np.stack([np_flatten_feature_set(i.astype(np.float64), DEFAULT_LAYERS) for i in imgs])
    assert f.shape == (5, 784) and np.array_equal(f, f2)
    m, s = oracle.gen_scales(f)
    m2, s2 = np_gen_scales(f)
    assert abs(m - m2) <= 1e-12 * abs(m) and abs(s - s2) <= 1e-12 * abs(s)
    np.testing.assert_allclose(oracle.standardize(f, m, s), np_standardize(f, m, s), rtol=1e-15, atol=0)


@pytest.mark.parametrize("dims,B", [([784, 30, 10], 10), ([784, 10, 10, 10], 4), ([12, 5, 3], 1)])
def test_c_vs_numpy_dense(oracle, dims, B):
    rng = np.random.default_rng(B)
    ws, bs = synthetic_params(dims, seed=B)
    X = np.maximum(rng.standard_normal((B, dims[0])), 0)
    Y = one_hot(rng.integers(0, dims[-1], B), dims[-1])
    gW, gb, cost = oracle.batch_gradient(ws, bs, X, Y)
    gW2, gb2, cost2 = np_batch_gradient(ws, bs, X, Y)
    assert abs(cost - cost2) <= 1e-12 * max(1.0, abs(cost))
    for a, b in zip(gW + gb, gW2 + gb2):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    nw, nb, _ = oracle.train_batch(ws, bs, X, Y, 3.0)
    nw2, nb2, _ = np_train_batch(ws, bs, X, Y, 3.0)
    for a, b in zip(nw + nb, nw2 + nb2):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------ hand-derivable quirk cases

def test_q1_same_padding_quirk_3x1_and_1x3(oracle):
    """kernel.rs:154-158: offset hard-coded to 1 on both axes.  3x1: column 0 zero, image shifted right,
    last column dropped; 1x3: row 0 zero, shifted down, last row dropped; 3x3: no shift."""
    rng = np.random.default_rng(1)
    m = rng.integers(1, 9, (6, 7)).astype(np.float64)
    ident_col = np.array([[0.0], [1.0], [0.0]])
    out = oracle.convolve_2d(m, ident_col, PAD_SAME)
    assert np.all(out[:, 0] == 0) and np.array_equal(out[:, 1:], m[:, :-1])
    ident_row = np.array([[0.0, 1.0, 0.0]])
    out = oracle.convolve_2d(m, ident_row, PAD_SAME)
    assert np.all(out[0, :] == 0) and np.array_equal(out[1:, :], m[:-1, :])
    # 1x1 Same: both shifts at once
    out = oracle.convolve_2d(m, np.array([[1.0]]), PAD_SAME)
    assert np.all(out[0, :] == 0) and np.all(out[:, 0] == 0) and np.array_equal(out[1:, 1:], m[:-1, :-1])


def test_q1_separated_same_equals_shifted_full_sobel_except_last_column(oracle):
    rng = np.random.default_rng(2)
    m = rng.integers(0, 256, (10, 12)).astype(np.float64)
    for op in range(4):
        sep = oracle.convolve_2d_separated(m, op, PAD_SAME)
        full = np.maximum(oracle.convolve_2d(m, oracle.sobel_full(op), PAD_SAME), 0)
        assert np.all(sep[0, :] == 0)                                  # row 0 is all zero
        assert np.array_equal(sep[1:, 1:-1], full[:-1, :-2])        # shifted by (+1,+1)
        # column 0 is NOT zero (SURVEY Q1 over-states this): only the kx=2 tap survives there
        col, row = oracle.sobel_separated(op)
        mp = np.pad(m, ((1, 1), (0, 0)))
        c0 = row[0, 2] * sum(col[ky, 0] * mp[ky:ky + 10, 0] for ky in range(3))
        assert np.array_equal(sep[1:, 0], np.maximum(c0[:-1], 0))
        mz = m.copy(); mz[:, -1] = 0                                   # last output column sees a zeroed last image column
        fullz = np.maximum(oracle.convolve_2d(mz, oracle.sobel_full(op), PAD_SAME), 0)
        assert np.array_equal(sep[1:, -1], fullz[:-1, -2])


def test_q2_cross_correlation_not_convolution(oracle):
    m = np.zeros((5, 5)); m[2, 2] = 1.0
    k = np.arange(9, dtype=np.float64).reshape(3, 3)
    out = oracle.convolve_2d(m, k, PAD_NONE)
    assert np.array_equal(out, k[::-1, ::-1])   # impulse response of a correlation is the flipped kernel


def test_padding_same_panics(oracle):
    m = np.ones((8, 8))
    for ks in ((2, 2), (2, 3), (5, 5), (5, 1), (1, 5)):   # even -> explicit panic :131-135; >=5 -> index panic :156
        with pytest.raises(OracleError):
            oracle.convolve_2d(m, np.ones(ks), PAD_SAME)
        with pytest.raises(OracleError):
            np_convolve_2d(m, np.ones(ks), PAD_SAME)
    with pytest.raises(OracleError):
        oracle.convolve_2d(np.ones((2, 2)), np.ones((3, 3)), PAD_NONE)     # kernel larger than target :123-128
    with pytest.raises(OracleError):
        oracle.convolve_2d_separated(np.ones((2, 8)), OP_TOP, PAD_SAME)    # :199-201
    with pytest.raises(OracleError):
        oracle.pool_2d(np.ones((1, 8)), PAD_SAME, POOL_MAX)                # :246-251
    with pytest.raises(OracleError):
        oracle.pool_2d(np.ones((4, 4)), PAD_SAME, POOL_AVERAGE)            # :283-285


def test_q6_pool_odd_dims(oracle):
    m = -np.arange(1, 16, dtype=np.float64).reshape(3, 5)   # all negative: zero padding wins in padded cells
    same = oracle.pool_2d(m, PAD_SAME, POOL_MAX)
    assert same.shape == (2, 3)
    assert same[0, 0] == -1 and same[0, 2] == 0 and same[1, 0] == 0 and same[1, 2] == 0
    none = oracle.pool_2d(m, PAD_NONE, POOL_MAX)
    assert none.shape == (1, 2) and none[0, 0] == -1 and none[0, 1] == -3


def test_q3_q4_feature_map_order_and_column_major_flatten(oracle):
    """rcn.rs:325-339 + :350-355.  Build the expected vector by hand from the operator calls."""
    rng = np.random.default_rng(5)
    m = rng.integers(0, 256, (28, 28)).astype(np.float64)
    sep = lambda x, op: oracle.convolve_2d_separated(x, op, PAD_SAME)
    pool = lambda x: oracle.pool_2d(x, PAD_SAME, POOL_MAX)
    f = [pool(sep(m, op)) for op in (OP_TOP, OP_LEFT, OP_RIGHT, OP_BOTTOM)]
    second = [sep(f[i], OP_BOTTOM) for i in range(4)]
    for i in range(4):
        second += [sep(f[i], OP_TOP), sep(f[i], OP_LEFT), sep(f[i], OP_RIGHT)]
    expect = np.concatenate([pool(x).ravel(order="F") for x in second])
    got = oracle.flatten_feature_set(m, DEFAULT_LAYERS)
    assert got.shape == (784,) and np.array_equal(got, expect)
    assert oracle.feature_len(28, 28, DEFAULT_LAYERS) == 784


def test_feature_edge_cases(oracle):
    m = np.ones((9, 11))
    # pool before any conv is a no-op on an empty feature_set (rcn.rs:343)
    a = oracle.flatten_feature_set(m, ((LAYER_POOL, POOL_MAX), (LAYER_CONV, PAD_SAME)))
    b = oracle.flatten_feature_set(m, ((LAYER_CONV, PAD_SAME),))
    assert np.array_equal(a, b) and a.size == 4 * 99
    assert oracle.flatten_feature_set(m, ()).size == 0
    assert oracle.flatten_feature_set(m, ((LAYER_POOL, POOL_MAX),)).size == 0
    with pytest.raises(OracleError):
        oracle.flatten_feature_set(np.ones((2, 9)), ((LAYER_CONV, PAD_SAME),))
    # fan-in formula rcn.rs:443 (integer division, left to right)
    assert oracle.first_layer_fan_in(DEFAULT_LAYERS, 784) == 784
    assert oracle.first_layer_fan_in(((LAYER_CONV, PAD_SAME),), 3136) == 4 * 3136
    assert oracle.first_layer_fan_in(((LAYER_POOL, POOL_MAX),), 100) == 0


def test_q7_standardize_clamps_at_zero(oracle):
    f = np.array([[0.0, 10.0, 20.0, 30.0]])
    m, s = oracle.gen_scales(f)
    assert m == 15.0 and abs(s - np.sqrt(125.0)) < 1e-15
    out = oracle.standardize(f, m, s)
    assert out[0, 0] == 0 and out[0, 1] == 0 and out[0, 2] > 0


def test_q9_eval_and_argmax_tie_semantics(oracle):
    assert oracle.classify_argmax([0.1, 0.9, 0.9, 0.2]) == 2            # max_by keeps the LAST max (rcn.rs:92-97)
    assert oracle.eval_accept([0.1, 0.9, 0.3], [0, 1, 0]) == 1
    assert oracle.eval_accept([0.9, 0.9, 0.3], [0, 1, 0]) == 0          # ties give two 1s -> mismatch (rcn.rs:155)
    assert oracle.eval_accept([0.1, 0.2, 0.3], [0, 1, 0]) == 0


# ------------------------------------------------------------------ gradient semantics

def test_backprop_matches_finite_differences(oracle):
    rng = np.random.default_rng(9)
    for dims in ([12, 5, 3], [10, 6, 4, 3]):
        ws, bs = synthetic_params(dims, seed=3)
        ws = [w * 0.5 for w in ws]
        x = rng.random(dims[0]); y = one_hot([1], dims[-1])[0]
        gW, gb = oracle.backprop(ws, bs, x, y)
        nW, nb = np_numeric_gradient(ws, bs, x, y)
        for a, b in zip(gW + gb, nW + nb):
            assert np.abs(a - b).max() <= 1e-6 * max(1e-3, np.abs(b).max())


def test_train_batch_invariants(oracle):
    rng = np.random.default_rng(4)
    dims = [30, 7, 5]
    ws, bs = synthetic_params(dims, seed=8)
    X = rng.random((6, 30)); Y = one_hot(rng.integers(0, 5, 6), 5)
    nw, nb, _ = oracle.train_batch(ws, bs, X, Y, 3.0)
    perm = rng.permutation(6)
    pw, pb, _ = oracle.train_batch(ws, bs, X[perm], Y[perm], 3.0)
    for a, b in zip(nw + nb, pw + pb):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)       # sample order only changes f64 rounding
    # B = 1 equals W - eta * grad (rcn.rs:214 with batch.len() == 1)
    gW, gb = oracle.backprop(ws, bs, X[0], Y[0])
    ow, ob, _ = oracle.train_batch(ws, bs, X[:1], Y[:1], 3.0)
    for l in range(2):
        assert np.array_equal(ow[l], ws[l] - 3.0 * gW[l]) and np.array_equal(ob[l], bs[l] - 3.0 * gb[l])
    # threaded variant (the cpu_baseline code path) computes the same step
    tw, tb, _ = oracle.train_batch(ws, bs, X, Y, 3.0, threads=3)
    for a, b in zip(nw + nb, tw + tb):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def test_sigmoid_form(oracle):
    for x in (-30.0, -1.5, 0.0, 0.3, 12.0):
        assert oracle.sigmoid(x) == 1.0 / (1.0 + np.e ** (-x))
        s = oracle.sigmoid(x)
        assert oracle.sigmoid_prime(x) == s * (1.0 - s)


# ------------------------------------------------------------------ committed golden vectors

def test_golden_operators(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "operators.npz"))
    for name in ("a5x6", "b28x28", "c7x3", "d9x9"):
        m = g[f"{name}_m"]
        for pad in (PAD_NONE, PAD_SAME):
            for op in range(4):
                assert np.array_equal(oracle.convolve_2d_separated(m, op, pad), g[f"{name}_sep_p{pad}_op{op}"])
                assert np.array_equal(np_convolve_2d_separated(m, op, pad), g[f"{name}_sep_p{pad}_op{op}"])
            assert np.array_equal(oracle.pool_2d(m, pad, POOL_MAX), g[f"{name}_pool_p{pad}"])
            assert np.array_equal(oracle.convolve_2d(m, g[f"{name}_k33"], pad), g[f"{name}_conv33_p{pad}"])


def test_golden_features(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "features.npz"))
    f = oracle.features(g["imgs"], DEFAULT_LAYERS)
    assert np.array_equal(f, g["feats"])
    m, s = oracle.gen_scales(f)
    assert m == float(g["mean"]) and s == float(g["sd"])
    assert np.array_equal(oracle.standardize(f, m, s), g["std"])
    for k in ("conv_none_pool", "conv_conv_pool", "pool_first"):
        lay = [tuple(int(v) for v in r) for r in g[f"layers_{k}"]]
        assert np.array_equal(oracle.features(g["small"], lay), g[f"small_{k}"])
        np2 = np.stack([np_flatten_feature_set(i.astype(np.float64), lay) for i in g["small"]])
        assert np.array_equal(np2, g[f"small_{k}"])


def test_golden_dense(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "dense.npz"))
    for name in ("tiny", "mid3", "one"):
        dims = list(g[f"{name}_dims"]); L = len(dims) - 1
        ws = [g[f"{name}_W{l}"] for l in range(L)]; bs = [g[f"{name}_b{l}"] for l in range(L)]
        X, Y, eta = g[f"{name}_X"], g[f"{name}_Y"], float(g[f"{name}_eta"])
        gW, gb, cost = oracle.batch_gradient(ws, bs, X, Y)
        nw, nb, _ = oracle.train_batch(ws, bs, X, Y, eta)
        assert cost == float(g[f"{name}_cost"])
        assert np.array_equal(oracle.classify_test(ws, bs, X), g[f"{name}_out"])
        for l in range(L):
            assert np.array_equal(gW[l], g[f"{name}_gW{l}"]) and np.array_equal(gb[l], g[f"{name}_gb{l}"])
            assert np.array_equal(nw[l], g[f"{name}_nW{l}"]) and np.array_equal(nb[l], g[f"{name}_nb{l}"])
        # the NumPy restatement (GEMM formulation) lands on the same numbers to f64 rounding
        nw2, nb2, cost2 = np_train_batch(ws, bs, X, Y, eta)
        assert abs(cost2 - cost) < 1e-13
        for l in range(L):
            np.testing.assert_allclose(nw2[l], nw[l], rtol=1e-11, atol=1e-13)
    imgs, labels = synthetic_images(32, seed=11)
    f = oracle.features(imgs, DEFAULT_LAYERS)
    m, s = oracle.gen_scales(f)
    X, Y = oracle.standardize(f, m, s), one_hot(labels)
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    nw, nb, cost = oracle.train_batch(ws, bs, X, Y, 3.0)
    assert cost == float(g["mnist_cost"])
    assert np.array_equal(nb[0], g["mnist_nb0"]) and np.array_equal(nb[1], g["mnist_nb1"])
    assert np.array_equal(nw[1], g["mnist_nW1"])
    assert np.array_equal(nw[0].ravel(order="F")[::37], g["mnist_nW0_strided"])


def test_c_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The C restatement, compiled with -fsanitize=address,undefined, runs the whole path (features, scales, standardise, two
    train_batch steps, forward, the operator entry points) without a report.  Sanitizers run on the CPU build only: the GPU
    pool has none.  A subprocess, because the sanitizer runtime has to be loaded before the interpreter."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no libasan in this toolchain")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "librcn_oracle_asan.so"
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-ffp-contract=off",
                    "-fPIC", "-shared", "-pthread", "-o", str(lib), os.path.join(root, "oracle", "rcn_oracle.c"), "-lm"], check=True)
    script = tmp_path / "run.py"
    script.write_text(f"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from oracle.rcn_oracle import COracle, DEFAULT_LAYERS, one_hot, synthetic_images, synthetic_params
o = COracle({str(lib)!r})
imgs, labels = synthetic_images(64, seed=2)
f = o.features(imgs, DEFAULT_LAYERS)
m, s = o.gen_scales(f)
x = o.standardize(f, m, s)
ws, bs = synthetic_params([784, 30, 10], seed=1)
ws = [w * 0.1 for w in ws]
y = one_hot(labels)
for j in range(2):
    ws, bs, c = o.train_batch(ws, bs, x[j * 32:(j + 1) * 32], y[j * 32:(j + 1) * 32], 3.0)
o.classify_test(ws, bs, x[:4])
mat = imgs[0].astype(np.float64)
o.convolve_2d(mat, o.sobel_full(0), 1); o.convolve_2d_separated(mat, 2, 0); o.pool_2d(mat[:27, :27], 1, 1)
print("ok", c)
""")
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-500:] + r.stderr[-2000:]
