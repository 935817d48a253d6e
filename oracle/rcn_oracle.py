"""CPU oracle for the rcn hot path -- TEST INFRASTRUCTURE ONLY.

Two things live here:

* ``COracle``  -- ctypes binding of ``oracle/librcn_oracle.so`` (the loop-faithful C restatement,
  ``oracle/rcn_oracle.c``).  This is *the* oracle the GPU parity tests compare against.
* ``np_*``     -- a second, independent, vectorised NumPy restatement written from the same
  reference source.  It exists only to cross-check the C code (two restatements that were written
  differently must agree to f64 rounding) because the Rust reference cannot be built here.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package (``mercer_research_amd``) must never import it.

Parity status: pinned by the reference's data-free KATs (utils/kernel.rs:402-417, :436-441, :421-432);
everything else is "parity unpinned" (no runnable reference, no fixtures in the reference).

Citations are file:line under /root/reference/rcn/src.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PAD_NONE, PAD_SAME = 0, 1                 # utils/kernel.rs:25-28
POOL_AVERAGE, POOL_MAX = 0, 1             # utils/kernel.rs:32-35
OP_TOP, OP_BOTTOM, OP_LEFT, OP_RIGHT = 0, 1, 2, 3   # utils/kernel.rs:16-21
LAYER_CONV, LAYER_POOL = 0, 1             # rcn.rs:35-38
SEP_OPS = (OP_TOP, OP_LEFT, OP_RIGHT, OP_BOTTOM)    # rcn.rs:41-46

#: the default architecture of rcn/src/main.rs:53-59
DEFAULT_LAYERS = ((LAYER_CONV, PAD_SAME), (LAYER_POOL, POOL_MAX), (LAYER_CONV, PAD_SAME), (LAYER_POOL, POOL_MAX))


class OracleError(RuntimeError):
    """Raised where the reference would panic (shape / unsupported)."""

    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: oracle status {code}")
        self.code = code


class _Layer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arg", C.c_int32)]


class _Net(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dims", C.POINTER(C.c_int)),
                ("W", C.POINTER(C.POINTER(C.c_double))), ("b", C.POINTER(C.POINTER(C.c_double)))]


def build_oracle(native: bool = False, out_dir: str | None = None) -> str:
    """Compile the C oracle.  ``native=True`` builds an -O3 -march=native copy (cpu_baseline timing)."""
    src = os.path.join(_HERE, "rcn_oracle.c")
    if not native:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        return os.path.join(_HERE, "librcn_oracle.so")
    out_dir = out_dir or _HERE
    out = os.path.join(out_dir, "librcn_oracle_native.so")
    subprocess.run(["gcc", "-O3", "-march=native", "-fno-fast-math", "-ffp-contract=off", "-fPIC", "-std=c11",
                    "-D_GNU_SOURCE", "-shared", "-o", out, src, "-lm", "-lpthread"], check=True)
    return out


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class COracle:
    """ctypes face of oracle/rcn_oracle.c.  Matrices cross as NumPy arrays in natural (row, col) indexing;
    the column-major conversion nalgebra uses is done here."""

    def __init__(self, path: str | None = None):
        path = path or os.path.join(_HERE, "librcn_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = L = C.CDLL(path)
        dpp = C.POINTER(C.POINTER(C.c_double))
        L.rcn_o_convolve_2d.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_conv_out_shape.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)] * 2
        L.rcn_o_convolve_2d_separated.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_pool_out_shape.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 2
        L.rcn_o_pool_2d.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_relu.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double)]
        L.rcn_o_sobel_separated.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rcn_o_sobel_full.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_get_pixel_matrix.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_feature_len.restype = C.c_long
        L.rcn_o_feature_len.argtypes = [C.c_int, C.c_int, C.POINTER(_Layer), C.c_int]
        L.rcn_o_flatten_feature_set.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(_Layer), C.c_int, C.POINTER(C.c_double)]
        L.rcn_o_gen_scales.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rcn_o_standardize.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_double]
        L.rcn_o_first_layer_fan_in.restype = C.c_long
        L.rcn_o_first_layer_fan_in.argtypes = [C.POINTER(_Layer), C.c_int, C.c_long]
        L.rcn_o_sigmoid.restype = C.c_double
        L.rcn_o_sigmoid.argtypes = [C.c_double]
        L.rcn_o_sigmoid_prime.restype = C.c_double
        L.rcn_o_sigmoid_prime.argtypes = [C.c_double]
        L.rcn_o_classify_test.argtypes = [C.POINTER(_Net), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rcn_o_classify_argmax.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.rcn_o_eval_accept.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.rcn_o_backprop.argtypes = [C.POINTER(_Net), C.POINTER(C.c_double), C.POINTER(C.c_double), dpp, dpp]
        L.rcn_o_train_batch.restype = C.c_double
        L.rcn_o_train_batch.argtypes = [C.POINTER(_Net), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_double]
        L.rcn_o_train_batch_mt.restype = C.c_double
        L.rcn_o_train_batch_mt.argtypes = [C.POINTER(_Net), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int]
        L.rcn_o_batch_gradient.argtypes = [C.POINTER(_Net), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, dpp, dpp, C.POINTER(C.c_double)]

    # -- helpers -----------------------------------------------------------------------------
    @staticmethod
    def _layers(layers: Sequence[Tuple[int, int]]):
        arr = (_Layer * max(1, len(layers)))()
        for i, (k, a) in enumerate(layers):
            arr[i].kind, arr[i].arg = int(k), int(a)
        return arr

    @staticmethod
    def _cm(m) -> np.ndarray:
        """(R,C) array -> column-major flat f64 buffer (nalgebra storage order)."""
        return np.asfortranarray(np.asarray(m, dtype=np.float64)).ravel(order="F").copy()

    class _NetHolder:
        """Keeps the NumPy buffers alive behind an rcn_o_net."""

        def __init__(self, weights: List[np.ndarray], biases: List[np.ndarray]):
            self.L = len(weights)
            dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
            self.dims = (C.c_int * len(dims))(*dims)
            self.Wbuf = [COracle._cm(w) for w in weights]
            self.bbuf = [_f64(b).copy() for b in biases]
            self.Wp = (C.POINTER(C.c_double) * self.L)(*[_dp(w) for w in self.Wbuf])
            self.bp = (C.POINTER(C.c_double) * self.L)(*[_dp(b) for b in self.bbuf])
            self.net = _Net(self.L, self.dims, self.Wp, self.bp)

        def weights(self) -> List[np.ndarray]:
            d = list(self.dims)
            return [self.Wbuf[l].reshape((d[l + 1], d[l]), order="F").copy() for l in range(self.L)]

        def biases(self) -> List[np.ndarray]:
            return [b.copy() for b in self.bbuf]

    def net(self, weights, biases) -> "COracle._NetHolder":
        return COracle._NetHolder([np.asarray(w, dtype=np.float64) for w in weights], list(biases))

    # -- kernel.rs ---------------------------------------------------------------------------
    def sobel_separated(self, op: int):
        c, r = np.zeros(3), np.zeros(3)
        self.lib.rcn_o_sobel_separated(op, _dp(c), _dp(r))
        return c.reshape(3, 1), r.reshape(1, 3)

    def sobel_full(self, op: int) -> np.ndarray:
        k = np.zeros(9)
        self.lib.rcn_o_sobel_full(op, _dp(k))
        return k.reshape((3, 3), order="F").copy()

    def convolve_2d(self, m, kernel, padding: int) -> np.ndarray:
        m = np.asarray(m, dtype=np.float64)
        kernel = np.asarray(kernel, dtype=np.float64)
        (R, Cc), (kr, kc) = m.shape, kernel.shape
        oR, oC = C.c_int(), C.c_int()
        st = self.lib.rcn_o_conv_out_shape(R, Cc, kr, kc, padding, C.byref(oR), C.byref(oC))
        if st != 0:
            raise OracleError(st, "convolve_2d")
        out = np.zeros(oR.value * oC.value)
        st = self.lib.rcn_o_convolve_2d(_dp(self._cm(m)), R, Cc, _dp(self._cm(kernel)), kr, kc, padding, _dp(out))
        if st != 0:
            raise OracleError(st, "convolve_2d")
        return out.reshape((oR.value, oC.value), order="F").copy()

    def convolve_2d_separated(self, m, op: int, padding: int) -> np.ndarray:
        m = np.asarray(m, dtype=np.float64)
        R, Cc = m.shape
        if R < 3 or Cc < 3:
            raise OracleError(-2, "convolve_2d_separated")
        oR, oC = (R, Cc) if padding == PAD_SAME else (R - 2, Cc - 2)
        out = np.zeros(max(1, oR * oC))
        st = self.lib.rcn_o_convolve_2d_separated(_dp(self._cm(m)), R, Cc, op, padding, _dp(out))
        if st != 0:
            raise OracleError(st, "convolve_2d_separated")
        return out[: oR * oC].reshape((oR, oC), order="F").copy()

    def relu(self, m) -> np.ndarray:
        m = _f64(m)
        out = np.zeros_like(m)
        self.lib.rcn_o_relu(_dp(m.ravel()), m.size, _dp(out.reshape(-1)))
        return out

    def pool_2d(self, m, padding: int, pooling: int) -> np.ndarray:
        m = np.asarray(m, dtype=np.float64)
        R, Cc = m.shape
        oR, oC = C.c_int(), C.c_int()
        st = self.lib.rcn_o_pool_out_shape(R, Cc, padding, C.byref(oR), C.byref(oC))
        if st != 0:
            raise OracleError(st, "pool_2d")
        out = np.zeros(oR.value * oC.value)
        st = self.lib.rcn_o_pool_2d(_dp(self._cm(m)), R, Cc, padding, pooling, _dp(out))
        if st != 0:
            raise OracleError(st, "pool_2d")
        return out.reshape((oR.value, oC.value), order="F").copy()

    # -- lib.rs / rcn.rs features ---------------------------------------------------------------
    def get_pixel_matrix(self, img_u8: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img_u8, dtype=np.uint8)
        H, W = img.shape
        out = np.zeros(H * W)
        self.lib.rcn_o_get_pixel_matrix(img.ctypes.data_as(C.POINTER(C.c_uint8)), H, W, _dp(out))
        return out.reshape((H, W), order="F").copy()

    def feature_len(self, H: int, W: int, layers) -> int:
        n = self.lib.rcn_o_feature_len(H, W, self._layers(layers), len(layers))
        if n < 0:
            raise OracleError(int(n), "feature_len")
        return int(n)

    def flatten_feature_set(self, m, layers) -> np.ndarray:
        m = np.asarray(m, dtype=np.float64)
        H, W = m.shape
        n = self.feature_len(H, W, layers)
        out = np.zeros(max(1, n))
        st = self.lib.rcn_o_flatten_feature_set(_dp(self._cm(m)), H, W, self._layers(layers), len(layers), _dp(out))
        if st != 0:
            raise OracleError(st, "flatten_feature_set")
        return out[:n].copy()

    def features(self, imgs_u8: np.ndarray, layers) -> np.ndarray:
        """[N,H,W] u8 -> [N,F] f64: get_pixel_matrix + flatten_feature_set per image (rcn.rs:399-401)."""
        imgs = np.asarray(imgs_u8, dtype=np.uint8)
        return np.stack([self.flatten_feature_set(self.get_pixel_matrix(i), layers) for i in imgs]) if len(imgs) else np.zeros((0, 0))

    def gen_scales(self, feats) -> Tuple[float, float]:
        f = _f64(feats)
        mean, sd = C.c_double(), C.c_double()
        self.lib.rcn_o_gen_scales(_dp(f.reshape(-1)), f.shape[0], f.shape[1], C.byref(mean), C.byref(sd))
        return mean.value, sd.value

    def standardize(self, feats, mean: float, sd: float) -> np.ndarray:
        f = _f64(feats).copy()
        self.lib.rcn_o_standardize(_dp(f.reshape(-1)), f.size, mean, sd)
        return f

    def first_layer_fan_in(self, layers, l: int) -> int:
        return int(self.lib.rcn_o_first_layer_fan_in(self._layers(layers), len(layers), l))

    # -- rcn.rs dense ------------------------------------------------------------------------------
    def sigmoid(self, x: float) -> float:
        return self.lib.rcn_o_sigmoid(float(x))

    def sigmoid_prime(self, x: float) -> float:
        return self.lib.rcn_o_sigmoid_prime(float(x))

    def classify_test(self, weights, biases, X) -> np.ndarray:
        h = self.net(weights, biases)
        X = np.atleast_2d(_f64(X))
        out = np.zeros((X.shape[0], h.dims[h.L]))
        for i in range(X.shape[0]):
            self.lib.rcn_o_classify_test(C.byref(h.net), _dp(X[i]), _dp(out[i]))
        return out

    def classify_argmax(self, out) -> int:
        o = _f64(out)
        return int(self.lib.rcn_o_classify_argmax(_dp(o), o.size))

    def eval_accept(self, out, expect) -> int:
        o, e = _f64(out), _f64(expect)
        return int(self.lib.rcn_o_eval_accept(_dp(o), _dp(e), o.size))

    def _grad_bufs(self, h):
        d = list(h.dims)
        gW = [np.zeros(d[l] * d[l + 1]) for l in range(h.L)]
        gb = [np.zeros(d[l + 1]) for l in range(h.L)]
        gWp = (C.POINTER(C.c_double) * h.L)(*[_dp(g) for g in gW])
        gbp = (C.POINTER(C.c_double) * h.L)(*[_dp(g) for g in gb])
        return d, gW, gb, gWp, gbp

    def backprop(self, weights, biases, x, y):
        h = self.net(weights, biases)
        d, gW, gb, gWp, gbp = self._grad_bufs(h)
        x, y = _f64(x), _f64(y)
        self.lib.rcn_o_backprop(C.byref(h.net), _dp(x), _dp(y), gWp, gbp)
        return [g.reshape((d[l + 1], d[l]), order="F").copy() for l, g in enumerate(gW)], gb

    def batch_gradient(self, weights, biases, X, Y):
        """Sum over samples (index order) of per-sample gradients, and the quadratic cost."""
        h = self.net(weights, biases)
        d, gW, gb, gWp, gbp = self._grad_bufs(h)
        X, Y = _f64(X), _f64(Y)
        cost = C.c_double()
        self.lib.rcn_o_batch_gradient(C.byref(h.net), _dp(X.reshape(-1)), _dp(Y.reshape(-1)), X.shape[0], gWp, gbp, C.byref(cost))
        return [g.reshape((d[l + 1], d[l]), order="F").copy() for l, g in enumerate(gW)], gb, cost.value

    def train_batch(self, weights, biases, X, Y, eta: float, threads: int = 0):
        """One SGD step; returns (new_weights, new_biases, cost_before_update)."""
        h = self.net(weights, biases)
        X, Y = _f64(X), _f64(Y)
        if threads > 0:
            cost = self.lib.rcn_o_train_batch_mt(C.byref(h.net), _dp(X.reshape(-1)), _dp(Y.reshape(-1)), X.shape[0], eta, threads)
        else:
            cost = self.lib.rcn_o_train_batch(C.byref(h.net), _dp(X.reshape(-1)), _dp(Y.reshape(-1)), X.shape[0], eta)
        return h.weights(), h.biases(), cost

    def train_steps_inplace(self, holder: "COracle._NetHolder", X, Y, B: int, steps: int, eta: float, threads: int = 0) -> int:
        """Run `steps` consecutive minibatch steps over X/Y (sample-major, contiguous batches); returns images seen."""
        X, Y = _f64(X), _f64(Y)
        F, Cc = X.shape[1], Y.shape[1]
        nb = X.shape[0] // B
        fn = self.lib.rcn_o_train_batch_mt if threads > 0 else self.lib.rcn_o_train_batch
        for s in range(steps):
            o = (s % nb) * B
            xp = C.cast(C.c_void_p(X.ctypes.data + o * F * 8), C.POINTER(C.c_double))
            yp = C.cast(C.c_void_p(Y.ctypes.data + o * Cc * 8), C.POINTER(C.c_double))
            if threads > 0:
                fn(C.byref(holder.net), xp, yp, B, eta, threads)
            else:
                fn(C.byref(holder.net), xp, yp, B, eta)
        return steps * B


# =================================================================================================
# Independent NumPy restatement (vectorised; written separately from the C code on purpose)
# =================================================================================================

def np_sobel_separated(op: int):
    """utils/kernel.rs:47-52"""
    one21 = np.array([1.0, 2.0, 1.0])
    return {
        OP_TOP: (np.array([1.0, 0.0, -1.0]).reshape(3, 1), one21.reshape(1, 3)),
        OP_BOTTOM: (np.array([-1.0, 0.0, 1.0]).reshape(3, 1), one21.reshape(1, 3)),
        OP_LEFT: (one21.reshape(3, 1), np.array([1.0, 0.0, -1.0]).reshape(1, 3)),
        OP_RIGHT: (one21.reshape(3, 1), np.array([-1.0, 0.0, 1.0]).reshape(1, 3)),
    }[op]


def np_convolve_2d(m: np.ndarray, k: np.ndarray, padding: int) -> np.ndarray:
    """utils/kernel.rs:110-194 as shifted-slice accumulation (ky outer, kx inner, like the source)."""
    m = np.asarray(m, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    (R, Cc), (kr, kc) = m.shape, k.shape
    if kr == 0 or kc == 0 or kr > R or kc > Cc:
        raise OracleError(-2, "np_convolve_2d")
    if padding == PAD_SAME:
        if kr % 2 == 0 or kc % 2 == 0:
            raise OracleError(-2, "np_convolve_2d")
        pr, pc = kr // 2, kc // 2
        if pr >= 2 or pc >= 2:  # copy loop runs off the source matrix -> bounds panic
            if R + pr - 1 >= 1 and Cc + pc - 1 >= 1:
                raise OracleError(-2, "np_convolve_2d")
        P = np.zeros((R + 2 * pr, Cc + 2 * pc))
        # kernel.rs:154-158: rows 1..R+pr-1, cols 1..C+pc-1 take m[cy-1, cx-1]
        P[1:R + pr, 1:Cc + pc] = m[0:R + pr - 1, 0:Cc + pc - 1]
        oR, oC, src = R, Cc, P
    else:
        oR, oC, src = R - kr + 1, Cc - kc + 1, m
    out = np.zeros((oR, oC))
    for ky in range(kr):
        for kx in range(kc):
            out = out + src[ky:ky + oR, kx:kx + oC] * k[ky, kx]
    return out


def np_convolve_2d_separated(m, op: int, padding: int) -> np.ndarray:
    """utils/kernel.rs:196-207"""
    m = np.asarray(m, dtype=np.float64)
    if m.shape[0] < 3 or m.shape[1] < 3:
        raise OracleError(-2, "np_convolve_2d_separated")
    col, row = np_sobel_separated(op)
    t = np_convolve_2d(np_convolve_2d(m, col, padding), row, padding)
    return np.where(t >= 0, t, 0.0)


def np_pool_2d(m, padding: int, pooling: int) -> np.ndarray:
    """utils/kernel.rs:245-349"""
    m = np.asarray(m, dtype=np.float64)
    R, Cc = m.shape
    if R < 2 or Cc < 2:
        raise OracleError(-2, "np_pool_2d")
    if pooling != POOL_MAX:
        raise OracleError(-3, "np_pool_2d")
    if padding == PAD_SAME:
        m = np.pad(m, ((0, R % 2), (0, Cc % 2)))
    else:
        m = m[: R - R % 2, : Cc - Cc % 2]
    r2, c2 = m.shape[0] // 2, m.shape[1] // 2
    return m.reshape(r2, 2, c2, 2).max(axis=(1, 3))


def np_flatten_feature_set(m, layers) -> np.ndarray:
    """rcn.rs:317-356 (map order Q3, column-major flatten Q4)."""
    fs: List[np.ndarray] = []
    for kind, arg in layers:
        if kind == LAYER_CONV:
            if fs:
                n = len(fs)
                for i in range(n):
                    src = fs[i]
                    for op in SEP_OPS[:-1]:
                        fs.append(np_convolve_2d_separated(src, op, arg))
                    fs[i] = np_convolve_2d_separated(src, SEP_OPS[-1], arg)
            else:
                fs = [np_convolve_2d_separated(m, op, arg) for op in SEP_OPS]
        else:
            fs = [np_pool_2d(f, PAD_SAME, arg) for f in fs]
    if not fs:
        return np.zeros(0)
    return np.concatenate([f.ravel(order="F") for f in fs])


def np_gen_scales(feats) -> Tuple[float, float]:
    """rcn.rs:230-251 (population mean / sd over every feature of every sample)."""
    f = np.asarray(feats, dtype=np.float64)
    mean = f.sum() / f.size
    return float(mean), float(np.sqrt(((f - mean) ** 2).sum() / f.size))


def np_standardize(feats, mean, sd):
    d = (np.asarray(feats, dtype=np.float64) - mean) / sd
    return np.where(d >= 0, d, 0.0)


def np_sigmoid(z):
    return 1.0 / (1.0 + np.power(np.e, -np.asarray(z, dtype=np.float64)))


def np_forward(weights, biases, X):
    """rcn.rs:105-116, batched: returns list of activations [X, a1, ..., aL] (sample-major)."""
    acts = [np.asarray(X, dtype=np.float64)]
    for w, b in zip(weights, biases):
        acts.append(np_sigmoid(acts[-1] @ np.asarray(w).T + np.asarray(b)))
    return acts


def np_batch_gradient(weights, biases, X, Y):
    """rcn.rs:260-314 summed over the batch as GEMMs (the formulation the GPU path uses)."""
    acts = np_forward(weights, biases, X)
    Y = np.asarray(Y, dtype=np.float64)
    L = len(weights)
    gW, gb = [None] * L, [None] * L
    delta = (acts[L] - Y) * acts[L] * (1.0 - acts[L])
    for l in range(L - 1, -1, -1):
        gW[l] = delta.T @ acts[l]
        gb[l] = delta.sum(axis=0)
        if l > 0:
            delta = (delta @ np.asarray(weights[l])) * acts[l] * (1.0 - acts[l])
    cost = float(((acts[L] - Y) ** 2).sum() / (2.0 * X.shape[0]))
    return gW, gb, cost


def np_train_batch(weights, biases, X, Y, eta):
    """rcn.rs:176-223"""
    gW, gb, cost = np_batch_gradient(weights, biases, X, Y)
    s = eta / X.shape[0]
    return [np.asarray(w) - s * g for w, g in zip(weights, gW)], [np.asarray(b) - s * g for b, g in zip(biases, gb)], cost


def np_numeric_gradient(weights, biases, x, y, eps=1e-6):
    """Central finite differences of C = 1/2 ||a_L - y||^2 for ONE sample (checks backprop)."""
    def cost(ws, bs):
        a = np_forward(ws, bs, x[None, :])[-1][0]
        return 0.5 * float(((a - y) ** 2).sum())
    gW = [np.zeros_like(np.asarray(w, dtype=np.float64)) for w in weights]
    gb = [np.zeros_like(np.asarray(b, dtype=np.float64)) for b in biases]
    ws = [np.array(w, dtype=np.float64) for w in weights]
    bs = [np.array(b, dtype=np.float64) for b in biases]
    for l in range(len(ws)):
        for idx in np.ndindex(*ws[l].shape):
            o = ws[l][idx]
            ws[l][idx] = o + eps; cp = cost(ws, bs)
            ws[l][idx] = o - eps; cm = cost(ws, bs)
            ws[l][idx] = o
            gW[l][idx] = (cp - cm) / (2 * eps)
        for i in range(bs[l].size):
            o = bs[l][i]
            bs[l][i] = o + eps; cp = cost(ws, bs)
            bs[l][i] = o - eps; cm = cost(ws, bs)
            bs[l][i] = o
            gb[l][i] = (cp - cm) / (2 * eps)
    return gW, gb


# =================================================================================================
# Synthetic workload of SURVEY.md §8(d) / BASELINE.md §2 (shared by tests and bench so both see the same data)
# =================================================================================================

def synthetic_images(n: int, h: int = 28, w: int = 28, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """MNIST-like u8 images: 4-px zero border, ~19 % non-zero pixels overall; labels uniform 0..9."""
    rng = np.random.default_rng(seed)
    imgs = np.zeros((n, h, w), dtype=np.uint8)
    b = 4 if min(h, w) > 12 else 0
    ih, iw = h - 2 * b, w - 2 * b
    frac = 0.19 * (h * w) / (ih * iw)
    vals = rng.integers(1, 256, size=(n, ih, iw), dtype=np.uint16).astype(np.uint8)
    mask = rng.random((n, ih, iw)) < frac
    imgs[:, b:h - b, b:w - b] = np.where(mask, vals, 0)
    labels = rng.integers(0, 10, size=n)
    return imgs, labels.astype(np.int32)


def synthetic_params(dims: Sequence[int], seed: int = 42):
    """N(0,1) weights (out x in) and biases in the reference's shapes/order (rcn.rs:500-523)."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(len(dims) - 1):
        # drawn in column-major order like DMatrix::from_iterator (rcn.rs:504-510)
        ws.append(rng.standard_normal(dims[i] * dims[i + 1]).reshape((dims[i + 1], dims[i]), order="F"))
        bs.append(rng.standard_normal(dims[i + 1]))
    return ws, bs


def one_hot(labels, classes: int = 10) -> np.ndarray:
    y = np.zeros((len(labels), classes))
    y[np.arange(len(labels)), np.asarray(labels)] = 1.0
    return y
