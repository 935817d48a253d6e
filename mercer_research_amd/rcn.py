"""Host-side mirror of the reference's Rust API (`rcn::rcn::RCN`, `rcn::utils::kernel::{Convolve2D, Pool2D}`)
on top of the gfx950 C-ABI library.  Same names, argument meaning and error behaviour as the reference so that
the parity tests read like the reference's own tests; all arithmetic happens in the HIP kernels.

Reference citations are file:line under /root/reference/rcn/src.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import F32, F64, RcnHipError, RcnPanic  # noqa: F401  (re-exported)


class Padding(enum.IntEnum):          # utils/kernel.rs:25-28
    NONE = 0
    SAME = 1


class Pooling(enum.IntEnum):          # utils/kernel.rs:32-35
    AVERAGE = 0
    MAX = 1


class SeparableOperator(enum.IntEnum):  # utils/kernel.rs:16-21
    TOP = 0
    BOTTOM = 1
    LEFT = 2
    RIGHT = 3


class RCNLayer:
    """rcn.rs:35-38: Convolve2D(Padding) | Pool2D(Pooling)"""
    CONVOLVE2D, POOL2D = 0, 1

    def __init__(self, kind: int, arg: int):
        self.kind, self.arg = int(kind), int(arg)

    @staticmethod
    def Convolve2D(p: Padding) -> "RCNLayer":
        return RCNLayer(RCNLayer.CONVOLVE2D, int(p))

    @staticmethod
    def Pool2D(p: Pooling) -> "RCNLayer":
        return RCNLayer(RCNLayer.POOL2D, int(p))

    def __repr__(self):
        return f"Convolve2D({Padding(self.arg).name})" if self.kind == 0 else f"Pool2D({Pooling(self.arg).name})"


def default_convpool() -> List[RCNLayer]:
    """The architecture hard-coded in rcn/src/main.rs:53-59."""
    return [RCNLayer.Convolve2D(Padding.SAME), RCNLayer.Pool2D(Pooling.MAX),
            RCNLayer.Convolve2D(Padding.SAME), RCNLayer.Pool2D(Pooling.MAX)]


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _cm(m) -> np.ndarray:
    """(R,C) -> flat column-major f64 (nalgebra storage)."""
    return np.asarray(m, dtype=np.float64).ravel(order="F").copy()


class RCN:
    """Mirror of `pub struct RCN` (rcn.rs:13-25) backed by an rcn_hip context.

    `RCN.new(classes, convpool_cfg, feedforward_cfg, training_path, testing_path)` matches rcn.rs:58-75; the
    extra keyword arguments are what a device context needs and the Rust struct does not carry
    (input image shape, device ordinal, arithmetic type)."""

    def __init__(self, classes: int, convpool_cfg: Sequence[RCNLayer], feedforward_cfg: Sequence[int],
                 training_path: str = "", testing_path: str = "", *, input_shape: Tuple[int, int] = (28, 28),
                 dtype: int = F32, device: int = 0, stream: Optional[int] = None, experiments: bool = False):
        self._lib = _lib.load_experiments() if experiments else _lib.load()
        self._ctx = C.c_void_p()
        self.classes = int(classes)
        self.convpool_cfg = list(convpool_cfg)
        self.feedforward_cfg = [int(h) for h in feedforward_cfg]
        self.training_path, self.testing_path = training_path, testing_path
        self.dtype = dtype
        self.np_dtype = np.float64 if dtype == F64 else np.float32
        layers = (_lib.Layer * max(1, len(self.convpool_cfg)))()
        for i, l in enumerate(self.convpool_cfg):
            layers[i].kind, layers[i].arg = l.kind, l.arg
        hidden = (C.c_int32 * max(1, len(self.feedforward_cfg)))(*self.feedforward_cfg)
        cfg = _lib.Cfg(C.sizeof(_lib.Cfg), device, dtype, int(input_shape[0]), int(input_shape[1]), len(self.convpool_cfg), layers,
                       len(self.feedforward_cfg), hidden, self.classes, C.c_void_p(stream) if stream else None)
        st = self._lib.rcn_hip_create(C.byref(cfg), C.byref(self._ctx))
        if st != 0:
            ctx, self._ctx = self._ctx, C.c_void_p()
            try:
                _lib.check(self._lib, ctx if ctx.value else None, st)
            finally:
                if ctx.value:
                    self._lib.rcn_hip_destroy(ctx)
        self.input_shape = (int(input_shape[0]), int(input_shape[1]))
        n = C.c_int64()
        self._ck(self._lib.rcn_hip_feature_len(self._ctx, C.byref(n)))
        self.feature_len = int(n.value)
        self.dims = [self.feature_len] + self.feedforward_cfg + [self.classes]
        self._weights_loaded = False        # layer_weights.is_empty() (rcn.rs:139)

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    # ------------------------------------------------------------------ plumbing
    def _ck(self, st: int):
        _lib.check(self._lib, self._ctx, st)

    @property
    def ctx(self) -> C.c_void_p:
        return self._ctx

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            _lib.FALLBACKS_SEEN += int(self._lib.rcn_hip_fallbacks_taken(self._ctx))     # (the tests assert that none goes unnoticed)
            self._lib.rcn_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._ck(self._lib.rcn_hip_synchronize(self._ctx))

    def set_dense_path(self, mode: int):
        """0 auto, 1 sample-tile kernels, 2 feature-sliced pipeline, 3 resident epoch kernel, 4 one launch per step (include/rcn_hip.h)."""
        self._ck(self._lib.rcn_hip_set_dense_path(self._ctx, int(mode)))

    def set_option(self, name: str, value: int):
        """Per-context option (include/rcn_hip.h: rcn_hip_set_option); the environment only seeds the defaults at creation."""
        self._ck(self._lib.rcn_hip_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        self._ck(self._lib.rcn_hip_get_option(self._ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def fallbacks_taken(self) -> int:
        """How often this context stepped down from the resident one-XCD kernel to the two-kernel pipeline by itself."""
        return int(self._lib.rcn_hip_fallbacks_taken(self._ctx))

    TIMEOUT_FIELDS = ("site", "worker", "step", "launch", "missing_lo", "missing_hi", "tag", "xcc", "rank", "world", "xsel", "workers", "code")

    def last_timeout(self):
        """The record of the newest expired wait of the resident kernel in this context (rcn_hip_last_timeout), or None; `text` adds the
        workspace's placement / flag tables as the failed launch left them."""
        import ctypes as C
        w = (C.c_uint32 * 16)()
        n = int(self._lib.rcn_hip_last_timeout(self._ctx, w, 16))
        if n == 0:
            return None
        rec = {k: int(w[i]) for i, k in enumerate(self.TIMEOUT_FIELDS[:n])}
        rec["step"] = rec["step"] - (1 << 32) if rec["step"] >= (1 << 31) else rec["step"]
        lo, hi = rec.pop("missing_lo"), rec.pop("missing_hi")
        if 5 <= rec["site"] <= 8:                                   # pushed exchange: rank mask, and the waiting lane's first parameter index
            rec["missing"], rec["param_index"] = lo, hi
        else:
            rec["missing"] = lo | (hi << 32)
        rec["text"] = (self._lib.rcn_hip_last_timeout_text(self._ctx) or b"").decode()
        return rec

    def set_feature_kernel(self, mode: int):
        """0 auto (specialised fused kernel for the default stack on 28x28), 1 always the generic kernel."""
        self._ck(self._lib.rcn_hip_set_feature_kernel(self._ctx, int(mode)))

    # ------------------------------------------------------------------ parameters (Weights / Bias)
    def set_params(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray]):
        """weights[l]: (out, in) array == Weights.0; biases[l]: (out,) == Bias.0  (rcn.rs:28,31)"""
        if len(weights) != len(self.dims) - 1 or len(biases) != len(weights):
            raise ValueError("need one (W, b) pair per dense layer")
        for l, (w, b) in enumerate(zip(weights, biases)):
            w = np.asarray(w, dtype=np.float64)
            if w.shape != (self.dims[l + 1], self.dims[l]) or np.asarray(b).shape != (self.dims[l + 1],):
                raise ValueError(f"layer {l}: expected W {(self.dims[l + 1], self.dims[l])}, b {(self.dims[l + 1],)}")
            self._ck(self._lib.rcn_hip_set_params(self._ctx, l, _dp(_cm(w)), _dp(_f64c(b))))
        self._weights_loaded = True

    def get_params(self) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        ws, bs = [], []
        for l in range(len(self.dims) - 1):
            w = np.zeros(self.dims[l] * self.dims[l + 1])
            b = np.zeros(self.dims[l + 1])
            self._ck(self._lib.rcn_hip_get_params(self._ctx, l, _dp(w), _dp(b)))
            ws.append(w.reshape((self.dims[l + 1], self.dims[l]), order="F").copy())
            bs.append(b)
        return ws, bs

    def load_weights_and_bias(self, seed: int = 0):
        """rcn.rs:425-457: N(0,1) weights and biases (seed 0 = nondeterministic like thread_rng)."""
        self._ck(self._lib.rcn_hip_init_params(self._ctx, seed))
        self._weights_loaded = True

    @property
    def scale_set(self) -> Tuple[float, float]:
        m, s = C.c_double(), C.c_double()
        self._ck(self._lib.rcn_hip_get_scale(self._ctx, C.byref(m), C.byref(s)))
        return m.value, s.value

    @scale_set.setter
    def scale_set(self, ms: Tuple[float, float]):
        self._ck(self._lib.rcn_hip_set_scale(self._ctx, float(ms[0]), float(ms[1])))

    # ------------------------------------------------------------------ private seams of the reference
    def flatten_feature_set(self, imgs_u8: np.ndarray) -> np.ndarray:
        """rcn.rs:317-356 (+ get_pixel_matrix, lib.rs:27-41) for one [H,W] or many [N,H,W] u8 images -> [N,F] f64."""
        imgs = np.ascontiguousarray(imgs_u8, dtype=np.uint8)
        single = imgs.ndim == 2
        if single:
            imgs = imgs[None]
        if imgs.shape[1:] != self.input_shape:
            raise ValueError(f"images must be {self.input_shape}, got {imgs.shape[1:]}")
        out = np.zeros((imgs.shape[0], self.feature_len))
        self._ck(self._lib.rcn_hip_features(self._ctx, imgs.ctypes.data_as(C.POINTER(C.c_uint8)), imgs.shape[0], _dp(out)))
        return out[0] if single else out

    def gen_scales(self, feats: np.ndarray) -> Tuple[float, float]:
        """rcn.rs:230-251; also overwrites scale_set like the reference."""
        f = _f64c(feats)
        m, s = C.c_double(), C.c_double()
        self._ck(self._lib.rcn_hip_gen_scales(self._ctx, _dp(f), f.shape[0], C.byref(m), C.byref(s)))
        return m.value, s.value

    def standardize(self, feats: np.ndarray) -> np.ndarray:
        """rcn.rs:407-412: max((x - mean)/sd, 0) with the current scale_set."""
        f = _f64c(feats).copy()
        self._ck(self._lib.rcn_hip_standardize(self._ctx, _dp(f), f.size))
        return f

    def train_batch(self, x: np.ndarray, y: np.ndarray, eta: float, want_loss: bool = False) -> Optional[float]:
        """rcn.rs:176-223.  x: [B,F], y: [B,classes] (the InputSet halves, rcn.rs:49)."""
        x, y = _f64c(x), _f64c(y)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0] or x.shape[1] != self.dims[0] or y.shape[1] != self.classes:
            raise ValueError("train_batch: x must be [B,F] and y [B,classes]")
        loss = C.c_double()
        self._ck(self._lib.rcn_hip_train_batch(self._ctx, _dp(x), _dp(y), x.shape[0], float(eta), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def classify_test(self, x: np.ndarray) -> np.ndarray:
        """rcn.rs:105-116 for one [F] or many [N,F] inputs."""
        x = _f64c(x)
        single = x.ndim == 1
        if single:
            x = x[None]
        out = np.zeros((x.shape[0], self.classes))
        self._ck(self._lib.rcn_hip_forward(self._ctx, _dp(x), x.shape[0], _dp(out)))
        return out[0] if single else out

    def evaluate(self, x: np.ndarray, y: np.ndarray) -> int:
        """The per-epoch test pass of rcn.rs:152-157: number of accepted samples."""
        x, y = _f64c(x), _f64c(y)
        n = C.c_int64()
        self._ck(self._lib.rcn_hip_evaluate(self._ctx, _dp(x), _dp(y), x.shape[0], C.byref(n)))
        return int(n.value)

    # ------------------------------------------------------------------ public API of the reference
    def classify(self, img_u8: np.ndarray) -> int:
        """rcn.rs:82-98 minus the PNG decode: a grayscale [H,W] u8 image -> class index."""
        img = np.ascontiguousarray(img_u8, dtype=np.uint8)
        cls = np.zeros(1, dtype=np.int32)
        self._ck(self._lib.rcn_hip_classify_images(self._ctx, img.ctypes.data_as(C.POINTER(C.c_uint8)), 1, cls.ctypes.data_as(C.POINTER(C.c_int32))))
        return int(cls[0])

    def classify_many(self, imgs_u8: np.ndarray) -> np.ndarray:
        imgs = np.ascontiguousarray(imgs_u8, dtype=np.uint8)
        cls = np.zeros(imgs.shape[0], dtype=np.int32)
        self._ck(self._lib.rcn_hip_classify_images(self._ctx, imgs.ctypes.data_as(C.POINTER(C.c_uint8)), imgs.shape[0], cls.ctypes.data_as(C.POINTER(C.c_int32))))
        return cls

    def load_data(self, imgs_u8: np.ndarray, labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """The arithmetic of rcn.rs:399-414 on already-decoded images: features, gen_scales (overwrites
        scale_set), standardise+clamp, one-hot expectations (rcn.rs:466-471)."""
        feats = self.flatten_feature_set(imgs_u8)
        self.gen_scales(feats)
        x = self.standardize(feats)
        y = np.zeros((len(labels), self.classes))
        y[np.arange(len(labels)), np.asarray(labels)] = 1.0
        return x, y

    def train(self, batch_size: int, epochs: int, eta: float, training_class_size_limit: int, testing_class_size_limit: int,
              rng: Optional[np.random.Generator] = None, log=print) -> List[int]:
        """RCN::train (rcn.rs:126-133): reads `training_path` / `testing_path` (one directory per class, PNG files),
        everything after the decode runs on the GPU (mercer_research_amd.data.train_from_directories)."""
        from .data import train_from_directories
        return train_from_directories(self, batch_size, epochs, eta, training_class_size_limit, testing_class_size_limit, rng, log)

    def classify_file(self, img_path: str) -> int:
        """RCN::classify (rcn.rs:82-98) including the PNG decode + grayscale."""
        from . import png
        with open(img_path, "rb") as f:
            return self.classify(png.to_pixel_matrix_u8(f.read()))

    def train_arrays(self, train_imgs, train_labels, test_imgs, test_labels, batch_size: int, epochs: int, eta: float,
                     rng: Optional[np.random.Generator] = None, log=print) -> List[int]:
        """RCN::train (rcn.rs:126-167) on decoded images: load both sets (test statistics end up in scale_set,
        SURVEY Q7), init weights if empty, then per epoch shuffle / chunks_exact / train_batch / evaluate."""
        rng = rng or np.random.default_rng()
        n, nt = self.load_set(0, train_imgs, train_labels), self.load_set(1, test_imgs, test_labels)      # rcn.rs:134-137, resident in HBM
        if not self._weights_loaded:                                   # rcn.rs:139-141
            self.load_weights_and_bias()
        accepted = []
        for e in range(epochs):                                        # rcn.rs:144
            order = rng.permutation(n).astype(np.int32)                # rcn.rs:146
            self.train_set_epoch(0, batch_size, eta, perm=order)       # chunks_exact + train_batch, rcn.rs:147-149: one call per epoch
            acc = self.evaluate_set(1)                                 # rcn.rs:152-157
            accepted.append(acc)
            if log:
                log("Epoch {}: {}/{} [{:.2f}%]".format(e, acc, nt, acc / nt * 100.0))   # rcn.rs:158-164
        return accepted

    # ------------------------------------------------------------------ RCN::train's data flow with the sets resident in HBM
    def load_set(self, slot: int, imgs_u8: np.ndarray, labels: np.ndarray) -> int:
        """load_data after the decode (rcn.rs:399-414) into device slot 0 (training) / 1 (testing); overwrites scale_set."""
        imgs = np.ascontiguousarray(imgs_u8, dtype=np.uint8)
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        if imgs.ndim != 3 or tuple(imgs.shape[1:]) != self.input_shape or len(lab) != len(imgs):
            raise ValueError(f"expected [n]{list(self.input_shape)} u8 images and n labels")
        self._ck(self._lib.rcn_hip_load_data(self._ctx, slot, imgs.ctypes.data_as(C.POINTER(C.c_uint8)), lab.ctypes.data_as(C.POINTER(C.c_int32)), len(imgs), None, None))
        return len(imgs)

    def train_set_epoch(self, slot: int, batch_size: int, eta: float, perm: Optional[np.ndarray] = None, seed: int = 0, want_loss: bool = False):
        """One pass of rcn.rs:146-149 over a loaded slot; perm = the shuffled order (or None: shuffled on the device with `seed`)."""
        n = C.c_int64()
        self._ck(self._lib.rcn_hip_set_size(self._ctx, slot, C.byref(n)))
        nb = n.value // batch_size if batch_size else 0
        p = np.ascontiguousarray(perm, dtype=np.int32) if perm is not None else None
        if p is not None and p.size < nb * batch_size:
            raise ValueError("perm shorter than the batches it has to cover")
        loss = np.zeros(max(nb, 1)) if want_loss else None
        self._ck(self._lib.rcn_hip_train_set_epoch(self._ctx, slot, p.ctypes.data_as(C.POINTER(C.c_int32)) if p is not None else None, seed & 0xFFFFFFFFFFFFFFFF,
                                                   batch_size, float(eta), loss.ctypes.data_as(C.POINTER(C.c_double)) if want_loss else None))
        return loss[:nb] if want_loss else None

    def evaluate_set(self, slot: int) -> int:
        acc = C.c_int64()
        self._ck(self._lib.rcn_hip_evaluate_set(self._ctx, slot, C.byref(acc)))
        return int(acc.value)


# ---------------------------------------------------------------------------------------------------------------
# Operator traits (utils/kernel.rs:61-100, 219-236) as free functions over NumPy matrices.
# ---------------------------------------------------------------------------------------------------------------
_op_ctx: Optional[RCN] = None


def _ops() -> RCN:
    """A small shared context for the stateless operator calls."""
    global _op_ctx
    if _op_ctx is None:
        _op_ctx = RCN(2, [RCNLayer.Convolve2D(Padding.SAME), RCNLayer.Pool2D(Pooling.MAX)], [2], input_shape=(4, 4), dtype=F64)
    return _op_ctx


def _stack(m) -> Tuple[np.ndarray, bool]:
    a = np.asarray(m, dtype=np.float64)
    if a.ndim == 2:
        return a[None], True
    if a.ndim != 3:
        raise ValueError("expected a matrix [R,C] or a batch [N,R,C]")
    return a, False


def convolve_2d(m, kernel, padding: Padding) -> np.ndarray:
    """Convolve2D::convolve_2d (kernel.rs:110-194).  Raises RcnPanic where the reference panics."""
    r = _ops()
    a, single = _stack(m)
    k = np.asarray(kernel, dtype=np.float64)
    if k.ndim != 2:
        raise ValueError("kernel must be a matrix")
    n, R, Cc = a.shape
    oR, oC = C.c_int(), C.c_int()
    _lib.check(r._lib, None, r._lib.rcn_hip_conv_out_shape(R, Cc, k.shape[0], k.shape[1], int(padding), C.byref(oR), C.byref(oC)))
    flat = np.concatenate([_cm(x) for x in a]) if n else np.zeros(0)
    out = np.zeros(n * oR.value * oC.value)
    r._ck(r._lib.rcn_hip_convolve_2d(r._ctx, _dp(flat), n, R, Cc, _dp(_cm(k)), k.shape[0], k.shape[1], int(padding), _dp(out)))
    res = np.stack([out[i * oR.value * oC.value:(i + 1) * oR.value * oC.value].reshape((oR.value, oC.value), order="F") for i in range(n)])
    return res[0] if single else res


def convolve_2d_separated(m, op: SeparableOperator, padding: Padding) -> np.ndarray:
    """Convolve2D::convolve_2d_separated (kernel.rs:196-207)."""
    r = _ops()
    a, single = _stack(m)
    n, R, Cc = a.shape
    flat = np.concatenate([_cm(x) for x in a]) if n else np.zeros(0)
    oR, oC = (R, Cc) if int(padding) == Padding.SAME else (R - 2, Cc - 2)
    out = np.zeros(max(1, n * max(oR, 0) * max(oC, 0)))
    r._ck(r._lib.rcn_hip_convolve_2d_separated(r._ctx, _dp(flat), n, R, Cc, int(op), int(padding), _dp(out)))
    res = np.stack([out[i * oR * oC:(i + 1) * oR * oC].reshape((oR, oC), order="F") for i in range(n)])
    return res[0] if single else res


def relu(m) -> np.ndarray:
    """Convolve2D::relu (kernel.rs:209-216)."""
    r = _ops()
    a = _f64c(m)
    out = np.zeros_like(a)
    r._ck(r._lib.rcn_hip_relu(r._ctx, _dp(a.reshape(-1)), a.size, _dp(out.reshape(-1))))
    return out


def pool_2d(m, padding: Padding, pooling: Pooling) -> np.ndarray:
    """Pool2D::pool_2d (kernel.rs:245-349)."""
    r = _ops()
    a, single = _stack(m)
    n, R, Cc = a.shape
    oR, oC = C.c_int(), C.c_int()
    _lib.check(r._lib, None, r._lib.rcn_hip_pool_out_shape(R, Cc, int(padding), C.byref(oR), C.byref(oC)))
    flat = np.concatenate([_cm(x) for x in a]) if n else np.zeros(0)
    out = np.zeros(max(1, n * oR.value * oC.value))
    r._ck(r._lib.rcn_hip_pool_2d(r._ctx, _dp(flat), n, R, Cc, int(padding), int(pooling), _dp(out)))
    sz = oR.value * oC.value
    res = np.stack([out[i * sz:(i + 1) * sz].reshape((oR.value, oC.value), order="F") for i in range(n)])
    return res[0] if single else res
