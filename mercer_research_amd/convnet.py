"""ctypes face of librcn_hipx.so (include/rcn_hipx.h): the Track-X trainable convolution network.  No reference
counterpart (see the header); torch tensors are used only as HBM buffers."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
LIBX_PATH = os.path.join(HERE, "librcn_hipx.so")
KIND = {"conv": 0, "pool": 1, "dense_relu": 2, "dense": 3}


class XLayer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("out", C.c_int32)]


_vp, _i = C.c_void_p, C.c_int
SIGNATURES = {
    "rcn_hipx_create": (_i, [_i, _i, _i, _i, C.POINTER(XLayer), _i, _i, _vp, C.POINTER(_vp)]),
    "rcn_hipx_destroy": (None, [_vp]),
    "rcn_hipx_last_error": (C.c_char_p, [_vp]),
    "rcn_hipx_synchronize": (_i, [_vp]),
    "rcn_hipx_param_count": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rcn_hipx_classes": (_i, [_vp]),
    "rcn_hipx_set_params": (_i, [_vp, C.POINTER(C.c_float)]),
    "rcn_hipx_get_params": (_i, [_vp, C.POINTER(C.c_float)]),
    "rcn_hipx_init_params": (_i, [_vp, C.c_uint64]),
    "rcn_hipx_forward_dev": (_i, [_vp, _vp, _i, _vp]),
    "rcn_hipx_train_step_dev": (_i, [_vp, _vp, _vp, _i, C.c_float, _vp]),
    "rcn_hipx_gradients_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "rcn_hipx_apply_dev": (_i, [_vp, _vp, C.c_float]),
    "rcn_hipx_unpad_host": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "rcn_hipx_set_precision": (_i, [_vp, _i]),
    "rcn_hipx_set_tiling": (_i, [_vp, _i]),
    "rcn_hipx_set_overlap": (_i, [_vp, _i]),
    "rcn_hipx_gradients_begin_dev": (_i, [_vp, _vp, _vp, _i, _vp, _vp, C.c_int64, C.POINTER(C.c_int)]),
    "rcn_hipx_gradients_bucket_dev": (_i, [_vp, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rcn_hipx_plan_buckets": (_i, [_i, _i, _i, C.POINTER(XLayer), _i, _i, _i, _i, C.c_int64, C.c_char_p, _i]),
    "rcn_hipx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "rcn_hipx_get_option": (_i, [_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "rcn_hipx_plan_net": (_i, [_vp, _i, C.c_char_p, _i]),
    "rcn_hipx_step_flops": (_i, [_vp, _i, C.POINTER(C.c_double)]),
    "rcn_hipx_plan": (_i, [_i, _i, _i, C.POINTER(XLayer), _i, _i, _i, _i, C.c_char_p, _i]),
}
_libx = None


def load():
    global _libx
    if _libx is None:
        if not os.path.exists(LIBX_PATH):
            raise ImportError(f"{LIBX_PATH} not found: build it with `python -m mercer_research_amd.build`; there is no CPU fallback")
        _lib.preload_hip_runtime()
        lib = C.CDLL(os.environ.get("RCN_HIPX_TEST_LIB") or LIBX_PATH)          # (RCN_HIPX_TEST_LIB: diagnostic builds, tools/ablate_halo_bf16.sh)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _libx = lib
    return _libx


class ConvNetError(RuntimeError):
    pass


DEFAULT_BUCKET_BYTES = 1 << 20      # gradient buckets of the data-parallel step: at least 1 MiB each (a ring all-reduce below that is latency)


PRECISIONS = {"fp32": 0, "bf16": 1, "bf16_stored": 2}


def plan(in_shape: Tuple[int, int, int], layers: Sequence[tuple], batch: int, precision: str = "fp32", tiling: str = "auto", buckets: Optional[int] = None) -> str:
    """Which kernels a training step of this net would launch, one line per launch (rcn_hipx_plan): the library's own dispatch code run
    with its launches replaced by notes.  Needs no GPU.  buckets = N: the bucketed GRADIENT step of a data-parallel rank instead
    (rcn_hipx_plan_buckets, buckets of at least N bytes): per bucket its launches and the slice of the flat gradient that becomes final."""
    lib = load()
    arr = (XLayer * len(layers))()
    for i, l in enumerate(layers):
        arr[i].kind, arr[i].out = KIND[l[0]], int(l[1]) if len(l) > 1 else 0
    buf = C.create_string_buffer(1 << 16)
    if buckets is not None:
        st = lib.rcn_hipx_plan_buckets(in_shape[0], in_shape[1], in_shape[2], arr, len(layers), batch, PRECISIONS[precision], {"gemm": 0, "auto": 1, "lds": 2}[tiling],
                                       int(buckets), buf, len(buf))
        if st != 0:
            raise ConvNetError(f"rcn_hipx_plan_buckets: {st}: {buf.value.decode()}")
        return buf.value.decode()
    st = lib.rcn_hipx_plan(in_shape[0], in_shape[1], in_shape[2], arr, len(layers), batch, PRECISIONS[precision], {"gemm": 0, "auto": 1, "lds": 2}[tiling], buf, len(buf))
    if st != 0:
        raise ConvNetError(f"rcn_hipx_plan: {st}: {buf.value.decode()}")
    return buf.value.decode()


class ConvNet:
    """layers: sequence of ("conv", Cout) | ("pool",) | ("dense_relu", units) | ("dense", classes)."""

    def __init__(self, in_shape: Tuple[int, int, int], layers: Sequence[tuple], max_batch: int, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("ConvNet needs a GPU; there is no CPU fallback")
        self.lib = load()
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.in_shape, self.layers, self.max_batch = tuple(in_shape), [tuple(l) for l in layers], max_batch
        arr = (XLayer * len(layers))()
        for i, l in enumerate(layers):
            arr[i].kind, arr[i].out = KIND[l[0]], int(l[1]) if len(l) > 1 else 0
        self.net = C.c_void_p()
        st = self.lib.rcn_hipx_create(device, in_shape[0], in_shape[1], in_shape[2], arr, len(layers), max_batch, C.c_void_p(self.stream.cuda_stream), C.byref(self.net))
        if st != 0:
            msg = self.lib.rcn_hipx_last_error(self.net).decode() if self.net.value else f"status {st}"
            if self.net.value:
                self.lib.rcn_hipx_destroy(self.net)
            self.net = C.c_void_p()
            raise ConvNetError(f"rcn_hipx_create: {st}: {msg}")
        a, b = C.c_int64(), C.c_int64()
        self.lib.rcn_hipx_param_count(self.net, C.byref(a), C.byref(b))
        self.n_logical, self.n_padded = int(a.value), int(b.value)
        self.classes = self.lib.rcn_hipx_classes(self.net)

    def _ck(self, st):
        if st != 0:
            raise ConvNetError(f"rcn_hipx status {st}: {self.lib.rcn_hipx_last_error(self.net).decode()}")

    def close(self):
        if getattr(self, "net", None) is not None and self.net.value:
            self.lib.rcn_hipx_destroy(self.net)
            self.net = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._ck(self.lib.rcn_hipx_synchronize(self.net))

    def set_precision(self, mode: str):
        """"fp32" (fp32 MFMA, default), "bf16" (bf16 MFMA operands, fp32 accumulate / storage / update) or "bf16_stored" (bf16 operands
        AND the convolutional stage's activations / gradients kept in HBM as bf16: include/rcn_hipx.h, RCN_HIPX_BF16_STORED)."""
        self._ck(self.lib.rcn_hipx_set_precision(self.net, PRECISIONS[mode]))

    def set_tiling(self, mode: str):
        """fp32 3x3 kernels: "gemm" (implicit GEMM only), "auto" (by shape, the default) or "lds" (LDS-tiled wherever they apply)."""
        self._ck(self.lib.rcn_hipx_set_tiling(self.net, {"gemm": 0, "auto": 1, "lds": 2}[mode]))

    def set_overlap(self, mode):
        """Backward pass: weight gradients on a second stream beside the input-gradient chain: 0 / False = no (default: measured no
        gain), 1 / True = every layer's, 2 = the dense layers' only."""
        self._ck(self.lib.rcn_hipx_set_overlap(self.net, int(mode)))

    def set_option(self, name: str, value: int):
        """A kernel-selection knob of THIS net (rcn_hipx_set_option; the environment only seeds the defaults at creation)."""
        self._ck(self.lib.rcn_hipx_set_option(self.net, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        if self.lib.rcn_hipx_get_option(self.net, name.encode(), C.byref(v)) != 0:
            raise ConvNetError(f"unknown option {name!r}")
        return int(v.value)

    def plan_of_this_net(self, batch: int) -> str:
        """The launches a training step of THIS net would make, with its own precision, tiling and options (rcn_hipx_plan_net)."""
        buf = C.create_string_buffer(1 << 16)
        st = self.lib.rcn_hipx_plan_net(self.net, int(batch), buf, len(buf))
        if st != 0:
            raise ConvNetError(f"rcn_hipx_plan_net: {st}: {buf.value.decode()}")
        return buf.value.decode()

    def set_params(self, flat: np.ndarray):
        f = np.ascontiguousarray(flat, dtype=np.float32)
        assert f.size == self.n_logical
        self._ck(self.lib.rcn_hipx_set_params(self.net, f.ctypes.data_as(C.POINTER(C.c_float))))

    def get_params(self) -> np.ndarray:
        f = np.zeros(self.n_logical, dtype=np.float32)
        self._ck(self.lib.rcn_hipx_get_params(self.net, f.ctypes.data_as(C.POINTER(C.c_float))))
        return f

    def init_params(self, seed: int = 1):
        self._ck(self.lib.rcn_hipx_init_params(self.net, seed))

    def to_device(self, a: np.ndarray):
        with self.torch.cuda.stream(self.stream):
            return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def forward(self, x):
        out = self.torch.empty(x.shape[0], self.classes, dtype=self.torch.float32, device=self.device)
        self._ck(self.lib.rcn_hipx_forward_dev(self.net, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr())))
        return out

    def train_step(self, x, labels, lr: float, loss=None):
        self._ck(self.lib.rcn_hipx_train_step_dev(self.net, C.c_void_p(x.data_ptr()), C.c_void_p(labels.data_ptr()), x.shape[0], lr,
                                                  C.c_void_p(loss.data_ptr()) if loss is not None else None))

    def gradients(self, x, labels, grad=None, loss=None):
        grad = grad if grad is not None else self.torch.empty(self.n_padded, dtype=self.torch.float32, device=self.device)
        self._ck(self.lib.rcn_hipx_gradients_dev(self.net, C.c_void_p(x.data_ptr()), C.c_void_p(labels.data_ptr()), x.shape[0], C.c_void_p(grad.data_ptr()),
                                                 C.c_void_p(loss.data_ptr()) if loss is not None else None))
        return grad

    def gradients_bucketed(self, x, labels, grad, loss=None, min_bucket_bytes: int = DEFAULT_BUCKET_BYTES, on_bucket=None):
        """The gradients of `gradients`, bucket by bucket (rcn_hipx_gradients_begin_dev / _bucket_dev): after every bucket's launches are
        enqueued, on_bucket(slice_of_grad, k, n) is called -- a data-parallel step starts that slice's all-reduce there, on another stream,
        behind an event of this net's stream.  Bit-identical to `gradients`."""
        nb = C.c_int()
        self._ck(self.lib.rcn_hipx_gradients_begin_dev(self.net, C.c_void_p(x.data_ptr()), C.c_void_p(labels.data_ptr()), x.shape[0], C.c_void_p(grad.data_ptr()),
                                                       C.c_void_p(loss.data_ptr()) if loss is not None else None, int(min_bucket_bytes), C.byref(nb)))
        off, ln = C.c_int64(), C.c_int64()
        for k in range(nb.value):
            self._ck(self.lib.rcn_hipx_gradients_bucket_dev(self.net, k, C.byref(off), C.byref(ln)))
            if on_bucket is not None:
                on_bucket(grad[off.value:off.value + ln.value], k, nb.value)
        return grad

    def apply(self, grad, scale: float):
        self._ck(self.lib.rcn_hipx_apply_dev(self.net, C.c_void_p(grad.data_ptr()), scale))

    def unpad(self, padded) -> np.ndarray:
        f = np.zeros(self.n_logical, dtype=np.float32)
        self._ck(self.lib.rcn_hipx_unpad_host(self.net, C.c_void_p(padded.data_ptr()), f.ctypes.data_as(C.POINTER(C.c_float))))
        return f

    def step_hbm_floor_bytes(self, B: int, stored16: bool = False) -> float:
        """HBM floor of one training step with activations stored as they are (fp32; stored16: the convolutional stage's maps and their
        gradients as bf16, "bf16_stored"): every layer's input read and output written once in the forward
        pass; in the backward pass dZ read and dX written once by the input-gradient GEMM (not for the first layer) and the input and dZ
        read once more by the weight-gradient GEMM; parameters read twice and written once.  Fusion (pool in the epilogue, ReLU masks in
        the consumer) can go below it only by not materialising a tensor at all."""
        H, W, Cc = self.in_shape
        total, first = 0.0, True
        es_in = 4                                            # bytes per element of the current layer's input tensor
        es_stage = 2 if stored16 else 4
        for l in self.layers:
            if l[0] == "conv":
                i, o = H * W * Cc * es_in, H * W * l[1] * es_stage
                total += B * ((i + o) + (0 if first else (i + o)) + (i + o)) + 3 * (9 * Cc * l[1] + l[1]) * 4
                Cc, first, es_in = l[1], False, es_stage
            elif l[0] == "pool":
                i, o = H * W * Cc * es_stage, (H // 2) * (W // 2) * Cc * es_stage
                total += B * 2 * (i + o)
                H, W = H // 2, W // 2
            else:
                i, o = H * W * Cc * es_in, l[1] * 4
                total += B * ((i + o) + (0 if first else (i + o)) + (i + o)) + 3 * (H * W * Cc * l[1] + l[1]) * 4
                H, W, Cc, first, es_in = 1, 1, l[1], False, 4
        return total

    def step_flops(self, B: int) -> float:
        f = C.c_double()
        self._ck(self.lib.rcn_hipx_step_flops(self.net, B, C.byref(f)))
        return f.value
