"""f64 NumPy oracle for the Track-X convolution network (mercer_research_amd/csrc/convnet.hpp) -- TEST INFRASTRUCTURE ONLY.

There is NO reference counterpart: the reference's conv layers are fixed Sobel filters without a backward pass
(rcn/src/utils/kernel.rs:38-53, rcn/src/rcn.rs:317-356), and it has no softmax / cross-entropy.  This oracle is therefore
pinned only by its own finite-difference gradient check (tests/test_convnet_oracle.py): "parity unpinned".

Layers: ("conv", Cout) 3x3 stride 1 pad 1 + bias + ReLU | ("pool",) 2x2/2 max | ("dense_relu", units) | ("dense", classes).
Activations NHWC; conv weights W[K][Cout] with K = (kh*3 + kw)*Cin + ci; dense weights W[features][units] with features
flattened in (h, w, c) order; loss = mean cross-entropy of softmax(logits)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def param_shapes(in_shape: Tuple[int, int, int], layers: Sequence[tuple]) -> List[Tuple[Tuple[int, int], int]]:
    H, W, C = in_shape
    out = []
    for l in layers:
        if l[0] == "conv":
            out.append(((9 * C, l[1]), l[1])); C = l[1]
        elif l[0] == "pool":
            H, W = H // 2, W // 2
        else:
            feat = H * W * C
            out.append(((feat, l[1]), l[1])); H, W, C = 1, 1, l[1]
    return out


def unflatten(flat: np.ndarray, in_shape, layers):
    ws, bs, o = [], [], 0
    for (k, c), nb in param_shapes(in_shape, layers):
        ws.append(np.asarray(flat[o:o + k * c], dtype=np.float64).reshape(k, c)); o += k * c
        bs.append(np.asarray(flat[o:o + nb], dtype=np.float64)); o += nb
    assert o == len(flat)
    return ws, bs


def flatten(ws, bs) -> np.ndarray:
    return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(ws, bs)])


def round_bf16(a: np.ndarray) -> np.ndarray:
    """Round to the nearest bfloat16 (ties to even), returned as f64 -- what v_cvt_pk_bf16_f32 does to a GEMM operand in
    the bf16 MFMA path (csrc/convnet_bf16.hpp).  Input goes through fp32 first, as on the device."""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


def _op(a: np.ndarray, operand: str) -> np.ndarray:
    return round_bf16(a) if operand == "bf16" else a


def _im2col(x: np.ndarray) -> np.ndarray:
    """[N,H,W,C] -> [N*H*W, 9*C] with k = (kh*3+kw)*C + ci (zero padding 1)."""
    N, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    cols = [xp[:, kh:kh + H, kw:kw + W, :] for kh in range(3) for kw in range(3)]
    return np.concatenate(cols, axis=-1).reshape(N * H * W, 9 * C)


def head_is_fp32(layers) -> bool:
    """The bf16 mode of the device library (include/rcn_hipx.h) rounds the operands of every GEMM except two places that run fp32
    kernels in either mode: a first layer whose whole 3x3xCin patch is one k-block (9 * Cin <= 32), and the classifier head when it is
    the fused one -- a logits layer of at most 32 classes on a ReLU dense layer of at most 256 units (csrc/rcn_hipx_api.hip head_fusable)."""
    return len(layers) >= 2 and layers[-1][0] == "dense" and layers[-1][1] <= 32 and layers[-2][0] == "dense_relu" and layers[-2][1] <= 256


def forward(x, ws, bs, layers, cache=None, operand="f64", stored=False):
    """operand="bf16": both operands of every forward GEMM are rounded to bf16 first (products and sums stay f64) -- except in the
    first layer when its patch is one k-block, and in the fused classifier head (head_is_fp32).
    stored=True mirrors RCN_HIPX_BF16_STORED (include/rcn_hipx.h): the maps the convolutional stage keeps in memory -- a convolution's
    output, or the pooled map when a pool follows it (the device then never writes the full-resolution one; its arg-max is taken on the
    unrounded sums) -- are rounded to bf16 where they are written."""
    a = np.asarray(x, dtype=np.float64)
    pi = 0
    for li, l in enumerate(layers):
        if l[0] == "conv":
            N, H, W, C = a.shape
            cols = _im2col(a)
            op = "f64" if cols.shape[1] <= 32 else operand
            z = _op(cols, op) @ _op(ws[pi], op) + bs[pi]
            y = np.maximum(z, 0).reshape(N, H, W, -1)
            if stored and not (li + 1 < len(layers) and layers[li + 1][0] == "pool"):
                y = round_bf16(y)
            if cache is not None:
                cache.append(("conv", cols, y, a.shape))
            a = y; pi += 1
        elif l[0] == "pool":
            N, H, W, C = a.shape
            win = a.reshape(N, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 4, C)
            idx = win.argmax(axis=3)                      # first maximum, positions ordered (dy, dx) = 00, 01, 10, 11
            p = np.take_along_axis(win, idx[:, :, :, None, :], axis=3)[:, :, :, 0, :]
            if stored:
                p = round_bf16(p)
            if cache is not None:
                cache.append(("pool", idx, a.shape))
            a = p
        else:
            f = a.reshape(a.shape[0], -1)
            op = "f64" if (li == len(layers) - 1 and head_is_fp32(layers)) else operand
            z = _op(f, op) @ _op(ws[pi], op) + bs[pi]
            y = np.maximum(z, 0) if l[0] == "dense_relu" else z
            if cache is not None:
                cache.append((l[0], f, y, a.shape))
            a = y; pi += 1
    return a


def loss_and_grads(x, labels, ws, bs, layers, operand="f64", stored=False):
    """operand="bf16": every GEMM takes bf16-rounded operands -- forward, input gradient and weight gradient -- except a first layer
    whose whole patch fits one k-block (9*Cin <= 32) and the fused classifier head (head_is_fp32), which run fp32 kernels on the
    device in either mode.  Bias gradients are plain sums of the unrounded dZ.
    stored=True (RCN_HIPX_BF16_STORED): besides the maps (forward), the gradient with respect to every map of the convolutional stage is
    rounded to bf16 where the input-gradient kernel of the layer above writes it; the bias gradients and the first layer's weight
    gradient are then sums over that rounded dZ."""
    cache = []
    logits = forward(x, ws, bs, layers, cache, operand, stored)
    B = logits.shape[0]
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z); p /= p.sum(axis=1, keepdims=True)
    loss = float(-np.log(p[np.arange(B), labels]).mean())
    d = p.copy(); d[np.arange(B), labels] -= 1.0; d /= B
    gws, gbs = [None] * len(ws), [None] * len(bs)
    pi = len(ws) - 1
    head32 = head_is_fp32(layers)
    for bi, (l, c) in enumerate(zip(reversed(layers), reversed(cache))):
        if c[0] == "pool":
            _, idx, shp = c
            N, H, W, C = shp
            g = np.zeros((N, H // 2, W // 2, 4, C))
            np.put_along_axis(g, idx[:, :, :, None, :], d[:, :, :, None, :], axis=3)
            d = g.reshape(N, H // 2, W // 2, 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, W, C)
        elif c[0] == "conv":
            _, cols, y, shp = c
            N, H, W, C = shp
            dz = (d * (y > 0)).reshape(N * H * W, -1)
            small = cols.shape[1] <= 32
            gws[pi] = cols.T @ dz if small else _op(cols, operand).T @ _op(dz, operand); gbs[pi] = dz.sum(axis=0)
            dcols = (_op(dz, operand) @ _op(ws[pi], operand).T).reshape(N, H, W, 9, C)
            dxp = np.zeros((N, H + 2, W + 2, C))
            t = 0
            for kh in range(3):
                for kw in range(3):
                    dxp[:, kh:kh + H, kw:kw + W, :] += dcols[:, :, :, t, :]; t += 1
            d = dxp[:, 1:-1, 1:-1, :]
            pi -= 1
        else:
            kind, f, y, shp = c
            dz = d * (y > 0) if kind == "dense_relu" else d
            op = "f64" if (bi == 0 and head32) else operand
            gws[pi] = _op(f, op).T @ _op(dz, op); gbs[pi] = dz.sum(axis=0)
            d = (_op(dz, op) @ _op(ws[pi], op).T).reshape(shp)
            pi -= 1
        li = len(layers) - 1 - bi
        if stored and c[0] != "pool" and li > 0 and layers[li - 1][0] in ("conv", "pool"):
            d = round_bf16(d)                              # written by this layer's input-gradient kernel into a bf16 tensor
    return loss, logits, gws, gbs


def sgd_step(x, labels, ws, bs, layers, lr, operand="f64", stored=False):
    loss, logits, gws, gbs = loss_and_grads(x, labels, ws, bs, layers, operand, stored)
    return [w - lr * g for w, g in zip(ws, gws)], [b - lr * g for b, g in zip(bs, gbs)], loss


def numeric_grad(x, labels, ws, bs, layers, which: int, idx, eps=1e-6, bias=False):
    arr = bs[which] if bias else ws[which]
    old = arr[idx]
    arr[idx] = old + eps; lp = loss_and_grads(x, labels, ws, bs, layers)[0]
    arr[idx] = old - eps; lm = loss_and_grads(x, labels, ws, bs, layers)[0]
    arr[idx] = old
    return (lp - lm) / (2 * eps)
