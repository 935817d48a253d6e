"""Synthetic workload of SURVEY.md §8(d) -- there is no dataset in the build environment: MNIST-like u8 images and
N(0,1) parameters in the reference's shapes and draw order.  Pure NumPy, part of the product (bench.py, tools); the oracle
keeps its own copy for the tests, and tests/test_formats.py checks the two generate identical arrays."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def synthetic_images(n: int, h: int = 28, w: int = 28, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """uint8 [n][h][w] images with a 4-pixel zero border and ~19 % non-zero pixels overall (MNIST's statistics, which also
    decide how often the Padding::Same edge quirk is visible), and labels uniform in 0..9."""
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
    """Weights (out x in) and biases drawn N(0,1) in the order RCN::load_weights_and_bias draws them (rcn.rs:500-523:
    column-major fill of each DMatrix, then the bias vector)."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(len(dims) - 1):
        ws.append(rng.standard_normal(dims[i] * dims[i + 1]).reshape((dims[i + 1], dims[i]), order="F"))
        bs.append(rng.standard_normal(dims[i + 1]))
    return ws, bs
