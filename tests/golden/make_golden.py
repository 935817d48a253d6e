#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/rcn_oracle.c via ctypes).

Provenance: the Rust reference cannot be built or run in this environment (no cargo/rustc), so these
vectors are NOT outputs of the reference.  They are outputs of the loop-faithful C restatement, which is
itself pinned only by the reference's two data-free KATs (utils/kernel.rs:402-417, :436-441) and
cross-checked by the independent NumPy restatement.  They pin the oracle against regressions and let the
GPU box (where /root/reference does not exist) check the HIP path against fixed numbers.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.rcn_oracle import (COracle, DEFAULT_LAYERS, LAYER_CONV, LAYER_POOL, PAD_NONE, PAD_SAME, POOL_MAX,  # noqa: E402
                               build_oracle, one_hot, synthetic_images, synthetic_params)

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    build_oracle()
    o = COracle()
    rng = np.random.default_rng(20240607)

    # ---- operators (kernel.rs) -------------------------------------------------------------------
    ops = {}
    for name, shape in (("a5x6", (5, 6)), ("b28x28", (28, 28)), ("c7x3", (7, 3)), ("d9x9", (9, 9))):
        m = rng.integers(0, 256, shape).astype(np.float64)
        ops[f"{name}_m"] = m
        for pad in (PAD_NONE, PAD_SAME):
            for op in range(4):
                ops[f"{name}_sep_p{pad}_op{op}"] = o.convolve_2d_separated(m, op, pad)
            ops[f"{name}_pool_p{pad}"] = o.pool_2d(m, pad, POOL_MAX)
        k33 = rng.standard_normal((3, 3))
        ops[f"{name}_k33"] = k33
        for pad in (PAD_NONE, PAD_SAME):
            ops[f"{name}_conv33_p{pad}"] = o.convolve_2d(m, k33, pad)
    np.savez_compressed(os.path.join(OUT, "operators.npz"), **ops)

    # ---- feature pipeline (rcn.rs:317-356) -------------------------------------------------------
    imgs, labels = synthetic_images(6, seed=7)
    feats = o.features(imgs, DEFAULT_LAYERS)
    mean, sd = o.gen_scales(feats)
    variants = {
        "conv_none_pool": ((LAYER_CONV, PAD_NONE), (LAYER_POOL, POOL_MAX)),
        "conv_conv_pool": ((LAYER_CONV, PAD_SAME), (LAYER_CONV, PAD_NONE), (LAYER_POOL, POOL_MAX)),
        "pool_first": ((LAYER_POOL, POOL_MAX), (LAYER_CONV, PAD_SAME), (LAYER_POOL, POOL_MAX)),
    }
    small = rng.integers(0, 256, (3, 9, 11)).astype(np.uint8)
    fx = dict(imgs=imgs, labels=labels, feats=feats, mean=mean, sd=sd, std=o.standardize(feats, mean, sd), small=small)
    for k, lay in variants.items():
        fx[f"small_{k}"] = o.features(small, lay)
        fx[f"layers_{k}"] = np.array(lay, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "features.npz"), **fx)

    # ---- dense step (rcn.rs:176-314) -------------------------------------------------------------
    dense = {}
    for name, dims, B, eta in (("tiny", [16, 5, 3], 4, 3.0), ("mid3", [48, 8, 6, 10], 7, 0.5), ("one", [24, 4, 10], 1, 3.0)):
        ws, bs = synthetic_params(dims, seed=len(name) + B)
        ws = [w * 0.3 for w in ws]
        X = np.maximum(rng.standard_normal((B, dims[0])), 0.0)
        Y = one_hot(rng.integers(0, dims[-1], B), dims[-1])
        gW, gb, cost = o.batch_gradient(ws, bs, X, Y)
        nw, nb, cost2 = o.train_batch(ws, bs, X, Y, eta)
        assert cost == cost2
        dense[f"{name}_dims"] = np.array(dims)
        dense[f"{name}_eta"] = eta
        dense[f"{name}_X"], dense[f"{name}_Y"] = X, Y
        dense[f"{name}_cost"] = cost
        dense[f"{name}_out"] = o.classify_test(ws, bs, X)
        for l in range(len(ws)):
            dense[f"{name}_W{l}"], dense[f"{name}_b{l}"] = ws[l], bs[l]
            dense[f"{name}_gW{l}"], dense[f"{name}_gb{l}"] = gW[l], gb[l]
            dense[f"{name}_nW{l}"], dense[f"{name}_nb{l}"] = nw[l], nb[l]
    # the default 784-30-10 net on synthetic features: inputs are reproducible from seeds, so only the
    # outputs (and a strided sample of the big matrix) are stored
    imgs, labels = synthetic_images(32, seed=11)
    f = o.features(imgs, DEFAULT_LAYERS)
    m, s = o.gen_scales(f)
    X, Y = o.standardize(f, m, s), one_hot(labels)
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    nw, nb, cost = o.train_batch(ws, bs, X, Y, 3.0)
    dense["mnist_cost"] = cost
    dense["mnist_out"] = o.classify_test(ws, bs, X)
    dense["mnist_nb0"], dense["mnist_nb1"], dense["mnist_nW1"] = nb[0], nb[1], nw[1]
    dense["mnist_nW0_strided"] = nw[0].ravel(order="F")[::37].copy()
    np.savez_compressed(os.path.join(OUT, "dense.npz"), **dense)
    for f_ in ("operators.npz", "features.npz", "dense.npz"):
        print(f_, os.path.getsize(os.path.join(OUT, f_)), "bytes")


if __name__ == "__main__":
    main()
