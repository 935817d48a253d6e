#!/usr/bin/env python3
"""The reference's own criterion benches (rcn/benches/convolve.rs:19-51, rcn/benches/train.rs:8-24), re-run on this
library: the five single-image operator calls through the C ABI (f64 host matrix in, f64 host matrix out -- what a
`DMatrix` caller of the `Convolve2D` / `Pool2D` traits sees) next to the same operator on the oracle's CPU restatement, and
"Train w/ 10 in Batch, 500 per class" on synthetic MNIST-shape data (the dataset is not in the build environment).  NOT
the BASELINE metric (that is bench.py).  A single 28x28 operator call is one PCIe round trip around a microsecond of work,
so the GPU loses these by construction; the point of the file is that the reference's benches have their counterpart and
an honest number, and that the batched feature path is where the same arithmetic belongs (last line)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    import mercer_research_amd as amd
    from mercer_research_amd.synth import synthetic_images
    from oracle.rcn_oracle import COracle          # the CPU side of the comparison (reported baseline)
    o = COracle()
    imgs, labels = synthetic_images(5000, seed=11)
    m = imgs[0].astype(np.float64)
    top = o.sobel_full(0)
    P, Pool, Sep = amd.Padding, amd.Pooling, amd.SeparableOperator
    out = {"unit": "microseconds per call, one 28x28 image, host matrix in / host matrix out"}
    benches = [
        ("Simple Convolution", lambda: amd.convolve_2d(m, top, P.NONE), lambda: o.convolve_2d(m, top, 0)),
        ("Separated Convolution", lambda: amd.convolve_2d_separated(m, Sep.TOP, P.NONE), lambda: o.convolve_2d_separated(m, 0, 0)),
        ("Simple Convolution (Padded: Same)", lambda: amd.convolve_2d(m, top, P.SAME), lambda: o.convolve_2d(m, top, 1)),
        ("Separated Convolution (Padded: Same)", lambda: amd.convolve_2d_separated(m, Sep.TOP, P.SAME), lambda: o.convolve_2d_separated(m, 0, 1)),
        ("Max Pooling", lambda: amd.pool_2d(m, P.SAME, Pool.MAX), lambda: o.pool_2d(m, 1, 1)),
    ]
    for name, g, c in benches:
        assert np.array_equal(g(), c()), name                       # same operator, bit-identical result
        out[name] = {"hip_c_abi_us": round(timeit(g, 300), 1), "cpu_restatement_us": round(timeit(c, 300), 1)}
    # rcn/benches/train.rs: RCN::train(batch 10, ... 500 per class) -- here: load (features + scales) and 10 epochs of B = 10
    r = amd.RCN(10, amd.default_convpool(), [30], dtype=amd.F32)
    r.load_weights_and_bias(seed=1)
    t0 = time.perf_counter()
    x, y = r.load_data(imgs, labels)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    n_steps = 0
    for _ in range(10):
        for j in range(0, len(x) - 9, 10):
            r.train_batch(x[j:j + 10], y[j:j + 10], 3.0)
            n_steps += 1
    t_train = time.perf_counter() - t0
    out["Train w/ 10 Epochs, 10 in Batch, 5000 images (host buffers through rcn_hip_train_batch)"] = {
        "load_features_and_scales_s": round(t_load, 3), "train_s": round(t_train, 3), "us_per_train_batch": round(t_train / n_steps * 1e6, 1)}
    # where the same arithmetic belongs: the batched feature kernel
    t = timeit(lambda: r.flatten_feature_set(imgs), 5)
    out["flatten_feature_set, 5000 images per call (host u8 in, host f64 out)"] = {"us_per_image": round(t / len(imgs), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
