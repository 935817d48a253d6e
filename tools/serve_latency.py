#!/usr/bin/env python3
"""single-image latency of RCN::classify (rcn.rs:82-98) behind rcn_hip_classify_images -- what one
request of the reference's backend (backend/src/main.rs:22-42) costs after the PNG decode.  Host buffer in, class out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mercer_research_amd as amd
from mercer_research_amd.synth import synthetic_images, synthetic_params
imgs, _ = synthetic_images(256, seed=3)
ws, bs = synthetic_params([784, 30, 10], seed=42)
for dtype, name in ((amd.F32, "f32"), (amd.F64, "f64")):
    r = amd.RCN(10, amd.default_convpool(), [30], dtype=dtype)
    r.set_params([w * 0.05 for w in ws], bs)
    r.scale_set = (435.4, 547.6)
    for n in (1, 16, 256):
        for _ in range(20):
            r.classify_many(imgs[:n])
        reps = 300
        t0 = time.perf_counter()
        for i in range(reps):
            r.classify_many(imgs[:n]) if n > 1 else r.classify(imgs[i % 256])
        el = (time.perf_counter() - t0) / reps
        print(f"{name} n={n:4d}: {el * 1e6:8.1f} us per call  ({n / el:10.0f} images/s)", flush=True)

# the same request on the host cores through the oracle's restatement of rcn's CPU path (reported baseline)
from oracle.rcn_oracle import COracle, DEFAULT_LAYERS
o = COracle()
wsd = [w * 0.05 for w in ws]
t0 = time.perf_counter()
for i in range(200):
    f = o.flatten_feature_set(o.get_pixel_matrix(imgs[i % 256]), DEFAULT_LAYERS)
    x = o.standardize(f[None], 435.4, 547.6)
    o.classify_argmax(o.classify_test(wsd, bs, x)[0])
print(f"oracle (C restatement of rcn CPU path, 1 thread, via ctypes) n=1: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us per call", flush=True)
