"""Throughput of flatten_feature_set on the device (HIP events on the context's stream): the fused kernel for the
default stack against the generic layer-walking kernel, with the HBM roofline of the algorithmic bytes
(784 B in + 784*4 B out per image).  Usage: python tools/feat_bench.py [n_images]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercer_research_amd.device import DeviceRCN  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    d = DeviceRCN(dtype=0)
    rng = np.random.default_rng(0)
    imgs = d.to_device(rng.integers(0, 256, (n, 28, 28)).astype(np.uint8))
    out = d.empty(n, d.F)
    res = {}
    for name, mode in (("k_features_cpcp", 0), ("k_features", 1)):
        d.set_feature_kernel(mode)
        for std in (0, 1):
            for _ in range(3):
                d.features(imgs, bool(std), out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(d.stream)
            for _ in range(reps):
                d.features(imgs, bool(std), out)
            e1.record(d.stream)
            d.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            byts = n * (784 + 784 * 4)
            res[f"{name}{'+standardise' if std else ''}"] = {"us": round(us, 1), "Mimg_per_s": round(n / us, 1), "GB_per_s": round(byts / us / 1e3, 1),
                                                             "frac_hbm_8TBs": round(byts / us / 1e3 / 8000, 4)}
    print(json.dumps({"n_images": n, "results": res}, indent=1))


if __name__ == "__main__":
    main()
