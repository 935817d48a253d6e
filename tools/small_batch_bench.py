#!/usr/bin/env python3
"""(--dtype f64: the reference's own arithmetic type.)  Device-timed step time of train_batch at the reference's own batch sizes (10: rcn/src/main.rs:36-37; 32: BASELINE configs[0]) and
the sizes between, on the default path (the resident one-XCD kernel's instantiation for the next of 32/64/128/256) and on the
two-kernel pipeline / sample-tile kernels it replaces there.  One JSON line.

    python tools/small_batch_bench.py [--batches 10,32,64,128,256] [--dims 784,30,10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(torch, amd, DeviceRCN, imgs_d, labels_d, dims, B, path, n_images, epochs=8, dtype=0):
    from mercer_research_amd.synth import synthetic_params
    d = DeviceRCN(classes=dims[-1], feedforward_cfg=dims[1:-1], input_shape=(28, 28), dtype=dtype)
    ws, bs = synthetic_params(dims, seed=42)
    d.set_params(ws, bs)
    if path:
        d.set_dense_path(path)
    X, Y = d.load_data(imgs_d, labels_d)
    nb = min(n_images // B, 256)                       # chunks_exact(B): the tail is dropped (rcn.rs:147)
    perm = torch.empty(n_images, dtype=torch.int32, device=d.device)

    def run(n, seed0):
        for e in range(n):
            d.shuffle(perm, n_images, 1, seed=seed0 + e)
            d.train_epoch(X, Y, perm, B, nb, 3.0, None)
    run(2, 11)
    d.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(d.stream)
    run(epochs, 100)
    b.record(d.stream)
    d.synchronize()
    us = a.elapsed_time(b) * 1e3 / (epochs * nb)
    k1, k2, _ = d.time_kernels(X[:B], Y[:B], reps=128)
    out = {"us_per_step": round(us, 3), "images_per_s": round(B / us * 1e6, 1), "steps_per_s": round(1e6 / us, 1), "resident_kernel": k1 == 0.0,
           "kernel_us_per_step": round(k2, 3) if k1 == 0.0 else None, "steps": epochs * nb, "fallbacks": d.fallbacks_taken()}
    d.rcn.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="10,32,64,128,256")
    ap.add_argument("--dims", default="784,30,10")
    ap.add_argument("--images", type=int, default=16384)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    args = ap.parse_args()
    import torch
    import mercer_research_amd as amd
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.synth import synthetic_images
    dims = [int(v) for v in args.dims.split(",")]
    imgs, labels = synthetic_images(args.images, seed=1234)
    dev = torch.device("cuda", 0)
    imgs_d, labels_d = torch.from_numpy(imgs).to(dev), torch.from_numpy(labels).to(dev)
    dt = amd.F64 if args.dtype == "f64" else amd.F32
    out = {"dims": dims, "dtype": args.dtype}
    for B in [int(v) for v in args.batches.split(",")]:
        out[f"B{B}"] = {"default_path": measure(torch, amd, DeviceRCN, imgs_d, labels_d, dims, B, 0, args.images, dtype=dt),
                        "two_kernel_or_sample_tile": measure(torch, amd, DeviceRCN, imgs_d, labels_d, dims, B, 2 if B % 256 == 0 else 1, args.images, dtype=dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
