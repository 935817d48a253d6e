#!/usr/bin/env python3
"""Track-X benchmark (NOT the BASELINE metric -- that is bench.py): training images/s of the trainable convolution
network BASELINE.json's north_star asks for, on the configs SURVEY.md §8(d) fixes:

  cifar   CIFAR-10 shape 32x32x3: conv3x3 3->32, pool, 32->64, pool, 64->128, pool -> 2048 -> 256 -> 10, B = 512   (configs[2])
  mnist   MNIST shape 28x28x1 LeNet-style: conv 1->32, pool, conv 32->64, pool -> 3136 -> 128 -> 10, B = 256      (configs[1], trainable form)

fp32 activations / weights, fp32 MFMA (v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s).  Reports achieved TFLOP/s of the whole
training step (algorithmic 2*MACs of forward + dgrad + wgrad) and its fraction of the fp32 MFMA peak.  The reference has no
trainable convolution, so there is no reference number to compare with."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np

CONFIGS = {
    "cifar": ((32, 32, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("conv", 128), ("pool",), ("dense_relu", 256), ("dense", 10)), 512),
    "mnist": ((28, 28, 1), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense_relu", 128), ("dense", 10)), 256),
}
F32_MFMA_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=list(CONFIGS), default="cifar")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    import torch
    from mercer_research_amd.convnet import ConvNet
    in_shape, layers, B = CONFIGS[args.config]
    B = args.batch or B
    net = ConvNet(in_shape, layers, B)
    net.init_params(1)
    rng = np.random.default_rng(0)
    nbuf = 8                                               # rotate over several resident batches
    xs = [net.to_device(rng.standard_normal((B,) + in_shape).astype(np.float32)) for _ in range(nbuf)]
    ys = [net.to_device(rng.integers(0, 10, B).astype(np.int32)) for _ in range(nbuf)]
    loss = torch.zeros(1, dtype=torch.float32, device=net.device)
    net.synchronize()
    for i in range(max(args.warmup, 2 * nbuf)):            # first use of each (x, y) pair instantiates its graph
        net.train_step(xs[i % nbuf], ys[i % nbuf], 0.01, loss)
    net.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        net.train_step(xs[i % nbuf], ys[i % nbuf], 0.01, loss)
    net.synchronize()
    el = time.perf_counter() - t0
    flops = net.step_flops(B)
    tf = flops * args.steps / el / 1e12
    print(json.dumps({"metric": "training images/sec (Track X, trainable conv net; not the BASELINE metric)", "config": args.config, "batch": B,
                      "value": round(B * args.steps / el, 1), "unit": "images/s", "ms_per_step": round(el / args.steps * 1e3, 4),
                      "step_gflop": round(flops / 1e9, 3), "achieved_tflops": round(tf, 2), "mfma_fp32_peak_tflops": F32_MFMA_PEAK_TFLOPS,
                      "frac_of_fp32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "dtype": "f32", "data": "synthetic", "final_loss": round(loss.item(), 4)}))


if __name__ == "__main__":
    main()
