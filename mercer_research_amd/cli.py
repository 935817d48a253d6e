"""Drop-in for the reference CLI (rcn/src/main.rs:8-79): same flags and defaults, the hard-coded architecture
conv(Same), pool(Max), conv(Same), pool(Max) + [30] hidden, loads ./rcn.bin when present, trains, prints the
reference's per-epoch line and writes ./rcn.bin in the reference's bincode format.

    python -m mercer_research_amd.cli --training-path images/mnist_png/training --testing-path images/mnist_png/testing
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="rcn", description="A convolutional neural network built in Rust -- MI355X hot path")
    ap.add_argument("-n", "--num-classes", type=int, default=10, help="Number of classes")
    ap.add_argument("--training-path", default="images/mnist_png/training", help="Training directory")
    ap.add_argument("--testing-path", default="images/mnist_png/testing", help="Testing/validation directory")
    ap.add_argument("--training-class-size", type=int, default=500, help="Number of items to train on per class")
    ap.add_argument("--testing-class-size", type=int, default=500, help="Number of items to test on per class")
    ap.add_argument("-l", "--learning-rate", type=float, default=3.0, help="Learning rate (eta value)")
    ap.add_argument("-b", "--batches", type=int, default=10, help="Number of training cycles before update")
    ap.add_argument("-e", "--epochs", type=int, default=30, help="Number of passes through the entire training set")
    # not in the reference: what a device context needs
    ap.add_argument("--model-path", default="./rcn.bin")
    ap.add_argument("--input-shape", type=int, nargs=2, default=(28, 28), metavar=("H", "W"))
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)

    from . import F32, F64, RCN, checkpoint, default_convpool
    kw = dict(input_shape=tuple(args.input_shape), dtype=F64 if args.dtype == "f64" else F32)
    if os.path.exists(args.model_path):                                   # main.rs:47-50
        model = checkpoint.load_model(args.model_path, **kw)
        model.training_path, model.testing_path = args.training_path, args.testing_path
    else:                                                                  # main.rs:51-62
        model = RCN(args.num_classes, default_convpool(), [30], args.training_path, args.testing_path, **kw)
    # main.rs:65-74 matches on train's Result, but load_data unwrap()s every I/O / decode error (rcn.rs:369-398): a bad file is
    # a panic in the reference, so exceptions simply propagate here
    model.train(args.batches, args.epochs, args.learning_rate, args.training_class_size, args.testing_class_size,
                rng=np.random.default_rng(args.seed))
    checkpoint.save_model(model, args.model_path)                          # main.rs:77
    return 0


if __name__ == "__main__":
    sys.exit(main())
