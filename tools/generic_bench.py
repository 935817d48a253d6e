"""Step time of the generic feature-sliced pipeline (dense_pipe.hpp) on the reference's own test net 784-10-10-10 (rcn.rs:558,577),
B = 256, f32: HIP-event kernel times (k_pipe_b, k_pipe_a, the alternating pair) and the epoch-loop rate."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
dims = [784] + [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "10,10").split(",")] + [10]
B, N = 256, 16384
d = DeviceRCN(dtype=0, feedforward_cfg=dims[1:-1])
ws, bs = synthetic_params(dims, seed=42)
d.set_params(ws, bs)
with torch.cuda.stream(d.stream):
    X = torch.rand(N, 784, device=d.device)
    Y = torch.zeros(N, 10, device=d.device); Y[:, 3] = 1
    perm = torch.randperm(N, device=d.device).int()
res = {}
for path in (0, 2, 1):
    d.set_dense_path(path)
    for _ in range(3):
        d.train_epoch(X, Y, perm, B, 64, 3.0, None)
    d.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(d.stream)
    for _ in range(16):
        d.train_epoch(X, Y, perm, B, 64, 3.0, None)
    b.record(d.stream)
    d.synchronize()
    us = a.elapsed_time(b) * 1e3 / (16 * 64)
    k1, k2, kp = d.time_kernels(X[:B], Y[:B], reps=256)
    res[{0: "auto", 2: "pipeline", 1: "sample-tile"}[path]] = {"us_per_step_epoch_loop": round(us, 2), "us_first": round(k1, 2), "us_second": round(k2, 2), "us_pair": round(kp, 2)}
print(json.dumps({"dims": dims, "B": B, **res}))
