"""One launch per step (dense path 4, dense_p2_step.hpp) against the two-kernel pipeline (path 2): same parameters bit for bit
after the same epochs, and the step time of each (HIP events on the context's stream, graph replays).
Usage: python tools/step_bench.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercer_research_amd.device import DeviceRCN  # noqa: E402
from mercer_research_amd.synth import synthetic_params  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N, nb = 16384, 16384 // B
    rng = np.random.default_rng(5)
    X = np.maximum(rng.standard_normal((N, 784)), 0).astype(np.float32)
    Y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)]
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    ws = [w * 0.1 for w in ws]
    out = {}
    for path in (2, 4):
        d = DeviceRCN(dtype=0)
        d.set_dense_path(path)
        d.set_params(ws, bs)
        Xd, Yd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype)
        perm = d.to_device(np.random.default_rng(1).permutation(N).astype(np.int32))
        loss = d.empty(nb)
        for _ in range(3):
            d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss)
        d.synchronize()
        params = [p.copy() for p in sum(d.get_params(), [])]
        l0 = loss.cpu().numpy().copy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record(d.stream)
        for _ in range(reps):
            d.train_epoch(Xd, Yd, perm, B, nb, 3.0, loss)
        e1.record(d.stream)
        d.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nb)
        out[path] = (params, l0, us)
        print(f"path {path}: {us:.3f} us per step, {B / us:.2f} M img/s, last cost {l0[-1]:.6f}", flush=True)
        d.rcn.close()
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[2][0], out[4][0]))
    print("parameters bit-identical:", same, " costs bit-identical:", np.array_equal(out[2][1].view(np.uint32), out[4][1].view(np.uint32)))
    if not same:
        print("max abs diff:", max(float(np.max(np.abs(a - b))) for a, b in zip(out[2][0], out[4][0])))


if __name__ == "__main__":
    main()
