import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
B, N = 256, 4096
rng = np.random.default_rng(2)
X = np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32)
Y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)]
ws, bs = synthetic_params([784, 30, 10], seed=8)
ws = [w * 0.1 for w in ws]
def run(path, calls):
    d = DeviceRCN(dtype=0)
    d.set_dense_path(path)
    d.set_params(ws, bs)
    Xd, Yd = d.to_device(X), d.to_device(Y)
    out = []
    for (j0, nb) in calls:
        loss = d.empty(nb)
        d.train_epoch(Xd[j0 * B:], Yd[j0 * B:], None, B, nb, 3.0, loss)
        d.synchronize()
        p = np.concatenate([a.ravel() for a in sum(d.get_params(), [])])
        out.append((p, loss.cpu().numpy().copy()))
    d.rcn.close()
    return out
for calls in ([(0, 1)], [(0, 2)], [(0, 3)], [(0, 3), (3, 3)], [(0, 1), (1, 1), (2, 1)], [(0, 8)], [(0, 16)]):
    a, b = run(5, calls), run(2, calls)
    for i, ((pa, la), (pb, lb)) in enumerate(zip(a, b)):
        print(os.environ.get("RCN_HIP_PACK_SEGMENT_BYTES", "-"), calls, "call", i, "max|dp|", float(np.abs(pa - pb).max()), "loss5", np.round(la, 5), "loss2", np.round(lb, 5), flush=True)
