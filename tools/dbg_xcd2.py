import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
B, N = 256, 2304
rng = np.random.default_rng(2)
X = np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32)
Y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)]
ws, bs = synthetic_params([784, 30, 10], seed=8)
ws = [w * 0.1 for w in ws]
perm = np.random.default_rng(5).permutation(N).astype(np.int32)
def run(path, nb, use_perm):
    d = DeviceRCN(dtype=0)
    d.set_dense_path(path)
    d.set_params(ws, bs)
    Xd, Yd = d.to_device(X), d.to_device(Y)
    pd = d.to_device(perm) if use_perm else None
    loss = d.empty(nb)
    d.train_epoch(Xd, Yd, pd, B, nb, 3.0, loss)
    d.synchronize()
    p = np.concatenate([a.ravel() for a in sum(d.get_params(), [])])
    out = (p, loss.cpu().numpy().copy())
    d.rcn.close()
    return out
for use_perm in (False, True):
    for nb in (3, 6, 8):
        (pa, la), (pb, lb) = run(5, nb, use_perm), run(2, nb, use_perm)
        print("perm" if use_perm else "id", nb, "max|dp|", float(np.abs(pa - pb).max()), "\n  loss5", np.round(la, 5), "\n  loss2", np.round(lb, 5), flush=True)
