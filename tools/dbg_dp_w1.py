"""debug: resident kernel DP form at world 1 vs the single-GPU kernel, per-step costs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCN_HIP_PACK_SEGMENT_BYTES"] = str(3 * 49 * 256 * 16 * 4)
os.environ["RCN_HIP_DP_P2P"] = "2"
import mercer_research_amd as amd
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
B, nb, N = int(os.environ.get("DBG_B", "256")), 7, 2048
rng = np.random.default_rng(12)
X = np.maximum(rng.standard_normal((N, 784)), 0.0).astype(np.float32)
Y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)]
ws, bs = synthetic_params([784, 30, 10], seed=3)
ws = [w * 0.1 for w in ws]
perm = np.random.default_rng(6).permutation(N).astype(np.int32)
got = {}
for form in ("dp", "single", "dp", "single"):
    d = DeviceRCN(dtype=0)
    d.set_dense_path(5)
    d.set_params(ws, bs)
    Xd, Yd, permd = d.to_device(X, d.tdtype), d.to_device(Y, d.tdtype), d.to_device(perm)
    loss, loss2 = d.empty(nb), d.empty(2)
    if form == "dp":
        d.dp_init()
        d.dp_train_epoch(Xd, Yd, permd, B, nb, 3.0, loss)
        d.dp_train_epoch(Xd, Yd, None, B, 2, 3.0, loss2)
    else:
        d.train_epoch(Xd, Yd, permd, B, nb, 3.0, loss)
        d.train_epoch(Xd, Yd, None, B, 2, 3.0, loss2)
    d.synchronize()
    p = np.concatenate([a.ravel() for a in sum(d.get_params(), [])])
    print(form, loss.cpu().numpy(), loss2.cpu().numpy(), float(np.abs(p).sum()), "fallbacks", d.fallbacks_taken())
    got.setdefault(form, []).append((loss.cpu().numpy().copy(), p))
    if form == "dp":
        d.dp_finalize()
    d.rcn.close()
print("dp vs dp identical:", np.array_equal(got["dp"][0][1], got["dp"][1][1]), "single vs single:", np.array_equal(got["single"][0][1], got["single"][1][1]))
print("loss diff dp-single:", got["dp"][0][0] - got["single"][0][0])
print("param max diff:", float(np.abs(got["dp"][0][1] - got["single"][0][1]).max()))
