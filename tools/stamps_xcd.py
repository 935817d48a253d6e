#!/usr/bin/env python3
"""Diagnostic only: -DRCN_STAMPS build, a few 64-step launches of the resident one-XCD kernel (dense path 5), then where each worker
was at each point of the launch's last-but-one step (100 MHz ticks -> us relative to the earliest stamp of that step)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercer_research_amd import build as hb, _lib
out = os.path.join(ROOT, "gpurun_out", "librcn_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([hb.hipcc()] + hb.FLAGS + ["-w", "-DRCN_STAMPS", "-o", out, os.path.join(hb.CSRC, "rcn_hip_api.hip")], check=True)
_lib.LIB_PATH = out
import torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
F64 = os.environ.get("STAMPS_DTYPE", "f32") == "f64"          # the reference's own type: the kernel's f64 instantiations (batches up to 128)
d = DeviceRCN(dtype=1 if F64 else 0)
lib = d.lib
lib.rcn_hip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
ws, bs = synthetic_params([784, 30, 10], seed=42)
d.set_params(ws, bs)
N, B = 16384, int(os.environ.get("STAMPS_B", "256"))
with torch.cuda.stream(d.stream):
    X = torch.rand(N, 784, device=d.device, dtype=d.tdtype)
    Y = torch.zeros(N, 10, device=d.device, dtype=d.tdtype); Y[:, 3] = 1
    perm = torch.randperm(N, device=d.device).int()
d.set_dense_path(5)
d.synchronize()
DP = os.environ.get("STAMPS_DP", "0") == "1"            # the data-parallel instantiation at a group of one (bootstrap over RCCL forced on)
if DP:
    d.dp_init()
    assert d.dp_resident(B), "the data-parallel step does not run on the resident kernel here"
for it in range(3):
    (d.dp_train_epoch if DP else d.train_epoch)(X, Y, perm, B, 64, 3.0, None)
d.synchronize()
print(("data-parallel instantiation (world 1)" if DP else "single-GPU instantiation") + f", {'f64' if F64 else 'f32'}, B = {B}")
st = np.zeros((2, 512, 16), dtype=np.uint64)
lib.rcn_hip_debug_read_stamps(d.ctx, st.ctypes.data_as(C.c_void_p))
r = st[0].astype(np.int64)[:32, :16]
t0 = r[r > 0].min()
rel = np.where(r > 0, (r - t0) / 100.0, np.nan)
names = ["0 step top", "1 slab+tail flags seen", "2 slab part sums in LDS (wave 7)", "3 past barrier (wave 0)", "4 tail done", "5 flagB stored", "6 all flagB seen (wave 1)",
         "7 past barrier", "8 gradient MFMA done", "9 partials summed barrier", "10 slice updated barrier", "11 forward stored+drained", "12 flagA / flagT stored", "13 forward done wave 0", "14 forward done wave 4", "15 forward done wave 7"]
np.set_printoptions(linewidth=200, precision=2, suppress=True)
for i, nme in enumerate(names):
    col = rel[:, i]
    print(f"{nme:38s} feature workers 0-24: mean {np.nanmean(col[:25]):5.2f} min {np.nanmin(col[:25]):5.2f} max {np.nanmax(col[:25]):5.2f} | tail 25-27: {np.round(col[25:28], 2)} | sample-only 28-31: {np.round(col[28:32], 2)}")

c = st[1].astype(np.int64)[:32, :4]
ghz = (c[:, 2] - c[:, 0]) / ((c[:, 3] - c[:, 1]) * 10.0)      # shader cycles per ns
print("in-kernel shader clock over steps 8..56 (GHz), per worker:", np.round(ghz, 3))
print("us per step over that stretch:", np.round((c[:, 3] - c[:, 1]) / 100.0 / 48, 3)[:4])
