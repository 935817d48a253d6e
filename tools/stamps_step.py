#!/usr/bin/env python3
"""Diagnostic only: -DRCN_STAMPS build, a few epochs of the one-launch step (dense path 4), then where each kind of workgroup
was at each point of the graph's 32nd step (100 MHz ticks -> us relative to the earliest stamp of that launch)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercer_research_amd import build as hb, _lib
out = os.path.join(ROOT, "gpurun_out", "librcn_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([hb.hipcc()] + hb.FLAGS + ["-w", "-DRCN_STAMPS", "-DRCN_HIP_EXPERIMENTS", "-o", out, os.path.join(hb.CSRC, "rcn_hip_api.hip")], check=True)
_lib.LIB_EXP_PATH = out
import torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
d = DeviceRCN(experiments=True)          # librcn_hip_exp.so
lib = d.lib
lib.rcn_hip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
ws, bs = synthetic_params([784, 30, 10], seed=42)
d.set_params(ws, bs)
N, B = 16384, 256
with torch.cuda.stream(d.stream):
    X = torch.rand(N, 784, device=d.device)
    Y = torch.zeros(N, 10, device=d.device); Y[:, 3] = 1
    perm = torch.randperm(N, device=d.device).int()
d.set_dense_path(4)
d.synchronize()
for it in range(3):
    d.train_epoch(X, Y, perm, B, 64, 3.0, None)
d.synchronize()
st = np.zeros((2, 512, 16), dtype=np.uint64)
lib.rcn_hip_debug_read_stamps(d.ctx, st.ctypes.data_as(C.c_void_p))
G, NS, NT = 49, 32, 3
t0 = st[0][:NS + G + NT, 0].astype(np.int64).min()
def show(name, rows, npts, k=0):
    r = st[k].astype(np.int64)[rows, :npts]
    rel = np.where(r > 0, (r - t0) / 100.0, np.nan)
    print(f"{name}\n   mean us:", np.round(np.nanmean(rel, axis=0), 2), " min:", np.round(np.nanmin(rel, axis=0), 2), " max:", np.round(np.nanmax(rel, axis=0), 2))
show("sample  (0 start, 1 slab summed [thread 0], 2 barrier, 3 outputs stored, 4 drained + flag)", slice(0, NS), 5)
show("feature (0 start, 1 flags seen, 2 d1 loaded + MFMA, 3 barrier, 4 slice updated, 5 next partials stored)", slice(NS, NS + G), 6)
show("tail    (0 start, 1 operands in + barrier)", slice(NS + G, NS + G + NT), 2)
print("--- the next launch (step 33), same clock origin")
show("sample ", slice(0, NS), 5, 1)
show("feature", slice(NS, NS + G), 6, 1)
