#!/usr/bin/env python3
"""Diagnostic only: builds a -DRCN_STAMPS copy of the library, runs a few pipelined steps and prints where each
workgroup of k_pipe_a / k_pipe_b spends its time (100 MHz s_memrealtime ticks -> microseconds)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercer_research_amd import build as hb, _lib
out = os.path.join(ROOT, "gpurun_out", "librcn_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([hb.hipcc()] + hb.FLAGS + ["-DRCN_STAMPS", "-o", out, os.path.join(hb.CSRC, "rcn_hip_api.hip")], check=True)
_lib.LIB_PATH = out
import torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
hidden = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [30]
d = DeviceRCN(feedforward_cfg=hidden)
lib = d.lib
lib.rcn_hip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
ws, bs = synthetic_params([784] + hidden + [10], seed=42)
d.set_params(ws, bs)
N, B = 16384, 256
with torch.cuda.stream(d.stream):
    X = torch.rand(N, 784, device=d.device)
    Y = torch.zeros(N, 10, device=d.device); Y[:, 3] = 1
    perm = torch.randperm(N, device=d.device).int()
path = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d.set_dense_path(path)
d.synchronize()
for it in range(3):
    d.train_epoch(X, Y, perm, B, 64, 3.0, None)
d.synchronize()
st = np.zeros((2, 512, 16), dtype=np.uint64)
lib.rcn_hip_debug_read_stamps(d.ctx, st.ctypes.data_as(C.c_void_p))
for kid, name, nwg, npts in ((0, "k_pipe_a (stamps 0-3: last launch, U only; 4-5 from the launch before)", 49, 6), (1, "k_pipe_b", 32, 7)):
    s = st[kid, :nwg, :npts].astype(np.int64)
    t0 = s[:, 0].min()
    print(name, "start spread us", (s[:, 0].max() - t0) / 100.0)
    rel = (s - t0) / 100.0
    rel = np.where(s > 0, rel, np.nan)
    print("  mean us since first WG start at each stamp:", np.round(np.nanmean(rel, axis=0), 2))
    print("  max  us:", np.round(np.nanmax(rel, axis=0), 2))
tail = st[0, 49:52, :7].astype(np.int64)
t0a = st[0, :52, 0].astype(np.int64).min()
print("k_p2_a tail tiles (db_0, [W_1|b_1] x2): start, end us since the first workgroup's start:", np.round((tail[:, 0] - t0a) / 100.0, 2), np.round((tail[:, 6] - t0a) / 100.0, 2))
