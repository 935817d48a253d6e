#!/usr/bin/env python3
"""Diagnostic: microseconds per train_batch step as a function of how many steps one hipGraph replay covers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mercer_research_amd.device import DeviceRCN
from mercer_research_amd.synth import synthetic_params
d = DeviceRCN()
ws, bs = synthetic_params([784, 30, 10], seed=42)
d.set_params(ws, bs)
N, B = 16384, 256
with torch.cuda.stream(d.stream):
    X = torch.rand(N, 784, device=d.device)
    Y = torch.zeros(N, 10, device=d.device); Y[:, 3] = 1
    perm = torch.cat([torch.randperm(N, device=d.device) for _ in range(8)]).int()
d.synchronize()
for nb in (16, 64, 128, 256, 512):
    d.prepare_epoch(X, Y, perm, B, nb, 3.0, None)
    reps = max(2, 2048 // nb)
    d.train_epoch(X, Y, perm, B, nb, 3.0, None); d.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d.train_epoch(X, Y, perm, B, nb, 3.0, None)
    d.synchronize()
    el = time.perf_counter() - t0
    print(f"nb={nb:4d} reps={reps:4d}: {el / (reps * nb) * 1e6:7.2f} us/step   ({el / reps * 1e6:9.1f} us per replay)", flush=True)

# ---- what does the shuffle plumbing of bench.py cost? (nb = 64 per replay)
nb = 64
side = torch.cuda.Stream(device=d.device)
ev_ready, ev_used = torch.cuda.Event(), torch.cuda.Event()
perm2 = torch.empty(N, dtype=torch.int32, device=d.device)
def variant(name, record=False, wait=False, shuffle=False, reps=32):
    ev_used.record(d.stream); ev_ready.record(side)
    d.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if wait: d.stream.wait_event(ev_ready)
        if shuffle:
            with torch.cuda.stream(side):
                side.wait_event(ev_used)
                perm2.copy_(torch.randperm(N, device=d.device))
                ev_ready.record(side)
        d.train_epoch(X, Y, perm, B, nb, 3.0, None)
        if record: ev_used.record(d.stream)
    d.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name:46s}: {el / (reps * nb) * 1e6:7.2f} us/step", flush=True)
variant("plain replays")
variant("+ event record after each replay", record=True)
variant("+ wait on an already-complete event", record=True, wait=True)
variant("+ randperm on a side stream per replay", record=True, wait=True, shuffle=True)
variant("randperm on side stream, no events on main", shuffle=True)
