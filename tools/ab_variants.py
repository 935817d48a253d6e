#!/usr/bin/env python3
"""A/B of kernel variants on one box: builds librcn_hip with extra -D flags into mercer_research_amd/variants/ (here, on the CPU:
`python tools/ab_variants.py build name=-DFLAG1,-DFLAG2 ...`), then on the GPU box times the resident kernel's step with every variant
found there, each in its own process (`python tools/ab_variants.py run [B] [rounds]`).  Diagnostic only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "mercer_research_amd", "variants")


def build(specs):
    from mercer_research_amd import build as hb
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        out = os.path.join(VDIR, f"librcn_hip_{name}.so")
        cmd = [hb.hipcc()] + hb.FLAGS + ["-w"] + [f for f in flags.split(",") if f] + ["-o", out, os.path.join(hb.CSRC, "rcn_hip_api.hip")]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        print(name, "rc", p.wait(), flush=True)


def one(path, B):
    from mercer_research_amd import _lib
    _lib.LIB_PATH = path
    import numpy as np
    import torch
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.synth import synthetic_params
    d = DeviceRCN()
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    d.set_params(ws, bs)
    N = 16384
    with torch.cuda.stream(d.stream):
        X = torch.rand(N, 784, device=d.device)
        Y = torch.zeros(N, 10, device=d.device); Y[:, 3] = 1
        perm = torch.randperm(N, device=d.device).int()
    nb = min(N // B, 64)
    for _ in range(3):
        d.train_epoch(X, Y, perm, B, nb, 3.0, None)
    d.synchronize()
    res = []
    for _ in range(5):
        k1, k2, _ = d.time_kernels(X[:B], Y[:B], reps=1024)
        res.append(k2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(d.stream)
    for _ in range(32):
        d.train_epoch(X, Y, perm, B, nb, 3.0, None)
    b.record(d.stream)
    d.synchronize()
    print(json.dumps({"kernel_us_per_step": [round(v, 4) for v in res], "resident": k1 == 0.0, "loop_us_per_step": round(a.elapsed_time(b) * 1e3 / (32 * nb), 4)}))


def run(B, rounds):
    libs = sorted(f for f in os.listdir(VDIR) if f.endswith(".so"))
    out = {}
    for r in range(rounds):
        for f in libs:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", os.path.join(VDIR, f), str(B)], capture_output=True, text=True, timeout=300)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
            try:
                j = json.loads(line)
                out.setdefault(f, []).append((min(j["kernel_us_per_step"]), j["loop_us_per_step"]))
            except Exception:
                out.setdefault(f, []).append(("error", p.stderr[-300:]))
    for f, v in out.items():
        print(f"{f:40s}", " ".join(f"{a}/{b}" for a, b in v), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "one":
        one(sys.argv[2], int(sys.argv[3]))
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 2)
