#!/usr/bin/env python3
"""Diagnostic: instruction mix of the rcn kernels (hipcc -S).  usage: asm_mix.py [name-substring ...]"""
import re, subprocess, sys, os
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/rcn_asm.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only", "-w",
                "-o", out, os.path.join(ROOT, "mercer_research_amd/csrc/rcn_hip_api.hip")], check=True)
lines = open(out).read().split("\n")
want = sys.argv[1:] or ["k_pipe_aIf", "k_pipe_bIf", "k_dense_fwdIfLb1ELb1ELb1E", "k_dense_wgradIfLb1"]
cur, body = None, {}
for l in lines:
    m = re.match(r"^(_ZN3rcn\w+):", l)
    if m:
        cur = m.group(1); body[cur] = []; continue
    if l.startswith(".Lfunc_end"):
        cur = None
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        body[cur].append(l.strip().split()[0])
for name, ins in body.items():
    if not any(w in name for w in want):
        continue
    c = Counter(ins)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(name[:44].ljust(44), "total", len(ins), "| mfma", g(lambda k: "mfma" in k), "| gload", g(lambda k: k.startswith(("global_load", "flat_load"))),
          "| gstore", g(lambda k: k.startswith(("global_store", "flat_store"))), "| ds", g(lambda k: k.startswith("ds_")), "| waitcnt", c["s_waitcnt"],
          "| cndmask", g(lambda k: k.startswith("v_cndmask")), "| branch", g(lambda k: k.startswith("s_cbranch")), "| valu", g(lambda k: k.startswith("v_") and "mfma" not in k),
          "| salu", g(lambda k: k.startswith("s_")), "| scratch", g(lambda k: "scratch" in k))
