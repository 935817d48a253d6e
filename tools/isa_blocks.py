#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel of a .hip file (device ISA): python tools/isa_blocks.py file.hip 'mangled-name-regex' [min_instrs]"""
import re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src] + sys.argv[4:], capture_output=True, text=True).stdout
lines = asm.split("\n")
start = next(i for i, l in enumerate(lines) if re.match(pat + r".*:", l))
cur, stats, order = "entry", {}, ["entry"]
for l in lines[start + 1:]:
    l = l.strip()
    if l.startswith("s_endpgm"): break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: cur = m.group(1); order.append(cur); continue
    if not l or l[0] in ";.": continue
    op = l.split()[0]
    k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch_") else "vmem" if op.split("_")[0] in ("global", "buffer", "flat") else "other"
    d = stats.setdefault(cur, {}); d[k] = d.get(k, 0) + 1
tot = {}
for b in order:
    d = stats.get(b, {})
    for k, v in d.items(): tot[k] = tot.get(k, 0) + v
    if sum(d.values()) >= minn: print(b, d)
print("total", tot)
