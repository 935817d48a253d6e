#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage), filtered by a substring.

    python tools/kernel_resources.py mercer_research_amd/csrc/rcn_hipx_api.hip halo_f32
"""
import re, subprocess, sys
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src] + sys.argv[3:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:Function )?Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void rcnx::", "")
    if pat in dem:
        print(f"{dem:60s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} spill {r.get('VGPR Spill', -1):3d} scratch {r.get('ScratchSize', -1):4d} LDS {r.get('LDS Size', -1):6d} occ {r.get('Occupancy', -1)}")
