#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE counter_collection.csv:
median over dispatches; fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128), the normalisation under which a kernel that keeps every
SIMD's matrix pipe busy for its whole duration reads 1.0 on gfx950 (same as profiles/r1_trackx_mfma_pmc.json).  usage: mfma_pmc_summary.py <dir> <out.json>"""
import csv, glob, json, statistics, sys
import os
rows = list(csv.DictReader(open(max(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime))))   # newest run
acc = {}
for r in rows:
    if 'rcnx::' not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    acc.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
out = {}
for k, d in sorted(acc.items()):
    e = {c: statistics.median(v) for c, v in d.items()}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e.get('GRBM_GUI_ACTIVE'):
        e['mfma_busy_fraction_of_simd_cycles'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] * 128), 4)
    e['dispatches'] = len(next(iter(d.values())))
    out[k] = e
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, e in out.items():
    print(f"{k[:70]:70s} {e.get('mfma_busy_fraction_of_simd_cycles')}")
