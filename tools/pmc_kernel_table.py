"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_kernel_table.py <dir> [name filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
for k, c in rows[:int(os.environ.get("TOP", "14"))]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    short = re.sub(r"^_ZN4rcnx\d+", "", k)[:70]
    print(f"{short:70s} n={len(calls[k]):3d} wave_cyc={wc:.3g} " + " ".join(f"{n[3:]}={v / wc:.3f}" for n, v in sorted(c.items()) if n != "SQ_WAVE_CYCLES"))
