"""Per-step kernel times from a rocprofv3 --stats kernel_stats.csv: python tools/kernel_stats_table.py <dir> <launches of the step that were traced>"""
import csv, glob, os, re, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    rows.append((float(r["TotalDurationNs"]) / 1e3 / steps, int(r["Calls"]) / steps, re.sub(r"^_ZN4rcnx\d+|^void rcnx::|^rcnx::", "", r["Name"])))
rows.sort(reverse=True)
print(f"{sum(r[0] for r in rows):9.1f} us per step in kernels")
for t, c, n in rows[:int(os.environ.get("TOP", "20"))]:
    print(f"{t:9.1f} us  x{c:<4.3g} {n[:120]}")
