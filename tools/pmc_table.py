#!/usr/bin/env python3
"""Median per-dispatch value of every counter in the rocprofv3 --pmc passes under <dir> (one sub-directory per pass), per kernel whose name
contains <filter>.  usage: pmc_table.py <dir> <filter> [out.json]"""
import csv, glob, json, statistics, sys
acc = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
out = {k: {c: statistics.median(v) for c, v in sorted(d.items())} for k, d in sorted(acc.items())}
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, e in out.items():
    print(k)
    for c, v in e.items():
        print(f"   {c:36s} {v:16.0f}")
