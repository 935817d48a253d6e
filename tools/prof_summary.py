#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel stats + steady-state dense-kernel durations."""
import csv, glob, statistics, sys
d = sys.argv[1]
st = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    print(r['Name'][:64].ljust(64), 'calls', r['Calls'].rjust(6), 'avg_ns', r['AverageNs'].rjust(12), 'min', r['MinNs'].rjust(8), 'max', r['MaxNs'].rjust(8), 'pct', r['Percentage'])
tr = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in tr if 'rcn::k_' in r['Kernel_Name'])
tail = ks[len(ks) // 2:]
by = {}
for a, b, n in tail:
    by.setdefault(n.split('(')[0][-48:], []).append(b - a)
for n, v in by.items():
    print('steady', n.ljust(50), 'n', len(v), 'median_ns', statistics.median(v), 'p10', sorted(v)[len(v)//10], 'p90', sorted(v)[len(v)*9//10])
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
print('median gap ns', statistics.median(gaps))
