#!/usr/bin/env python3
"""Reads a CSV of tools/halo_stamps.hip: per workgroup [id, xcc, hw_id, 30 stamps in 10 ns ticks since the first start].
Prints the distribution of workgroups over CUs and, per stage, the median / p90 duration."""
import sys, collections, statistics as st
rows = []
nph = 3
for line in open(sys.argv[1]):
    if line.startswith("#"):
        print(line.strip()); nph = int(line.split("nph")[1].split()[0]); continue
    v = line.strip().split(",")
    rows.append((int(v[0]), int(v[1]), int(v[2]), [int(x) for x in v[3:]]))
def cu_of(xcc, hw):  # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
percu = collections.Counter(cu_of(r[1], r[2]) for r in rows)
print("CUs used", len(percu), "workgroups per CU histogram", sorted(collections.Counter(percu.values()).items()))
perxcc = collections.Counter(r[1] for r in rows)
print("per XCC", sorted(perxcc.items()))
def q(v, p): v = sorted(v); return v[int(p * (len(v) - 1))] / 100.0
names = ["entry"]
for item in range(2):
    for p in range(nph): names += [f"i{item}p{p}_staged", f"i{item}p{p}_loads"]
    names += [f"i{item}_mfma", f"i{item}_epi"]
for k in range(1, min(len(names), 29)):
    d = [r[3][k] - r[3][k - 1] for r in rows if r[3][k] >= 0 and r[3][k - 1] >= 0]
    t = [r[3][k] for r in rows if r[3][k] >= 0]
    if d: print(f"{names[k]:16s} n {len(d):4d}  dt p10 {q(d,.1):6.2f} p50 {q(d,.5):6.2f} p90 {q(d,.9):6.2f} max {q(d,1):6.2f}   at p10 {q(t,.1):6.2f} p50 {q(t,.5):6.2f} p90 {q(t,.9):6.2f} max {q(t,1):6.2f}")
end = [r[3][29] for r in rows]
print("exit at p10 %.2f p50 %.2f p90 %.2f max %.2f" % (q(end, .1), q(end, .5), q(end, .9), q(end, 1)))
# per CU: number of items processed and last exit
items_cu = collections.Counter(); last_cu = collections.defaultdict(int)
for r in rows:
    c = cu_of(r[1], r[2]); n_items = 2 if r[3][2 * nph + 3] >= 0 else 1
    items_cu[c] += n_items; last_cu[c] = max(last_cu[c], r[3][29])
print("items per CU histogram", sorted(collections.Counter(items_cu.values()).items()))
by = collections.defaultdict(list)
for c, n in items_cu.items(): by[n].append(last_cu[c] / 100.0)
for n in sorted(by): print(f"  CUs with {n} items: {len(by[n])}, last exit median {st.median(by[n]):.2f} us, max {max(by[n]):.2f}")
