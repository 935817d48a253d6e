#!/usr/bin/env python3
"""Assembles profiles/r4_* from the rocprofv3 output merged back under gpurun_out/ by tools/prof_r4.sh (run on the dev box)."""
import csv, glob, hashlib, json, os, shutil, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
G = "gpurun_out"
def one(pat):
    m = glob.glob(pat, recursive=True)
    assert m, pat
    return max(m, key=os.path.getmtime)          # gpurun merges every call's files into the same directory: the newest run counts
sys.path.insert(0, ROOT)
from bench import csrc_sha16          # the fingerprint bench.py compares with: the sources librcn_hip.so is built from
shutil.copy(one(f"{G}/prof4_stats/**/*kernel_stats.csv"), "profiles/r4_bench_kernel_stats.csv")
for f in ("r4_bench_n1.json", "r4_bench_n1_steps20.json", "r4_bench_under_rocprof.json"):
    line = [l for l in open(f"{G}/{f}").read().splitlines() if l.startswith("{")][-1]      # (a log of the run may precede the line)
    open(f"profiles/{f}", "w").write(line + "\n")
tr = list(csv.DictReader(open(one(f"{G}/prof4_stats/**/*kernel_trace.csv"))))
by = {}
for r in tr:
    if 'rcn::' in r['Kernel_Name']:
        by.setdefault(r['Kernel_Name'].split('(')[0], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
lines = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 4096 --warmup 128   (MI355X, round 4)",
         "# per-dispatch durations (ns) over the whole run; k_xcd_epoch runs ALL steps of one epoch segment (64 in the bench): its duration / 64 is the step",
         "kernel,calls,median_ns,mean_ns,p10_ns,p90_ns"]
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    lines.append(f"{n},{len(v)},{statistics.median(v):.0f},{sum(v)/len(v):.0f},{v[len(v)//10]},{v[len(v)*9//10]}")
open('profiles/r4_bench_kernel_trace_summary.csv', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
out = {"_how": "tools/prof_r4.sh: rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py --steps 512 "
               "--warmup 64 --no-cpu-baseline --no-extras; counter unit KiB per dispatch, median over the dispatches of 64 steps (k_xcd_epoch) / all dispatches (others).  "
               "gfx950: FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly 1/2 of a 16-B-per-lane (or wave-contiguous 4-B) stream (MI355X_MICROARCH.md, HBM); "
               "64-B gather pieces are counted in full.  hbm_bytes_per_launch = (2*FETCH + WRITE)*1024 (every read of the resident kernel is a 16-B-per-lane stream), "
               "hbm_bytes_per_launch_low = (FETCH + WRITE)*1024.",
       "csrc_sha16": csrc_sha16()}
# the steps per dispatch of k_xcd_epoch come from the trace of the same run: the bench issues 64-step launches in steady state
for name, d in (('FETCH_SIZE', 'prof4_fetch'), ('WRITE_SIZE', 'prof4_write')):
    rows = list(csv.DictReader(open(one(f"{G}/{d}/**/*counter_collection.csv"))))
    acc = {}
    for r in rows:
        if 'rcn::' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rcn::', '').split('<')[0]
            acc.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in acc.items():
        if k == 'k_xcd_epoch':
            big = [x for x in v if x >= 0.5 * max(v)]           # the 64-step launches (warm-up / roofline legs also issue shorter ones)
            v = big
        out.setdefault(k, {})[name + "_KiB_median"] = round(statistics.median(v), 1)
        out[k]["dispatches_" + name] = len(v)
for k, v in out.items():
    if isinstance(v, dict) and 'FETCH_SIZE_KiB_median' in v and 'WRITE_SIZE_KiB_median' in v:
        f, w = v['FETCH_SIZE_KiB_median'], v['WRITE_SIZE_KiB_median']
        v['hbm_bytes_per_launch_low'] = int((f + w) * 1024)
        v['hbm_bytes_per_launch'] = int((2 * f + w) * 1024)
if 'k_xcd_epoch' in out:
    out['k_xcd_epoch']['steps_per_launch'] = 64
    out['k_xcd_epoch']['hbm_bytes_per_step'] = out['k_xcd_epoch']['hbm_bytes_per_launch'] // 64
json.dump(out, open('profiles/r4_pmc_summary.json', 'w'), indent=1)
for k in ('k_xcd_epoch', 'k_p2_a', 'k_p2_b', 'k_pack_epoch', 'k_features_cpcp'):
    print(k, out.get(k))
