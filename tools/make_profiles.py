#!/usr/bin/env python3
"""Assembles profiles/r1_* from the rocprofv3 output directories merged back under gpurun_out/ (run on the dev box).
usage: make_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_json> <bench_under_rocprof_json>"""
import csv, glob, json, shutil, statistics, sys
stats_dir, fdir, wdir, bench_json, prof_json = sys.argv[1:6]
shutil.copy(glob.glob(stats_dir + '/**/*kernel_stats.csv', recursive=True)[0], 'profiles/r1_bench_kernel_stats.csv')
shutil.copy(bench_json, 'profiles/r1_bench_n1.json')
shutil.copy(prof_json, 'profiles/r1_bench_under_rocprof.json')
tr = list(csv.DictReader(open(glob.glob(stats_dir + '/**/*kernel_trace.csv', recursive=True)[0])))
by = {}
for r in tr:
    if 'rcn::' in r['Kernel_Name']:
        by.setdefault(r['Kernel_Name'].split('(')[0], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
lines = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (MI355X, round 1)",
         "# per-dispatch durations (ns) over the whole run; k_p2_b + k_p2_a are the two kernels of one train_batch",
         "kernel,calls,median_ns,mean_ns,p10_ns,p90_ns"]
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    lines.append(f"{n},{len(v)},{statistics.median(v):.0f},{sum(v)/len(v):.0f},{v[len(v)//10]},{v[len(v)*9//10]}")
open('profiles/r1_bench_kernel_trace_summary.csv', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines[:10]))
out = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py --steps 512 "
               "--warmup 64 --no-cpu-baseline; counter unit KiB per dispatch, median over dispatches.  gfx950: FETCH_SIZE tallies 128-B requests at 64 B, "
               "i.e. reads exactly 1/2 of a 16-B-per-lane (or wave-contiguous 4-B) stream (MI355X_MICROARCH.md, HBM) -- confirmed here by k_standardize "
               "(25096 KiB counted for a 50176 KiB f32 stream) -- while 64-B gather pieces (k_pack_epoch) are counted in full.  "
               "hbm_bytes_per_launch = (2*FETCH + WRITE)*1024, hbm_bytes_per_launch_low = (FETCH + WRITE)*1024."}
for name, d in (('FETCH_SIZE', fdir), ('WRITE_SIZE', wdir)):
    rows = list(csv.DictReader(open(glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0])))
    acc = {}
    for r in rows:
        if 'rcn::' in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rcn::', '').split('<')[0]
            acc.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in acc.items():
        out.setdefault(k, {})[name + "_KiB_median"] = round(statistics.median(v), 1)
        out[k]["dispatches_" + name] = len(v)
for k, v in out.items():
    if isinstance(v, dict) and 'FETCH_SIZE_KiB_median' in v and 'WRITE_SIZE_KiB_median' in v:
        f, w = v['FETCH_SIZE_KiB_median'], v['WRITE_SIZE_KiB_median']
        v['hbm_bytes_per_launch_low'] = int((f + w) * 1024)
        v['hbm_bytes_per_launch'] = int((2 * f + w) * 1024)
json.dump(out, open('profiles/r1_pmc_summary.json', 'w'), indent=1)
for k in ('k_p2_a', 'k_p2_b', 'k_pack_epoch'):
    print(k, out.get(k))
