# Per-kernel durations of one Track X configuration (default: CIFAR shape, fp32) under rocprofv3 --kernel-trace --stats.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-cifar}; shift
rm -rf $R/gpurun_out/px_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/px_stats -- python3 $R/bench_convnet.py --config $CFG --steps 20 --warmup 5 "$@" > /dev/null 2> $R/gpurun_out/px_stats.err || { tail -5 $R/gpurun_out/px_stats.err; exit 1; }
python3 - $R/gpurun_out/px_stats 25 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:20]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {int(r["Calls"])/int(sys.argv[2]):5.1f}/step {float(r["TotalDurationNs"])/tot*100:5.1f}%  {r["Name"].split("(")[0][:70]}')
print(f"sum per step: {tot/1e3/int(sys.argv[2]):.1f} us")
PY
find $R/gpurun_out/px_stats -type f -delete
