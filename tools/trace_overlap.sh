# Do k_pack_epoch and k_xcd_epoch overlap in time?  kernel trace of a short bench run, then the timeline of the last few dispatches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tr_ov
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_ov -- python3 $R/bench.py --steps 2048 --warmup 128 --no-extras --no-cpu-baseline --no-e2e > /dev/null 2> $R/gpurun_out/tr_ov.err || { tail -5 $R/gpurun_out/tr_ov.err; exit 1; }
python3 - $R/gpurun_out/tr_ov <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_pack_epoch" in r["Kernel_Name"] or "k_xcd_epoch" in r["Kernel_Name"] or "k_shuffle" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-40]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} .. {(int(r["End_Timestamp"]) - t0) / 1e3:10.1f} us  q={r.get("Queue_Id", "?"):>3}  {r["Kernel_Name"].split("(")[0][-40:]}')
PY
find $R/gpurun_out/tr_ov -type f -delete
