# Per-DISPATCH durations of one Track X configuration (rocprofv3 --kernel-trace): tools/prof_trackx_trace.sh <config> <precision> [extra args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; P=$2; shift 2
D=$R/gpurun_out/prof_trace_${CFG}_${P}
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench_convnet.py --config $CFG --precision $P --steps 10 --warmup 2 "$@" > $D.json 2> $D.err || exit 1
find $D -type f ! -name '*kernel_trace.csv' -delete
