# Per-kernel times of one Track X configuration (rocprofv3 --kernel-trace --stats): tools/prof_trackx_one.sh <config> <precision> [extra bench_convnet args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; P=$2; shift 2
D=$R/gpurun_out/prof_one_${CFG}_${P}
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench_convnet.py --config $CFG --precision $P --steps 20 --warmup 4 "$@" > $D.json 2> $D.err || exit 1
find $D -type f ! -name '*kernel_stats.csv' -delete
