# Per-kernel times of the synth-224 step in bf16 mode and with bf16 storage (rocprofv3 --kernel-trace --stats), side by side.
# Run on the GPU box through gpurun; the two *_kernel_stats.csv files land in gpurun_out/prof_s16_{b,s}/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-synth224}
for m in b s; do
  P=bf16; [ $m = s ] && P=bf16_stored
  rm -rf $R/gpurun_out/prof_s16_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s16_$m -- python3 $R/bench_convnet.py --config $CFG --precision $P --steps 20 --warmup 4 > $R/gpurun_out/prof_s16_$m.json 2> $R/gpurun_out/prof_s16_$m.err || exit 1
  find $R/gpurun_out/prof_s16_$m -type f ! -name '*kernel_stats.csv' -delete
done
