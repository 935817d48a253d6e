# What a phase of k_conv3x3_halo_bf16p costs without parts of it (diagnostic builds -DRCNX_ABL=1..4 under mercer_research_amd/variants/,
# built by hand: hipcc ... -DRCNX_ABL=k -o mercer_research_amd/variants/librcn_hipx_ablk.so csrc/rcn_hipx_api.hip).  Results are wrong by design.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in ${ABL_SET:-0 1 2 3 4}; do
  D=$R/gpurun_out/abl_$k
  rm -rf $D
  if [ $k = 0 ]; then unset RCN_HIPX_TEST_LIB; else export RCN_HIPX_TEST_LIB=$R/mercer_research_amd/variants/librcn_hipx_abl$k.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench_convnet.py --config synth224 --precision bf16_stored --steps 10 --warmup 2 $EXTRA > $D.json 2> $D.err || exit 1
  find $D -type f ! -name '*kernel_stats.csv' -delete
done
