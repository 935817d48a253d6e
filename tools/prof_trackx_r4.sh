# Track X, round 4: (1) MFMA-busy of the convolution kernels (CIFAR shape, fp32): one PMC pass over bench_convnet.py, turned into
# profiles/r4_trackx_mfma_pmc.json by tools/mfma_pmc_summary.py; (2) the bench lines of every configuration and precision, bf16 storage
# included; (3) per-kernel times of the synth-224 step in bf16 mode and with bf16 storage (rocprofv3 --kernel-trace --stats) and one SQ
# counter pass over the bf16-storage step.  Run on the GPU box through gpurun; tools/make_profiles_trackx_r4.py copies the results.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof4_trackx
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof4_trackx -- python3 $R/bench_convnet.py --config cifar --steps 20 --warmup 10 > $R/gpurun_out/prof4_trackx.json 2> $R/gpurun_out/prof4_trackx.err || exit 1
find $R/gpurun_out/prof4_trackx -type f ! -name '*counter_collection.csv' -delete
python3 $R/bench_convnet.py --config cifar > $R/gpurun_out/r4_trackx_bench_cifar_f32.json 2>/dev/null
python3 $R/bench_convnet.py --config cifar --precision bf16 > $R/gpurun_out/r4_trackx_bench_cifar_bf16.json 2>/dev/null
python3 $R/bench_convnet.py --config cifar --precision bf16_stored > $R/gpurun_out/r4_trackx_bench_cifar_bf16_stored.json 2>/dev/null
python3 $R/bench_convnet.py --config synth224 --steps 30 --warmup 6 > $R/gpurun_out/r4_trackx_bench_224_f32.json 2>/dev/null
python3 $R/bench_convnet.py --config synth224 --precision bf16 --steps 30 --warmup 6 > $R/gpurun_out/r4_trackx_bench_224_bf16.json 2>/dev/null
python3 $R/bench_convnet.py --config synth224 --precision bf16_stored --steps 30 --warmup 6 > $R/gpurun_out/r4_trackx_bench_224_bf16_stored.json 2>/dev/null
python3 $R/bench_convnet.py --config mnist --batch 4096 --precision bf16 > $R/gpurun_out/r4_trackx_bench_mnist4096_bf16.json 2>/dev/null
python3 $R/bench_convnet.py --config mnist --batch 4096 --precision bf16_stored > $R/gpurun_out/r4_trackx_bench_mnist4096_bf16_stored.json 2>/dev/null
for m in bf16 bf16_stored; do
  D=$R/gpurun_out/prof4_trackx_stats_$m
  rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench_convnet.py --config synth224 --precision $m --steps 20 --warmup 4 > $D.json 2> $D.err || exit 1
  find $D -type f ! -name '*kernel_stats.csv' -delete
done
D=$R/gpurun_out/prof4_trackx_sq
rm -rf $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $D -- python3 $R/bench_convnet.py --config synth224 --precision bf16_stored --steps 6 --warmup 2 > $D.json 2> $D.err || exit 1
find $D -type f ! -name '*counter_collection.csv' -delete
cat $R/gpurun_out/r4_trackx_bench_*.json | cut -c1-330
