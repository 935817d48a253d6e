# MFMA-busy of the Track X convolution kernels (CIFAR shape, fp32): one PMC pass over bench_convnet.py.  Run on the GPU box through gpurun;
# tools/mfma_pmc_summary.py turns the counter file into profiles/r4_trackx_mfma_pmc.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof4_trackx
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof4_trackx -- python3 $R/bench_convnet.py --config cifar --steps 20 --warmup 10 > $R/gpurun_out/prof4_trackx.json 2> $R/gpurun_out/prof4_trackx.err || exit 1
find $R/gpurun_out/prof4_trackx -type f ! -name '*counter_collection.csv' -delete
python3 $R/bench_convnet.py --config cifar > $R/gpurun_out/r4_trackx_bench_cifar_f32.json 2>/dev/null
python3 $R/bench_convnet.py --config cifar --precision bf16 > $R/gpurun_out/r4_trackx_bench_cifar_bf16.json 2>/dev/null
python3 $R/bench_convnet.py --config synth224 --steps 30 --warmup 6 > $R/gpurun_out/r4_trackx_bench_224_f32.json 2>/dev/null
python3 $R/bench_convnet.py --config synth224 --precision bf16 --steps 30 --warmup 6 > $R/gpurun_out/r4_trackx_bench_224_bf16.json 2>/dev/null
python3 $R/bench_convnet.py --config mnist --batch 4096 --precision bf16 > $R/gpurun_out/r4_trackx_bench_mnist4096_bf16.json 2>/dev/null
cat $R/gpurun_out/r4_trackx_bench_*.json | cut -c1-400
