# Per-kernel durations of the Track X MNIST-shape step (B = 256 fp32 and B = 4096 bf16): rocprofv3 --kernel-trace --stats over bench_convnet.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "256 fp32" "4096 bf16"; do
  set -- $c
  rm -rf $R/gpurun_out/prof3_mnist_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_mnist_$1 -- python3 $R/bench_convnet.py --config mnist --batch $1 --precision $2 --steps 50 --warmup 16 > $R/gpurun_out/prof3_mnist_$1.json 2> $R/gpurun_out/prof3_mnist_$1.err || exit 1
  f=$(find $R/gpurun_out/prof3_mnist_$1 -name '*kernel_stats.csv' | head -1)
  cp $f $R/gpurun_out/r3_trackx_mnist$1_kernel_stats.csv
  find $R/gpurun_out/prof3_mnist_$1 -type f -delete
done
