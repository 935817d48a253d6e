# Round-4 profile collection (same recipe as round 3) (run on the GPU box through gpurun): kernel trace + stats of the bench command, then the two PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950) on a shorter run of the same workload.  Output under
# gpurun_out/; tools/make_profiles_r4.py turns it into profiles/r4_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof4_stats $R/gpurun_out/prof4_fetch $R/gpurun_out/prof4_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4_stats -- python3 $R/bench.py --steps 4096 --warmup 128 > $R/gpurun_out/r4_bench_under_rocprof.json 2> $R/gpurun_out/prof4_stats.err || exit 1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof4_fetch -- python3 $R/bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-extras > $R/gpurun_out/prof4_fetch.json 2> $R/gpurun_out/prof4_fetch.err || exit 1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof4_write -- python3 $R/bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-extras > $R/gpurun_out/prof4_write.json 2> $R/gpurun_out/prof4_write.err || exit 1
echo "write done"
python3 $R/bench.py --steps 4096 --warmup 128 > $R/gpurun_out/r4_bench_n1.json 2> $R/gpurun_out/r4_bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r4_bench_n1_steps20.json 2> $R/gpurun_out/r4_bench_n1_steps20.err
# keep only what make_profiles_r4.py reads (the merged directory is capped at 64 MiB)
find $R/gpurun_out/prof4_stats $R/gpurun_out/prof4_fetch $R/gpurun_out/prof4_write -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' -delete
du -sh $R/gpurun_out/prof4_stats $R/gpurun_out/prof4_fetch $R/gpurun_out/prof4_write
