cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_n1 -- python3 $R/bench.py --steps 4096 --warmup 128 > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_n1.err
ls -R $R/gpurun_out/prof_n1 | head -20
