# Where the waves of the Track X bf16-storage step spend their cycles (one SQ PMC pass; per-kernel sums by tools/pmc_kernel_table.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-synth224}
rm -rf $R/gpurun_out/prof_s16_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/prof_s16_pmc -- python3 $R/bench_convnet.py --config $CFG --precision bf16_stored --steps 6 --warmup 2 > $R/gpurun_out/prof_s16_pmc.json 2> $R/gpurun_out/prof_s16_pmc.err || exit 1
find $R/gpurun_out/prof_s16_pmc -type f ! -name '*counter_collection.csv' -delete
ls -la $R/gpurun_out/prof_s16_pmc/*/
