# PMC passes over the pipelined training step (k_p2_a / k_p2_b); usage (on the GPU box): bash tools/pmc_step.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step/p$i -- python3 $R/bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-e2e > $R/gpurun_out/pmc_step_p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_step k_p2_ $R/gpurun_out/pmc_step.json
