R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_f_a $R/gpurun_out/pmc_f_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_f_a -o p -- python $R/scripts/ab/fused_only.py > $R/gpurun_out/pmc_f_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_f_b -o p -- python $R/scripts/ab/fused_only.py > $R/gpurun_out/pmc_f_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_f_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "deconv_prob"; done > gpurun_out/pmc_fused_summary.txt 2>&1
find gpurun_out/pmc_f_a gpurun_out/pmc_f_b -name "*.db" -delete
