R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export CDS_PROB_TY=${CDS_PROB_TY:-8}
rm -rf $R/gpurun_out/pmc_p_a $R/gpurun_out/pmc_p_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_p_a -o p -- python $R/scripts/ab/prob_time.py > $R/gpurun_out/pmc_p_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_p_b -o p -- python $R/scripts/ab/prob_time.py > $R/gpurun_out/pmc_p_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_p_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "prob_sbf"; done > gpurun_out/pmc_prob_summary.txt 2>&1
find gpurun_out/pmc_p_a gpurun_out/pmc_p_b -name "*.db" -delete
