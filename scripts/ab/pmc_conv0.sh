R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_c0_a $R/gpurun_out/pmc_c0_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_c0_a -o p -- python $R/scripts/time_conv3d_sbf.py conv0 conv1 conv2 conv11 > $R/gpurun_out/pmc_c0_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_c0_b -o p -- python $R/scripts/time_conv3d_sbf.py conv0 conv1 conv2 conv11 > $R/gpurun_out/pmc_c0_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_c0_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "sbf"; done > gpurun_out/pmc_conv0_summary.txt 2>&1
find gpurun_out/pmc_c0_a gpurun_out/pmc_c0_b -name "*.db" -delete
