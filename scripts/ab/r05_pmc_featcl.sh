#!/bin/bash
# PMC passes over the channels-last FeatureNet kernels (counter passes only: no trace domains).  Usage: r05_pmc_featcl.sh [layers...]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fcl_a $R/gpurun_out/pmc_fcl_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_fcl_a -o p -- python $R/scripts/time_feat_cl.py "$@" > $R/gpurun_out/pmc_fcl_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_fcl_b -o p -- python $R/scripts/time_feat_cl.py "$@" > $R/gpurun_out/pmc_fcl_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_fcl_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "dynconv_cl_kernel\|dynconv_branches_sbf"; done > gpurun_out/pmc_fcl_summary.txt 2>&1
find gpurun_out/pmc_fcl_a gpurun_out/pmc_fcl_b -name "*.db" -delete
