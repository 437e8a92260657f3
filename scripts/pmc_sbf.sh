#!/bin/bash
# PMC passes over the split-bf16 convolution kernels (counter passes only: no trace domains).  Usage: pmc_sbf.sh <layers...>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sbf_a $R/gpurun_out/pmc_sbf_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_sbf_a -o p -- python $R/scripts/time_conv3d_sbf.py "$@" > $R/gpurun_out/pmc_sbf_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sbf_b -o p -- python $R/scripts/time_conv3d_sbf.py "$@" > $R/gpurun_out/pmc_sbf_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_sbf_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A12 "sbf_kernel"; done > gpurun_out/pmc_sbf_summary.txt 2>&1
find gpurun_out/pmc_sbf_a gpurun_out/pmc_sbf_b -name "*.db" -delete
