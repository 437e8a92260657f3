#!/bin/bash
# SQ counter passes over K1 / K3 at M1 (counter passes only).  Output: gpurun_out/pmc_warp_summary.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_warp_a $R/gpurun_out/pmc_warp_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_warp_a -o p -- python $R/scripts/run_warp_only.py > $R/gpurun_out/pmc_warp_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_warp_b -o p -- python $R/scripts/run_warp_only.py > $R/gpurun_out/pmc_warp_b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/pmc_warp_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "warp_"; done > gpurun_out/pmc_warp_summary.txt 2>&1
find gpurun_out/pmc_warp_a gpurun_out/pmc_warp_b -name "*.db" -delete
