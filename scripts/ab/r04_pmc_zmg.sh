#!/bin/bash
# round 4: SQ counters of the z-marching kernels (conv0 / conv1 / conv2 / cascade 16 -> 8 conv0); counter passes only
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
o=$R/gpurun_out/r04; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rm -rf $o/pmc_a $o/pmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $o/pmc_a -o p -- python $R/scripts/time_conv3d_sbf.py conv0 conv1 conv2 s2conv0 > $o/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $o/pmc_b -o p -- python $R/scripts/time_conv3d_sbf.py conv0 conv1 conv2 s2conv0 > $o/pmc_b.log 2>&1
cd $R
for d in a b; do f=$(find $o/pmc_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -A9 "zmg"; done > $o/pmc_zmg.txt 2>&1
rm -rf $o/pmc_a $o/pmc_b
cat $o/pmc_zmg.txt
