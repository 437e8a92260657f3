#!/bin/bash
# Kernel trace + MFMA-utilisation PMC pass of the 1600x1184 cascade forward (BASELINE config 3).  Usage: profile_m3.sh r02
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_m3 $O/prof_m3_mfma
CMD="python $R/scripts/time_forward.py 1184 1600 5"
rocprofv3 --kernel-trace --stats -d $O/prof_m3 -o t -- $CMD > $O/${tag}_m3_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $O/prof_m3_mfma -o p -- $CMD > $O/${tag}_m3_mfma.log 2>&1
cd $R
python scripts/kernel_breakdown.py $(find $O/prof_m3 -name "*.db" | head -1) > $O/${tag}_m3_breakdown.txt 2>&1
python scripts/pmc_summary.py $(find $O/prof_m3_mfma -name "*.db" | head -1) > $O/${tag}_m3_pmc_mfma.txt 2>&1
find $O/prof_m3 $O/prof_m3_mfma -name "*.db" -delete
