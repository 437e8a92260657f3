#!/bin/bash
# Kernel trace of the 1920x1056, N=7 cascade forward (BASELINE config 4 on one GPU) -> gpurun_out/r06_m4_breakdown.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_m4
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_m4 -o t -- python $R/scripts/time_forward.py 1056 1920 7 > $O/r06_m4_trace.log 2>&1
cd $R
db=$(find $O/prof_m4 -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python scripts/kernel_breakdown.py $db > $O/r06_m4_breakdown.txt 2>&1
find $O/prof_m4 -name "*.db" -delete
head -30 $O/r06_m4_breakdown.txt | cut -c1-130
