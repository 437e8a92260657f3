#!/bin/bash
# Kernel trace of the 640x512, N=5 cascade forward -> gpurun_out/r06_m2_breakdown.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_m2
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_m2 -o t -- python $R/scripts/time_forward.py 512 640 5 > $O/r06_m2_trace.log 2>&1
cd $R
db=$(find $O/prof_m2 -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python scripts/kernel_breakdown.py $db > $O/r06_m2_breakdown.txt 2>&1
find $O/prof_m2 -name "*.db" -delete
head -64 $O/r06_m2_breakdown.txt | cut -c1-120
