#!/bin/bash
# every kernel of the headline step (rocprofv3 kernel trace, second half of 30 steps)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_m1 -o t -- python $R/bench.py --no-pmc --no-extras --steps 20 --warmup 10 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_m1 -name "*.db" | head -1)
cd $R; TOPN=40 python scripts/kernel_breakdown.py $f 2 > gpurun_out/m1_trace.txt; find gpurun_out/prof_m1 -name "*.db" -delete
