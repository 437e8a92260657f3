#!/bin/bash
# Kernel-time sum vs wall span of the 640x512 cascade forward (is it launch-bound?  no: 5.3 ms of kernels per 4.6 ms forward
# under the profiler).  Usage on the GPU box: bash scripts/profile_m2.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out; rm -rf $O/prof_m2
timeout 300 rocprofv3 --kernel-trace -d $O/prof_m2 -o t -- python $R/scripts/time_forward.py 512 640 5 > $O/m2_trace.log 2>&1
cd $R; python scripts/kernel_breakdown.py $(find $O/prof_m2 -name "*.db" | head -1) > $O/m2_breakdown.txt 2>&1; find $O/prof_m2 -name "*.db" -delete
tail -3 $O/m2_trace.log; tail -1 $O/m2_breakdown.txt
