#!/bin/bash
# Kernel trace of the config-5 training step (768x576, N=5, refine): last step of 1 warm-up + 2 timed.  Usage: profile_t5.sh r02
# STORAGE=bf16: the step under the bf16-storage / f32-accumulate policy (bench.py --train-act-storage), output <tag>_t5_bf16_*
tag=${1:-r04}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_t5
timeout 400 rocprofv3 --kernel-trace -d $O/prof_t5 -o t -- python $R/bench.py --workload T5 --steps 3 --warmup 3 --train-act-storage ${STORAGE:-f32} > $O/${tag}_t5${STORAGE:+_$STORAGE}_trace.log 2>&1
cd $R
TOPN=${TOPN:-30} python scripts/kernel_breakdown.py $(find $O/prof_t5 -name "*.db" | head -1) 6 > $O/${tag}_t5${STORAGE:+_$STORAGE}_breakdown.txt 2>&1
find $O/prof_t5 -name "*.db" -delete
