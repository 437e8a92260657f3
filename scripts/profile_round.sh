#!/bin/bash
# Round profile on the GPU box (counter passes separate from the trace pass; no trace domains with --pmc):
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> gpurun_out/rNN_kernel_stats.md
#   2. PMC MFMA-pipe utilisation of the same command                      -> gpurun_out/rNN_pmc_mfma.txt
#   3. PMC FETCH_SIZE / WRITE_SIZE passes of K1 / K3 (+ calibration kernels) -> gpurun_out/pmc_traffic.{md,json}
# Usage: scripts/profile_round.sh r02
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_trace $O/prof_mfma $O/prof_fetch $O/prof_write
CMD="python $R/bench.py --no-extras --steps 5 --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o t -- $CMD > $O/${tag}_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $O/prof_mfma -o p -- $CMD > $O/${tag}_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o p -- python $R/scripts/run_k3_traffic.py > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o p -- python $R/scripts/run_k3_traffic.py > $O/${tag}_write.log 2>&1
cd $R
python scripts/make_kernel_stats_md.py $(find $O/prof_trace -name "*.db" | head -1) $O/${tag}_kernel_stats.md "rocprofv3 --kernel-trace --stats, round ${tag#r}" "rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --steps 5 --cpu-sample 0 (M1: 640x512, D=192, C=8, N=5; 3 warm-up + 5 timed steps)" > /dev/null
python scripts/pmc_summary.py $(find $O/prof_mfma -name "*.db" | head -1) > $O/${tag}_pmc_mfma.txt
python scripts/make_pmc_traffic.py $(find $O/prof_fetch -name "*.db" | head -1) $(find $O/prof_write -name "*.db" | head -1) $O/pmc_traffic.md $O/pmc_traffic.json > /dev/null
find $O/prof_trace $O/prof_mfma $O/prof_fetch $O/prof_write -name "*.db" -delete
