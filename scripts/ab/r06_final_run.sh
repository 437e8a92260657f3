R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu > $O/suite_final.log 2>&1; tail -3 $O/suite_final.log
timeout 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 300 $O/bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_trace
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o t -- python $R/bench.py --no-extras --steps 5 --cpu-sample 0 > $O/r06_trace.log 2>&1
cd $R
db=$(find $O/prof_trace -name "*.db" | head -1)
if [ -n "$db" ]; then timeout 120 python scripts/make_kernel_stats_md.py $db $O/r06_kernel_stats.md "rocprofv3 --kernel-trace --stats, round 06" "rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --steps 5 --cpu-sample 0 (M1: 640x512, D=192, C=8, N=5; 3 warm-up + 5 timed steps)" > /dev/null; fi
find $O/prof_trace -name "*.db" -delete
head -c 600 $O/bench_final.json
