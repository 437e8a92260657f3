cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05b_gpu_suite.txt
python bench.py > gpurun_out/r05b_bench_line.json 2> gpurun_out/r05b_bench_err.txt
for k in 0 1; do CDS_OVERLAP_STAGE2=$k python bench.py --workload M3 --steps 6 --warmup 3 --no-extras --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('stage2 overlap', $k, d['ms_per_step'])"; done > gpurun_out/r05b_m3_overlap.txt 2>&1
for k in 0 1; do CDS_OVERLAP_STAGE2=$k python bench.py --workload M4 --steps 6 --warmup 3 --no-extras --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('M4 stage2 overlap', $k, d['ms_per_step'])"; done >> gpurun_out/r05b_m3_overlap.txt 2>&1
