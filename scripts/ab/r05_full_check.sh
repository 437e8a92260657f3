cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r05/pytest_gpu.txt
cat gpurun_out/r05/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r05/bench_c.json 2> gpurun_out/r05/bench_c.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/bench_c.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'kernel_ms',d['kernel_ms'])
print({k:v for k,v in d['other_workloads'].items() if not isinstance(v,dict)})
print(d['other_workloads']['M3_K3_roofline_by_stage'])
print('cpu',d['cpu_baseline']['value'], d['cpu_baseline'].get('abs_depth_l1_vs_gpu'))
PY
