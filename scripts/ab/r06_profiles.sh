#!/bin/bash
# round 5: the profile set committed under profiles/r06_* (kernel stats, MFMA utilisation, HBM traffic, cascade breakdown, T5 breakdown, bench line)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O/r06
cd $R
scripts/profile_round.sh r06 > /dev/null 2>&1
TOPN=40 scripts/profile_m3.sh r06 > /dev/null 2>&1
TOPN=45 scripts/profile_t5.sh r06 > /dev/null 2>&1
cp $O/r06_kernel_stats.md $O/r06_pmc_mfma.txt $O/pmc_traffic.md $O/pmc_traffic.json $O/r06_m3_breakdown.txt $O/r06_m3_pmc_mfma.txt $O/r06_t5_breakdown.txt $O/r06/ 2>/dev/null
python bench.py > $O/r06/bench_b.json 2> $O/r06/bench_b.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06/bench_b.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'traffic',d['roofline']['traffic'],'kernel_ms',d['kernel_ms'])
print({k:v for k,v in d['other_workloads'].items() if not isinstance(v,dict)})
print({k:v['frac'] for k,v in d['other_workloads']['M3_K3_roofline_by_stage'].items()})
PY
