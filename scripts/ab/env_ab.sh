#!/bin/bash
# same-box A/B of the headline step under environment knobs: env_ab.sh "CDS_PROB_MFMA=1" ["CDS_PROB_MFMA=1 CDS_PROB_TY=8" ...]
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('head', round(d['value'], 2), round(d['ms_per_step'], 4), d.get('kernel_ms', {}).get('costreg'))"
  for v in "$@"; do
    env $v python bench.py --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v', round(d['value'], 2), round(d['ms_per_step'], 4), d.get('kernel_ms', {}).get('costreg'))"
  done
done
