#!/bin/bash
# same-box A/B of the config-5 training step under an environment knob: t5_env_ab.sh CDS_TRAIN_SIDE_STREAM=0
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --workload T5 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('head', d['ms_per_step'])"
  env "$@" python bench.py --workload T5 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"
done
