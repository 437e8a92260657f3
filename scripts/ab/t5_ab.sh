#!/bin/bash
# same-box A/B of the config-5 training step: head library vs a variant (arg 1 = variant tag)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in "" cds_mvsnet_amd/_variants/libcdsmvs_hip.$1.so; do
    CDS_MVSNET_LIB=$v python bench.py --workload T5 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('${v:-head}', d['ms_per_step'])"
  done
done
