#!/bin/bash
# same-box A/B of the headline step: conv2d epilogues with / without the accumulate path
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in "" cds_mvsnet_amd/_variants/libcdsmvs_hip.noaccum.so; do
    CDS_MVSNET_LIB=$v python bench.py --no-pmc --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$v' or 'head', d['value'], d['ms_per_step'])"
  done
done
