#!/bin/bash
# same-box A/B: stage 1 on a side stream next to FeatureNet's finer levels (CDS_OVERLAP_STAGE1=1) at the cascade shapes
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for shape in "1184 1600 5" "1056 1920 7" "512 640 5"; do
    for v in 0 1; do
      echo -n "overlap2=$v  "; CDS_OVERLAP_STAGE2=$v python scripts/time_forward.py $shape 2>/dev/null | tail -1
    done
  done
done
