#!/bin/bash
# same-box A/B of the cascade forward: head library vs a variant (arg 1 = variant tag)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for shape in "1184 1600 5" "512 640 5"; do
    echo -n "head      "; python scripts/time_forward.py $shape 2>/dev/null | tail -1
    echo -n "$1  "; CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$1.so python scripts/time_forward.py $shape 2>/dev/null | tail -1
  done
done
