#!/bin/bash
# K3 with more than four source views (two launches, the second accumulates): partial sums read one iteration ahead (shipped) against
# -DCDS_K3_ACC_SERIAL (read at the top of their own iteration), and the 3 + 3 split of six views against 4 + 2 (CDS_K3_SPLIT=4).
# Build: ONLY=warp_lds bash scripts/build_variant.sh accserial -DCDS_K3_ACC_SERIAL=1
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for rep in 1 2; do
  for tag in accserial base; do
    for split in 3 4; do
      lib=$R/cds_mvsnet_amd/_variants/libcdsmvs_hip.$tag.so; [ $tag = base ] && lib=$R/cds_mvsnet_amd/libcdsmvs_hip.so
      export CDS_MVSNET_LIB=$lib TAG=$tag-split$split CL=1 EXACT=1 NVIEWS=7 CDS_K3_SPLIT=$split
      timeout 120 python scripts/time_warp.py 264 480 48 32 2>&1 | tail -1
      timeout 120 python scripts/time_warp.py 528 960 32 16 560 660 2>&1 | tail -1
      timeout 120 python scripts/time_warp.py 1056 1920 8 8 600 615 2>&1 | tail -1
    done
  done
done
