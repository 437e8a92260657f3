#!/bin/bash
# K1 / K3 chunk preamble A/B (round 6): all hypothesis loads of a chunk in flight at once
# (shipped) against the rolled range loop + one stage_box per view (-DCDS_RANGE_SERIAL -DCDS_STAGE_SERIAL: the code of rounds 2-6).
# Build here:  ONLY=warp_lds bash scripts/build_variant.sh serial -DCDS_RANGE_SERIAL=1 -DCDS_STAGE_SERIAL=1
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for rep in 1 2; do
  for tag in serial base; do
    lib=$R/cds_mvsnet_amd/_variants/libcdsmvs_hip.$tag.so; [ $tag = base ] && lib=$R/cds_mvsnet_amd/libcdsmvs_hip.so
    CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 timeout 120 python scripts/time_warp.py 2>&1 | tail -1
    CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 timeout 120 python scripts/time_warp.py 296 400 48 32 2>&1 | tail -1
    CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 timeout 120 python scripts/time_warp.py 592 800 32 16 560 660 2>&1 | tail -1
    CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 timeout 120 python scripts/time_warp.py 1184 1600 8 8 600 615 2>&1 | tail -1
  done
done
