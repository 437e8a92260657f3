#!/bin/bash
# K3 plane-loop probes: how much of the kernel is LDS reads, how much VALU?  Wrong results, right timing.
# Build here (no GPU):  bash scripts/ab/r06_k3_probe.sh build     Run on the GPU box:  bash scripts/ab/r06_k3_probe.sh
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  ONLY="warp_lds" bash $R/scripts/build_variant.sh k3half -DCDS_PROBE_K3=1
  ONLY="warp_lds" bash $R/scripts/build_variant.sh k3nolds -DCDS_PROBE_K3=2
  ONLY="warp_lds" bash $R/scripts/build_variant.sh k3nopos -DCDS_PROBE_K3_POS=1
  ONLY="warp_lds" bash $R/scripts/build_variant.sh k3noposhalf -DCDS_PROBE_K3_POS=1 -DCDS_PROBE_K3=1
  exit 0
fi
cd $R
for rep in 1 2; do
  for tag in base k3half k3nolds k3nopos k3noposhalf; do
    lib=$R/cds_mvsnet_amd/_variants/libcdsmvs_hip.$tag.so; [ $tag = base ] && lib=$R/cds_mvsnet_amd/libcdsmvs_hip.so
    CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 python scripts/time_warp.py 2>&1 | tail -1
  done
done
for tag in base k3half k3nolds; do
  lib=$R/cds_mvsnet_amd/_variants/libcdsmvs_hip.$tag.so; [ $tag = base ] && lib=$R/cds_mvsnet_amd/libcdsmvs_hip.so
  CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 python scripts/time_warp.py 296 400 48 32 2>&1 | tail -1
  CDS_MVSNET_LIB=$lib TAG=$tag CL=1 EXACT=1 python scripts/time_warp.py 592 800 32 16 560 660 2>&1 | tail -1
done
