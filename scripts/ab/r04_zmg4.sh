#!/bin/bash
# round 4: consumer priorities / skew A/B of the z-marching kernels (cw 8 and 4)
o=gpurun_out/r04; mkdir -p $o
L="conv0 conv1 conv2 conv3 s2conv0 s3conv0"
for v in base noskew cprio0; do
  lib=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so; [ $v = base ] && lib=cds_mvsnet_amd/libcdsmvs_hip.so
  for cw in 8 4; do
    CDS_MVSNET_LIB=$lib CDS_ZMG=2 CDS_ZMG_CW=$cw python scripts/time_conv3d_sbf.py $L 2>&1 | grep "split-bf16" | sed -E "s/^([a-z0-9]+):.*split-bf16 +([0-9.]+) us.*/$v cw$cw \1 \2/"
  done
done | tee $o/zmg4.txt
