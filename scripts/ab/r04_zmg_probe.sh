#!/bin/bash
# round 4: where the z-marching kernels' time goes (timing probes built by scripts/ubench/zmg_probe_build.sh: wrong results by construction)
o=gpurun_out/r04; mkdir -p $o
L="conv0 conv1 conv2 s2conv0"
for v in base krep2 nostore noload; do
  lib=cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_$v.so; [ $v = base ] && lib=cds_mvsnet_amd/libcdsmvs_hip.so
  CDS_MVSNET_LIB=$lib python scripts/time_conv3d_sbf.py $L 2>&1 | grep "split-bf16" | sed -E "s/^([a-z0-9]+):.*split-bf16 +([0-9.]+) us.*/$v \1 \2/"
done | tee $o/zmg_probe.txt
