#!/bin/bash
o=gpurun_out/r04; mkdir -p $o
export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_timeline.so CDS_ZMG=2
for cw in 8 4; do for l in conv0 conv2 conv1; do echo "=== cw $cw"; CDS_ZMG_CW=$cw python scripts/ubench/zmg_timeline_run.py $l; done; done > $o/zmg_timeline.txt 2>&1
head -50 $o/zmg_timeline.txt
