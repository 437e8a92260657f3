#!/bin/bash
# round 4: s_memtime stage timeline of one workgroup of the z-marching kernels (probe library: scripts/ubench/zmg_timeline_build.py)
o=gpurun_out/r04; mkdir -p $o
export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_timeline.so
for l in conv0 conv2 conv1; do python scripts/ubench/zmg_timeline_run.py $l; done > $o/zmg_timeline.txt 2>&1
head -50 $o/zmg_timeline.txt
