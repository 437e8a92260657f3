#!/bin/bash
# round 4: first run of the generalised z-marching kernels: parity tests, then per-layer times tiled / z-march / variants
o=gpurun_out/r04; mkdir -p $o
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv3d_split or costreg or deconv3d_split" > $o/zmg1_tests.txt 2>&1
tail -3 $o/zmg1_tests.txt
L="conv0 conv1 conv2 conv3 s2conv0 s1conv0 s3conv0"
CDS_ZMG=0 python scripts/time_conv3d_sbf.py $L > $o/zmg1_tiled.txt 2>&1
CDS_ZMG=1 python scripts/time_conv3d_sbf.py $L > $o/zmg1_zm.txt 2>&1
CDS_ZMG=2 python scripts/time_conv3d_sbf.py conv0 s3conv0 > $o/zmg1_zm_conv0.txt 2>&1
V=cds_mvsnet_amd/_variants/libcdsmvs_hip.scalar.so
CDS_MVSNET_LIB=$V CDS_ZMG=1 python scripts/time_conv3d_sbf.py $L > $o/zmg1_zm_scalar.txt 2>&1
CDS_MVSNET_LIB=$V CDS_ZMG=2 python scripts/time_conv3d_sbf.py conv0 s3conv0 > $o/zmg1_zm_conv0_scalar.txt 2>&1
grep -h "split-bf16" $o/zmg1_tiled.txt $o/zmg1_zm.txt | cut -c1-10,60-130
