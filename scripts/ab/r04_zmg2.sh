#!/bin/bash
# round 4: 4 vs 8 consumer waves in the z-marching kernels (parity first)
o=gpurun_out/r04; mkdir -p $o
for cw in 4 8; do
  CDS_ZMG=2 CDS_ZMG_CW=$cw python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv3d_split or costreg" > $o/zmg2_tests_cw$cw.txt 2>&1
  tail -1 $o/zmg2_tests_cw$cw.txt
done
L="conv0 conv1 conv2 conv3 s2conv0 s3conv0"
for cw in 4 8; do
  CDS_ZMG=2 CDS_ZMG_CW=$cw python scripts/time_conv3d_sbf.py $L > $o/zmg2_cw$cw.txt 2>&1
  echo "== cw $cw"; grep -h "split-bf16" $o/zmg2_cw$cw.txt | sed -E 's/fp32 kernel +[0-9.]+ us \( *[0-9.]+ TF\) +//' | cut -c1-100
done
