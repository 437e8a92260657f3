#!/bin/bash
o=gpurun_out/r04; mkdir -p $o
L="conv0 conv1 conv2 conv3 s2conv0 s3conv0"
for cw in 4 8; do
  CDS_ZMG=2 CDS_ZMG_CW=$cw python scripts/time_conv3d_sbf.py $L > $o/zmg3_cw$cw.txt 2>&1
  echo "== cw $cw"; grep -h "split-bf16" $o/zmg3_cw$cw.txt | sed -E 's/fp32 kernel +[0-9.]+ us \( *[0-9.]+ TF\) +//' | cut -c1-100
done
scripts/ab/r04_zmg_timeline.sh > /dev/null
grep -A12 "conv0: 64" $o/zmg_timeline.txt | head -14
