#!/bin/bash
# quick A/B of the fused conv11 + prob kernel: product + every dpz_* probe build + the timeline
O=gpurun_out/r04; mkdir -p $O
{
python scripts/ab/dpz_check.py 2>&1 | grep -v amdgpu.ids | tail -14
python scripts/ab/dpz_time.py
for t in $(ls cds_mvsnet_amd/_variants/ | grep "dpz_" | grep -v timeline | sed 's/libcdsmvs_hip.\(.*\).so/\1/'); do
  CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$t.so python scripts/ab/dpz_time.py
done
CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.dpz_timeline.so python scripts/ubench/dpz_timeline_run.py | head -30
} 2>&1 | grep -v amdgpu.ids > $O/dpz_quick.txt
