for v in diet diet_noskip; do
  export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so
  TAG=$v python scripts/ab/fused_prob.py 2>&1 | grep -v amdgpu
done
