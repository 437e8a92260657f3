for v in base prio1 prio3 nts base; do
  echo "=== $v"
  if [ $v = base ]; then unset CDS_MVSNET_LIB; else export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so; fi
  python scripts/time_conv3d_sbf.py 2>&1 | grep -v amdgpu.ids | sed -E 's/fp32 kernel +[0-9.]+ us \( *[0-9.]+ TF\) +//'
done
