cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base ""; do
  if [ -n "$v" ]; then export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so; else unset CDS_MVSNET_LIB; fi
  echo "--- ${v:-new}"
  python scripts/time_feat_cl.py conv01 conv10 conv20 out2 out3 2>&1 | grep -v amdgpu | grep -i "conv\|out"
  python scripts/time_forward.py 1184 1600 5 2>&1 | grep full
done; done
unset CDS_MVSNET_LIB
python -m pytest tests/test_feat_cl_gpu.py -q -x 2>&1 | tail -2
