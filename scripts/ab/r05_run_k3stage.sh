cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in oldstage ""; do
  if [ -n "$v" ]; then export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so; else unset CDS_MVSNET_LIB; fi
  export TAG=${v:-batched}
  CL=1 python scripts/time_warp.py 2>&1 | grep K1
  CL=1 python scripts/time_warp.py 296 400 48 32 2>&1 | grep K1
  CL=1 python scripts/time_warp.py 592 800 32 16 560 640 2>&1 | grep K1
  CL=1 python scripts/time_warp.py 1184 1600 8 8 595 605 2>&1 | grep K1
done; done
