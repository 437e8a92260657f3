# config-4 stage shapes with the cascade's REAL hypothesis ranges (stage 2: 32 planes x 3.75 = 120; stage 3: 8 x 1.875 = 15)
for sp in 0 1 0 1; do
  CDS_K3_SPLIT_VIEWS=$sp NVIEWS=7 CL=1 EXACT=1 python scripts/time_warp.py 528 960 32 16 600 720 2>&1 | grep -v amdgpu
done
for sp in 0 1; do
  CDS_K3_SPLIT_VIEWS=$sp NVIEWS=7 CL=1 EXACT=1 python scripts/time_warp.py 1056 1920 8 8 600 615 2>&1 | grep -v amdgpu
  CDS_K3_SPLIT_VIEWS=$sp NVIEWS=7 CL=1 EXACT=1 python scripts/time_warp.py 264 480 48 32 2>&1 | grep -v amdgpu
done
