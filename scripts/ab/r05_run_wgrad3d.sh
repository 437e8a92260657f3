cd $GRAFT_REPO_ROOT
for n in 256 512 1024 2048; do echo "--- CDS_WG3_WGS=$n"; CDS_WG3_WGS=$n python scripts/ab/r05_time_wgrad3d.py 2>&1 | grep -v amdgpu.ids | grep "conv\|total" | sed 's/(.*//' | awk '{printf "%s ", $(NF-1)} END {print ""}'; done
