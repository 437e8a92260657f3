cd $GRAFT_REPO_ROOT
for n in 512 1024 2048; do echo "--- CDS_WG2_WGS=$n"; CDS_WG2_WGS=$n python scripts/ab/r05_time_wgrad2d.py 2>&1 | grep -v amdgpu.ids | grep "conv\|out\|total" | sed 's/(.*//' | awk '{printf "%s ", $(NF-1)} END {print ""}'; done
python scripts/ab/r05_time_wgrad2d.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_train2d_gpu.py -q -x 2>&1 | tail -2
