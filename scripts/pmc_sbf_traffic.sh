#!/bin/bash
# Fabric traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the split-bf16 convolution kernels.  Usage: pmc_sbf_traffic.sh <layers...>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sbf_t $R/gpurun_out/pmc_sbf_u
timeout 150 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_sbf_t -o p -- python $R/scripts/time_conv3d_sbf.py "$@" > $R/gpurun_out/pmc_sbf_t.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_sbf_u -o p -- python $R/scripts/time_conv3d_sbf.py "$@" > $R/gpurun_out/pmc_sbf_u.log 2>&1
cd $R
for d in t u; do f=$(find gpurun_out/pmc_sbf_$d -name "*.db" | head -1); python scripts/pmc_summary.py $f | grep -B2 -A8 "sbf_\|zmg\|k3_pipe\|k3_mfma"; done > gpurun_out/pmc_sbf_traffic.txt 2>&1
find gpurun_out/pmc_sbf_t gpurun_out/pmc_sbf_u -name "*.db" -delete
