cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_feat_cl_gpu.py -q 2>&1 | tail -6 > gpurun_out/r5_t1.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q 2>&1 | tail -6 > gpurun_out/r5_t2.log
timeout 600 python scripts/time_forward.py 1184 1600 5 > gpurun_out/r5_fwd_m3_cl.log 2>&1
timeout 600 python scripts/time_forward.py 1056 1920 7 >> gpurun_out/r5_fwd_m3_cl.log 2>&1
timeout 600 python scripts/time_forward.py 512 640 5 >> gpurun_out/r5_fwd_m3_cl.log 2>&1
AGGR_ONLY="cl " python scripts/ab/r05_aggressor.py 2>&1 | grep -v amdgpu.ids | grep aggressor > gpurun_out/r5_aggr.log
cat gpurun_out/r5_t1.log gpurun_out/r5_t2.log gpurun_out/r5_fwd_m3_cl.log gpurun_out/r5_aggr.log
