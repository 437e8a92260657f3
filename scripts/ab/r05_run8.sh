cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_feat_cl_gpu.py -q -x 2>&1 | tail -12 > gpurun_out/r5_t1.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q -k "featurenet or full_forward or cascade" 2>&1 | tail -6 > gpurun_out/r5_t2.log
timeout 600 python scripts/time_feat_cl.py conv00 > gpurun_out/r5_time_feat_cl.log 2>&1
timeout 600 python scripts/time_forward.py 1184 1600 5 > gpurun_out/r5_fwd_m3_cl.log 2>&1
timeout 600 python scripts/time_forward.py 1056 1920 7 >> gpurun_out/r5_fwd_m3_cl.log 2>&1
cat gpurun_out/r5_t1.log gpurun_out/r5_t2.log gpurun_out/r5_time_feat_cl.log gpurun_out/r5_fwd_m3_cl.log
