cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AGGR_ONLY="cl " python scripts/ab/r05_aggressor.py 2>&1 | grep -v amdgpu.ids | grep aggressor > gpurun_out/r5_aggr.log
python scripts/ab/r05_stress_cl.py 3 2>&1 | grep -v amdgpu.ids | grep "bad\|TOTAL" > gpurun_out/r5_stress.log
python scripts/ab/r05_det2.py 2>&1 | grep -v amdgpu.ids | grep "channels-last" > gpurun_out/r5_det2.log
timeout 1500 python -m pytest tests/test_feat_cl_gpu.py tests/test_train_harness.py -q 2>&1 | tail -6 > gpurun_out/r5_t1.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q 2>&1 | tail -6 > gpurun_out/r5_t2.log
timeout 600 python scripts/time_feat_cl.py > gpurun_out/r5_time_feat_cl.log 2>&1
timeout 600 python scripts/time_forward.py 1184 1600 5 > gpurun_out/r5_fwd_m3_cl.log 2>&1
cat gpurun_out/r5_aggr.log; tail -1 gpurun_out/r5_stress.log; cat gpurun_out/r5_det2.log gpurun_out/r5_t1.log gpurun_out/r5_t2.log gpurun_out/r5_time_feat_cl.log gpurun_out/r5_fwd_m3_cl.log
