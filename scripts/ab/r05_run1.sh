set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_feat_cl_gpu.py -x -q -s 2>&1 | tail -40 > gpurun_out/r5_t1.log
timeout 600 python scripts/time_feat_cl.py > gpurun_out/r5_time_feat_cl.log 2>&1
timeout 600 python scripts/time_forward.py 1184 1600 5 > gpurun_out/r5_fwd_m3_cl.log 2>&1
CDS_FEAT_CL=0 timeout 600 python scripts/time_forward.py 1184 1600 5 > gpurun_out/r5_fwd_m3_planar.log 2>&1
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "featurenet or dynconv or full_forward or visibility_layers or row_windows or stage_net" 2>&1 | tail -15 > gpurun_out/r5_t2.log
cat gpurun_out/r5_t1.log gpurun_out/r5_time_feat_cl.log gpurun_out/r5_fwd_m3_cl.log gpurun_out/r5_fwd_m3_planar.log gpurun_out/r5_t2.log
