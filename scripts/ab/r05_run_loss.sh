cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train2d_gpu.py -q -x -k "fused_final_loss or feat_target or full_training" 2>&1 | tail -15
timeout 300 python scripts/time_train_step.py 2>&1 | grep -v amdgpu | tail -3
