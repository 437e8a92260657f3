cd $GRAFT_REPO_ROOT
python scripts/ab/r05_time_wgrad3d.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests -q -x -m gpu -k "wgrad or cost_regularization or costreg" 2>&1 | tail -2
