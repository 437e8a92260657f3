cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train2d_gpu.py tests/test_train_harness.py -q -x 2>&1 | tail -3
python scripts/ab/r05_aten_train.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_aten_train2.txt; head -70 gpurun_out/r05_aten_train2.txt
bash scripts/ab/r05_t5_ab.sh CDS_X=1
