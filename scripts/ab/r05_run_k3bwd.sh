cd $GRAFT_REPO_ROOT
echo "--- tree"; python scripts/ab/r05_time_k3_bwd.py 2>&1 | grep "stage"
python -m pytest tests -q -x -m gpu -k "aggregate and (backward or bwd or grad)" 2>&1 | tail -3
python -m pytest tests/test_train2d_gpu.py tests/test_train_harness.py -q -x 2>&1 | tail -3
bash scripts/ab/r05_t5_ab.sh CDS_X=1
