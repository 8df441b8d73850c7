cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reads_nothing or nan_reaches" 2>&1 | tail -12 | cut -c1-220
