cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain_is_deterministic or ring_protocol or nan_reaches" 2>&1 | tail -12
