cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -15
