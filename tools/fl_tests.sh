cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fl.py -x -q -m gpu 2>&1 | tail -12
