cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "not subprocess and not train and not backward and not adam and not ema and not degradation and not dist" 2>&1 | tail -15
