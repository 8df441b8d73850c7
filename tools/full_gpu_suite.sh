cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/full_suite.txt 2>&1
tail -12 gpurun_out/full_suite.txt
timeout -k 10 200 python tools/per_op_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/per_op_now.txt; grep "conv\|att \|total" gpurun_out/per_op_now.txt
for i in 1 2; do timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
