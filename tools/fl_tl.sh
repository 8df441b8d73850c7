set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DRS_FL=1 FLC_OUT=/tmp/a.pt timeout -k 10 300 python tools/fl_check.py 2>&1 | grep "oracle"
DRS_LIB=$PWD/variants/libdrs_tl.so DRS_FL=1 timeout -k 10 300 python tools/per_op_table.py --iters 2 > gpurun_out/tl_table.txt 2> gpurun_out/tl_err.txt
grep -v amdgpu.ids gpurun_out/tl_err.txt | tail -24
DRS_FL=1 timeout -k 10 200 python tools/per_op_table.py 2>&1 | grep -v amdgpu.ids | grep "conv\|att \|total"
DRS_FL=1 timeout -k 10 200 python bench.py --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FL=1', d['value'], d['ms_per_step'])"
DRS_FL=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FL=0', d['value'], d['ms_per_step'])"
