cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
b() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
t() { timeout -k 10 300 python bench.py --workload train --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train $1', d['value'], d['ms_per_step'])"; }
{ for rep in 1 2; do DRS_LIB=$PWD/variants/libdrs_base.so b base; b split2; done
for rep in 1 2; do DRS_LIB=$PWD/variants/libdrs_base.so t base; t split2; done; } > gpurun_out/s_check.txt 2>&1
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "train or grad or golden or stem or wgrad or operator" 2>&1 | tail -3 >> gpurun_out/s_check.txt
cat gpurun_out/s_check.txt
