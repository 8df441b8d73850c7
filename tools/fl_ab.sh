set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DRS_FL=1 FLC_OUT=/tmp/a.pt timeout -k 10 300 python tools/fl_check.py 2>&1 | grep "oracle"
b() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
DRS_FL=1 b "relaxed mover polls"
DRS_LIB=$PWD/variants/libdrs_acq.so DRS_FL=1 b "acquire polls"
done
