# A/B of library builds on ONE box: tools/ab.sh NAME [NAME ...]  (variants/libdrs_NAME.so; "default" = the in-tree build)
cd $GRAFT_REPO_ROOT
b() { timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for v in "$@"; do
if [ "$v" = default ]; then b default; else DRS_LIB=$PWD/variants/libdrs_$v.so b $v; fi
done; done
