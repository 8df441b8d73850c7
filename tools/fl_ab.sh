set -e
cd $GRAFT_REPO_ROOT
for v in copy nocross nomain; do
echo "== $v"
DRS_LIB=$PWD/variants/libdrs_$v.so DRS_FL=1 timeout -k 10 200 python tools/per_op_table.py 2>&1 | grep -v amdgpu.ids | grep "bottle_neck.conv1\|ups.0.conv \|ups.2.conv \|conv_blocks.2.conv2\|total"
done
