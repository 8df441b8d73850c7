set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DRS_FL=1 FLC_OUT=/tmp/a.pt timeout -k 10 300 python tools/fl_check.py 2>&1 | grep -v amdgpu.ids
DRS_FL=0 FLC_OUT=/tmp/b.pt timeout -k 10 300 python tools/fl_check.py 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import torch
a=torch.load('/tmp/a.pt'); b=torch.load('/tmp/b.pt')
print("FL vs bf16x3: max-rel %.3e rel-L2 %.3e" % (float((a-b).abs().max()/b.abs().max()), float((a-b).norm()/b.norm())))
PY
DRS_FL=1 timeout -k 10 200 python tools/per_op_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/per_op_fl1.txt
DRS_FL=0 timeout -k 10 200 python tools/per_op_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/per_op_fl0.txt
paste gpurun_out/per_op_fl1.txt gpurun_out/per_op_fl0.txt | awk '{printf "%-34s %8s %8s   | %8s %8s\n",$1,$2,$4,$7,$9}'
DRS_FL=1 timeout -k 10 200 python bench.py --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FL=1', d['value'], d['ms_per_step'])"
DRS_FL=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FL=0', d['value'], d['ms_per_step'])"
