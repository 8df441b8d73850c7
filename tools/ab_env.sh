#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (the pool's boxes differ by a few per cent):
#   tools/ab_env.sh VAR "v1 v2 ..." [op-name-substring]     -> steps/s (+ the per-op line) per value
VAR=$1; VALS=$2; OP=$3
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v DRS_BENCH_OPS=gpurun_out/ab_ops.txt python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$VAR=$v', d['value'])
"
  [ -n "$OP" ] && grep "$OP" gpurun_out/ab_ops.txt | cut -c1-60
done; done
