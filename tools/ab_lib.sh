#!/bin/bash
# A/B of library builds inside ONE gpurun call: tools/ab_lib.sh "libA.so libB.so" [op-name-substring ...]
LIBS=$1; shift
for rep in 1 2; do for lib in $LIBS; do
  DRS_LIB=$PWD/$lib DRS_BENCH_OPS=gpurun_out/ab_ops.txt python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$lib', d['value'])
"
  for op in "$@"; do grep "$op" gpurun_out/ab_ops.txt | cut -c1-60; done
done; done
