#!/bin/bash
# per-op times of library builds inside ONE gpurun call: tools/ab_ops.sh "libA.so libB.so ..." op-substring ...
LIBS=$1; shift
for lib in $LIBS; do
  DRS_LIB=$PWD/$lib DRS_BENCH_OPS=gpurun_out/ab_ops.txt python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "== $lib $(python -c "import json; print(json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])['value'])")"
  for op in "$@"; do grep "$op" gpurun_out/ab_ops.txt | cut -c1-58; done
done
