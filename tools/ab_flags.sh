#!/bin/bash
# per-op times for DRS_DEBUG_FLAGS values inside ONE gpurun call: tools/ab_flags.sh "0 256 ..." op-substring ...
VALS=$1; shift
for v in $VALS; do
  DRS_DEBUG_FLAGS=$v DRS_BENCH_OPS=gpurun_out/ab_ops.txt python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "== flags $v $(python -c "import json; print(json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])['value'])")"
  for op in "$@"; do grep "$op" gpurun_out/ab_ops.txt | cut -c1-58; done
done
