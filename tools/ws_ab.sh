#!/bin/bash
# A/B of the wave-specialised conv kernel on one box: parity subset, bench with DRS_WS=0/1, phase timeline (variants/libdrs_tl.so)
set -o pipefail
DRS_WS=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
for w in 0 1 0 1; do echo "WS $w"; DRS_WS=$w timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75; done
if [ -f variants/libdrs_tl.so ]; then
  DRS_LIB=$PWD/variants/libdrs_tl.so DRS_CONCURRENT=0 DRS_WS=1 timeout -k 10 200 python tools/per_op_table.py --iters 1 > gpurun_out/tl.txt 2>&1
fi
