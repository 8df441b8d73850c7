#!/bin/bash
# Round-end evidence, one GPU call: kernel trace of the default bench + the three PMC passes (run from the repo root).
# Products land in gpurun_out/prof/; copy the summaries into profiles/ (see profiles/README.md).
set -o pipefail
R=$PWD
O=$R/gpurun_out/prof
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/f /tmp/w /tmp/sq
cp $R/bench.py $R/tools/profile_forward.py /tmp/ 2>/dev/null
cd $R
DRS_BENCH_OPS=$O/bench_ops.txt rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.log || exit 1
python3 tools/rocpd_sequence.py $(find /tmp/kt -name "b_results.db" | head -1) 40 > $O/step_kernel_sequence.txt || exit 1
python3 tools/rocpd_stats.py $(find /tmp/kt -name "b_results.db" | head -1) 40 --csv > $O/bench_kernel_stats.csv || exit 1
bash tools/collect_pmc_passes.sh $O || exit 1
# configs[2] per-rank train step (exact-fp32 MFMA): kernel stats + the bench line
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o t -- python3 bench.py --workload train --steps 8 --warmup 2 > $O/train_under_rocprof.json 2> $O/kt2.log || exit 1
python3 tools/rocpd_stats.py $(find /tmp/kt2 -name "t_results.db" | head -1) 40 --csv > $O/train_kernel_stats.csv || exit 1
python3 bench.py --workload train --steps 10 --warmup 2 > $O/bench_train.json 2>> $O/kt2.log
python3 tools/latency_regime.py --steps 50 --json $O/latency_regime.json > $O/latency_regime.txt 2>&1 || exit 1
python3 tools/per_op_table.py > $O/per_op_table.txt 2>&1 || exit 1
python3 tools/diag_grads.py > $O/diag_grads.txt 2>&1 || true
python3 tools/mfma_clock.py $O/mfma_clock.json > $O/mfma_clock.log 2>&1 || true
python3 bench.py > $O/bench_final.json 2> $O/bench_final.log
tail -c 600 $O/bench_final.json
