#!/bin/bash
# The three rocprofv3 counter passes of one configs[1] chain (FETCH_SIZE | WRITE_SIZE | SQ_*: separate runs, as
# MI355X_MICROARCH.md prescribes) + the fold into <out>/pmc_traffic.json.  Usage (repo root): tools/collect_pmc_passes.sh OUT_DIR
set -o pipefail
O=${1:-$PWD/gpurun_out/prof}
mkdir -p $O && export TMPDIR=/tmp
rm -rf /tmp/f /tmp/w /tmp/sq
# (3 chain steps + the logged forward = 4 forwards per pass; every pass writes the same launch log)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/f -- python3 tools/profile_forward.py --steps 3 --launch-log $O/launch_log.txt > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/w -- python3 tools/profile_forward.py --steps 3 --launch-log $O/launch_log_w.txt > $O/w.log 2>&1 || exit 1
cmp $O/launch_log.txt $O/launch_log_w.txt || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/sq -- python3 tools/profile_forward.py --steps 3 --launch-log $O/launch_log_sq.txt > $O/sq.log 2>&1 || exit 1
python3 tools/collect_pmc.py $O/pmc_traffic.json 4 /tmp/f /tmp/w /tmp/sq --ops=$O/launch_log.txt || exit 1
