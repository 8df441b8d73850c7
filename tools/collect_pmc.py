#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (one directory per pass, --output-format csv) into profiles/<name>.json:
per kernel: launches, HBM read bytes (FETCH_SIZE KB x 1024 x 2: gfx950 reports half of wide reads, see
/opt/skills/guides/MI355X_MICROARCH.md), HBM write bytes (WRITE_SIZE KB x 1024), SQ_* ratios to SQ_WAVE_CYCLES.
Usage: collect_pmc.py OUT.json STEPS DIR [DIR ...]   (STEPS = UNet forwards in the profiled run)"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--ops=")]
ops_file = next((a[6:] for a in sys.argv[1:] if a.startswith("--ops=")), None)
out, steps, dirs = args[0], int(args[1]), args[2:]
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))[:160]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row.get("Dispatch_Id"))
            if row["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVE_CYCLES") and key not in seen:
                seen.add(key)
                launches[(k, row["Counter_Name"])] += 1
# per-dispatch HBM bytes of the LAST forward, in launch order, labelled with the plan's op names (bench.py writes them
# with DRS_BENCH_OPS): the kernels of the library are launched one per op, in schedule order
per_op = None
if ops_file and os.path.exists(ops_file):
    names = [ln.split()[0] for ln in open(ops_file) if ln.strip() and not ln.startswith("lr_branch")]
    disp = defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "kernel": ""})
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    e = disp[int(row["Dispatch_Id"])]
                    e[row["Counter_Name"]] += float(row["Counter_Value"])
                    e["kernel"] = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))[:120]
    mine = [v for _, v in sorted(disp.items()) if re.search(r"tapconv|stem_kernel|time_mlp|attn_gate|conv3x3_direct|conv_s2_sp|convt_sp|upfuse_sp_kernel|upfuse_edges", v["kernel"])]
    if len(mine) >= len(names):
        last = mine[-len(names):]  # the last forward re-uses the cached conditioning branch: exactly the listed ops
        per_op = [{"op": n, "kernel": v["kernel"], "hbm_read_bytes": 2 * v["FETCH_SIZE"] * 1024,
                   "hbm_write_bytes": v["WRITE_SIZE"] * 1024} for n, v in zip(names, last)]
res = {}
for k, c in acc.items():
    n = max(launches.get((k, "FETCH_SIZE"), 0), launches.get((k, "SQ_WAVE_CYCLES"), 0), 1)
    e = {"launches_per_forward": n / steps}
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes_per_forward"] = 2 * c["FETCH_SIZE"] * 1024 / steps
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes_per_forward"] = c["WRITE_SIZE"] * 1024 / steps
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / n
    if c.get("SQ_WAVE_CYCLES"):
        e["sq_ratios_to_wave_cycles"] = {m: round(v / c["SQ_WAVE_CYCLES"], 4) for m, v in sorted(c.items()) if m.startswith("SQ_")}
    res[k] = e
json.dump({"command": "rocprofv3 --pmc <COUNTERS> --output-format csv -d DIR -- python3 tools/profile_forward.py --steps %d "
                      "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*), MI355X, bf16x3, BASELINE configs[1]" % steps,
           "note": "FETCH_SIZE/WRITE_SIZE are KB; hbm_read = 2 * FETCH_SIZE * 1024 (gfx950 correction of MI355X_MICROARCH.md); "
                   "values per UNet forward unless named per_launch",
           "per_op_last_forward": per_op,
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_read_bytes_per_forward", 0)))},
          open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
