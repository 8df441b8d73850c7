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

out, steps, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
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
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_read_bytes_per_forward", 0)))},
          open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
