#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (one directory per pass, --output-format csv) into profiles/<name>.json:
per kernel: launches, HBM read bytes (FETCH_SIZE KB x 1024 x 2: gfx950 reports half of wide reads, see
/opt/skills/guides/MI355X_MICROARCH.md), HBM write bytes (WRITE_SIZE KB x 1024), SQ_* ratios to SQ_WAVE_CYCLES.
Usage: collect_pmc.py OUT.json STEPS DIR [DIR ...] [--ops=LAUNCH_LOG]   (STEPS = UNet forwards in the profiled run,
the logged one included; LAUNCH_LOG = the file tools/profile_forward.py --launch-log wrote in EVERY pass)"""
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
# Per-dispatch HBM bytes of the LAST forward, attributed through the plan's own launch log (tools/profile_forward.py
# --launch-log: one "op <TAB> kernel" line per launch, written by drs_unet_profile_launch): the logged forward is the last
# GPU work of the profiled process, so its launches are the process's last len(log) dispatches, in this order.  Nothing is
# matched by a hand-kept list of kernel names: every dispatch's kernel name must equal the logged one, and the per-op sums
# must add up to the per-kernel totals of the same dispatch range - otherwise the collection FAILS (round 3's file zipped
# op names onto the wrong dispatches after two kernels were added to the plan but not to a regex here).
def norm_kernel(k):
    k = k.replace("(anonymous namespace)::", "").replace(" [clone .kd]", "")
    k = re.sub(r"\.kd$", "", k.strip())
    k = re.sub(r"^void\s+", "", k)
    return re.sub(r"\s+", "", k.split("(")[0])


per_op = per_launch = None
if ops_file:
    log = [ln.rstrip("\n").split("\t") for ln in open(ops_file) if ln.strip()]
    disp = defaultdict(lambda: {"FETCH_SIZE": None, "WRITE_SIZE": None, "kernel": ""})
    per_pass_ids = []
    for d in dirs:
        ids = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                ids.add(int(row["Dispatch_Id"]))
                if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    e = disp[int(row["Dispatch_Id"])]
                    e[row["Counter_Name"]] = (e[row["Counter_Name"]] or 0.0) + float(row["Counter_Value"])
                    e["kernel"] = row["Kernel_Name"]
        per_pass_ids.append(ids)
    # every pass ran the same program: the dispatch ids must line up (same count per pass)
    counts = {len(i) for i in per_pass_ids if i}
    if len(counts) != 1:
        sys.exit(f"collect_pmc: the passes saw different numbers of dispatches {sorted(counts)}: not the same program")
    ordered = [v for _, v in sorted(disp.items())]
    if len(ordered) < len(log):
        sys.exit(f"collect_pmc: {len(ordered)} dispatches with HBM counters < {len(log)} logged launches")
    last = ordered[-len(log):]
    bad = [(i, op, kn, v["kernel"]) for i, ((op, kn), v) in enumerate(zip(log, last)) if norm_kernel(kn) != norm_kernel(v["kernel"])]
    if bad:
        for b in bad[:8]:
            print("collect_pmc: launch %d op %s: plan logged %r, profiler saw %r" % b, file=sys.stderr)
        sys.exit("collect_pmc: the last dispatches of the counter passes are not the logged forward: no attribution")
    if any(v["FETCH_SIZE"] is None or v["WRITE_SIZE"] is None for v in last):
        sys.exit("collect_pmc: a dispatch of the last forward lacks FETCH_SIZE or WRITE_SIZE (pass missing?)")
    per_launch = [{"op": op, "kernel": norm_kernel(v["kernel"]), "hbm_read_bytes": 2 * v["FETCH_SIZE"] * 1024,
                   "hbm_write_bytes": v["WRITE_SIZE"] * 1024} for (op, kn), v in zip(log, last)]
    per_op, order = {}, []
    for e in per_launch:
        if e["op"] not in per_op:
            per_op[e["op"]] = {"op": e["op"], "kernel": e["kernel"], "launches": 0, "hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0}
            order.append(e["op"])
        o = per_op[e["op"]]
        o["launches"] += 1
        if e["kernel"] not in o["kernel"].split(" + "):
            o["kernel"] += " + " + e["kernel"]
        o["hbm_read_bytes"] += e["hbm_read_bytes"]
        o["hbm_write_bytes"] += e["hbm_write_bytes"]
    per_op = [per_op[k] for k in order]
    # cross-check: per-op sums == per-kernel sums over the same dispatch range (within 1 %)
    by_k = defaultdict(float)
    for e in per_launch:
        by_k[e["kernel"]] += e["hbm_read_bytes"] + e["hbm_write_bytes"]
    tot_ops = sum(o["hbm_read_bytes"] + o["hbm_write_bytes"] for o in per_op)
    if abs(tot_ops - sum(by_k.values())) > 0.01 * max(tot_ops, 1.0):
        sys.exit("collect_pmc: per-op and per-kernel sums of the last forward differ")
    print("collect_pmc: last forward = %d launches, %.1f MB read + %.1f MB written; attribution verified against the plan's launch log"
          % (len(per_launch), sum(o["hbm_read_bytes"] for o in per_op) / 1e6, sum(o["hbm_write_bytes"] for o in per_op) / 1e6))
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
           "attribution": ("plan launch log (drs_unet_profile_launch), kernel name verified per dispatch" if per_op else None),
           "per_op_last_forward": per_op,
           "per_kernel_last_forward": ({k: v for k, v in sorted(by_k.items(), key=lambda kv: -kv[1])} if per_op else None),
           "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_read_bytes_per_forward", 0)))},
          open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
