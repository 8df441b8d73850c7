#!/usr/bin/env python3
"""Clock the chip holds under a full split-bf16 MFMA load (the matrix pipe is power-limited).
Builds tools/micro/cons_loop.hip (the consumer side of the SP 3x3 kernel: LDS fragment reads + MFMAs, no global memory),
runs it for a few seconds of back-to-back launches and records ticks per step, the MFMA-rate fraction and the in-kernel
clock (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md 'DVFS give-back' item 6) for constant and for
random operands, on 8 and on 256 CUs.
    python tools/mfma_clock.py profiles/r03_mfma_clock.json
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tools", "micro", "cons_loop.hip")
exe = os.path.join(os.environ.get("TMPDIR", "/tmp"), "drs_cons_loop")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-o", exe, src], check=True)
# 4096 steps of ~4.5 us per launch, 120 launches per configuration: ~2 s of load each
out = subprocess.run([exe, "4096", "120"], check=True, capture_output=True, text=True).stdout
rows, data = [], None
for line in out.splitlines():
    m = re.match(r"data (\d)", line)
    if m:
        data = int(m.group(1))
        continue
    m = re.match(r"(.+?)\s+blocks\s+(\d+):\s+(\d+) ticks/step \((\d+) % of the MFMA rate\), ([\d.]+) us, ([\d.]+) GHz \(event\), "
                 r"([\d.]+) GHz", line)
    if m:
        us_step = float(m.group(5)) / 4096
        clk = float(m.group(7))
        # (the in-kernel tick count is wave 0's: with two waves per SIMD the older wave finishes early, so the rate is taken
        # from the launch's wall time at the in-kernel clock instead)
        rows.append({"variant": m.group(1).strip(), "operands": "random mantissas" if data else "near-constant", "cus": int(m.group(2)),
                     "us_per_step": round(us_step, 4), "clock_ghz_in_kernel": clk,
                     "ticks_per_step_wall": round(us_step * clk * 1e3), "mfma_rate": round(6912 / (us_step * clk * 1e3), 3)})
# (V2 is the fp16 + fp6 study of DESIGN.md section 9: its step has fewer MFMAs, the 6912-tick reference does not apply)
rows = [r for r in rows if not r["variant"].startswith("V2")]
full = [r for r in rows if r["cus"] == 256 and r["operands"].startswith("random")]
res = {"source": "tools/micro/cons_loop.hip via tools/mfma_clock.py", "steps_per_launch": 4096, "launches": 120,
       "rows": rows,
       "power_limited_ceiling": None}
if full:
    clk = min(r["clock_ghz_in_kernel"] for r in full)
    rate = max(r["mfma_rate"] for r in full)
    res["power_limited_ceiling"] = {
        "clock_ghz": clk, "mfma_rate": rate,
        "mfma_work_pflops": round(2.5 * rate * clk / 2.4, 3),
        "algorithmic_pflops_at_3_mfma_per_product": round(2.5 * rate * clk / 2.4 / 3, 3),
        "note": "2.5 PFLOP/s dense bf16 x issue-rate fraction x (clock under load / 2.4 GHz)"}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_mfma_clock.json")
with open(path, "w") as f:
    json.dump(res, f, indent=1)
print(out)
print(json.dumps(res["power_limited_ceiling"]))
