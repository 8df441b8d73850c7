#!/usr/bin/env python3
"""Ordered kernel dispatches of the LAST forward in a rocprofv3 rocpd result: start offset, duration, grid, kernel.
Usage: rocpd_sequence.py results.db [n_last]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
dcols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
grid = "d.grid_size_x" if "grid_size_x" in dcols else "0"
wg = "d.workgroup_size_x" if "workgroup_size_x" in dcols else "0"
rows = c.execute(f"select d.start, d.end, {grid}, {wg}, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id "
                 f"order by d.start").fetchall()[-n_last:]
t0 = rows[0][0]
for st, en, g, w, name in rows:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:90]
    print(f"{(st - t0) / 1e3:9.1f} us  {(en - st) / 1e3:7.1f} us  grid {g:>8} x {w:<4} {short}")
