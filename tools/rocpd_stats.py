#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace -d DIR -o NAME` without
--output-format csv writes NAME_results.db).  Usage: rocpd_stats.py results.db [top_n] [--csv]"""
import re
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
rows = c.execute(f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                 f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
sep = "," if "--csv" in sys.argv else "  "
print(sep.join(["calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "kernel"]))
for name, n, tot, mn, mx in rows[:top]:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:150]
    print(sep.join([str(n), f"{tot/1e6:.3f}", f"{tot/n/1e3:.1f}", f"{mn/1e3:.1f}", f"{mx/1e3:.1f}", f"{100*tot/total:.1f}",
                    '"' + short + '"']))
print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
