#!/usr/bin/env python3
"""Per-kernel totals of ONE stream of a rocprofv3 rocpd result.  Usage: rocpd_stream_stats.py results.db STREAM_ID [top_n]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sid = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
rows = c.execute(f"select s.{name_col}, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
                 f"where d.stream_id = ? group by s.{name_col} order by 3 desc", (sid,)).fetchall()
total = sum(r[2] for r in rows)
for name, n, tot in rows[:top]:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:100]
    print(f"{n:6d} {tot / 1e6:9.3f} ms {tot / n / 1e3:8.1f} us {100 * tot / total:5.1f}%  {short}")
print(f"# stream {sid}: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
