#!/usr/bin/env python3
"""Busy time and span per HIP stream / queue in a rocprofv3 rocpd result.  Usage: rocpd_streams.py results.db"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
print([x for x in cols if 'stream' in x or 'queue' in x])
key = 'stream_id' if 'stream_id' in cols else 'queue_id'
for row in c.execute(f"select {key}, count(*), sum(end-start)/1e6, min(start), max(end) from {kd} group by {key}"):
    print(row[0], row[1], "%.1f ms busy" % row[2], "span %.1f ms" % ((row[4]-row[3])/1e6))
