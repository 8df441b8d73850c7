#!/usr/bin/env python3
"""Timeline of the LAST training step in a rocprofv3 rocpd result: every dispatch with its stream, start offset and
duration, then per-stream busy time, the time both streams are busy, and the main stream's idle gaps.
A step is delimited by the `adam_multi_kernel` launches (one per step).
Usage: rocpd_train_timeline.py results.db [out.txt]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
dcols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
key = "stream_id" if "stream_id" in dcols else "queue_id"
rows = c.execute(f"select d.start, d.end, d.{key}, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id "
                 f"order by d.start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_multi" in r[3]]
if len(adam) < 2:
    sys.exit("fewer than two adam_multi launches: not a training trace")
step = rows[adam[-2] + 1: adam[-1] + 1]
t0 = step[0][0]
streams = {}
for st, en, sid, name in step:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:80]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  s{sid}  {short}", file=out)
    streams.setdefault(sid, []).append((st, en, short))
print(f"# step span {(step[-1][1] - t0) / 1e6:.3f} ms, {len(step)} dispatches", file=out)
main = max(streams, key=lambda k: sum(e - s for s, e, _ in streams[k]))
for sid, ks_ in streams.items():
    busy = sum(e - s for s, e, _ in ks_)
    print(f"# stream {sid}{' (main)' if sid == main else ''}: {len(ks_)} launches, busy {busy / 1e6:.3f} ms", file=out)
# overlap of main with the others
others = sorted((s, e) for sid, ks_ in streams.items() if sid != main for s, e, _ in ks_)
ov = 0
for s, e, _ in streams[main]:
    for a, b in others:
        if b <= s:
            continue
        if a >= e:
            break
        ov += min(e, b) - max(s, a)
print(f"# main-stream time with another stream's kernel running: {ov / 1e6:.3f} ms", file=out)
gaps = []
m = streams[main]
for (s0, e0, n0), (s1, e1, n1) in zip(m, m[1:]):
    if s1 - e0 > 3000:
        gaps.append((s1 - e0, n0, n1))
gaps.sort(reverse=True)
print(f"# main-stream gaps > 3 us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.3f} ms", file=out)
for g, a, b in gaps[:25]:
    print(f"#   {g / 1e3:7.1f} us between {a[:50]} -> {b[:50]}", file=out)
