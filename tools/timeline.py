#!/usr/bin/env python3
"""Per-queue busy time and the timeline of the last full step from a rocprofv3 kernel-trace DB."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_init_tables" in r[0]]
i0, i1 = idx[-3], idx[-2]
t0 = rows[i0][1]
print("step span us:", (rows[i1][1] - t0) / 1e3)
qs = {}
for r in rows[i0:i1]:
    qs.setdefault(r[3], []).append(r)
for q, rs in qs.items():
    print("queue", q, "n=", len(rs), "busy us=%.1f" % (sum(r[2] - r[1] for r in rs) / 1e3), rs[0][0][:40])
# union busy of all queues
ev = sorted((r[1], r[2]) for r in rows[i0:i1])
busy, cur_s, cur_e = 0, None, None
for s, e in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy us: %.1f" % (busy / 1e3))
if len(sys.argv) > 2:
    for r in rows[i0:i1 + 20]:
        print("%8.1f %7.1f q%d %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].split("(")[0][-48:]))
