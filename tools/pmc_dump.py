#!/usr/bin/env python3
"""Average PMC counter values per kernel symbol from rocprofv3 rocpd databases.
usage: pmc_dump.py <substring of kernel name> <db> [<db> ...]"""
import collections
import glob
import sqlite3
import sys

sub = sys.argv[1]
for pat in sys.argv[2:]:
    for db in sorted(glob.glob(pat, recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value) from pmc_events "
                               "group by name, counter_name, dispatch_id").fetchall()
        except sqlite3.Error as e:
            print(db, "->", e)
            continue
        agg = collections.defaultdict(list)
        for name, cn, _, v in rows:
            if sub in name:
                agg[(name.split("(")[0].replace("void ", ""), cn)].append(v)
        for (k, cn), v in sorted(agg.items()):
            print(f"{k:40s} {cn:34s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
