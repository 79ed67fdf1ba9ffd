#!/usr/bin/env python3
"""Average per launch of every counter in a rocprofv3 --pmc database, for the kernels whose name matches a regex.
usage: pmc_dump.py <results.db> <kernel regex>"""
import collections
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value) from pmc_events "
                   "group by name, counter_name, dispatch_id").fetchall()
agg = collections.defaultdict(list)
pat = re.compile(sys.argv[2])
for name, counter, _, v in rows:
    short = name.split("(")[0].replace("void ", "")
    if pat.search(short):
        agg[(short, counter)].append(v)
for (k, c), v in sorted(agg.items()):
    print(f"{k:40s} {c:36s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
