#!/usr/bin/env python3
"""Timeline of the streaming pipeline from a rocprofv3 kernel-trace DB: where the upload / download kernels of a job sit
relative to the neighbouring forwards, and which hardware queue every stream landed on.
usage: python tools/stream_timeline.py <results.db> [out.txt]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
def short(n): return n.replace("(anonymous namespace)::", "").replace("imf::", "").replace("void ", "").split("(")[0][-44:]
heads = [i for i, r in enumerate(rows) if "k_pointwise_head" in r[0]]
lo = rows[heads[-8]][1] if len(heads) >= 8 else rows[0][1]
hi = rows[heads[-3]][2] if len(heads) >= 3 else rows[-1][2]
sel = [r for r in rows if lo <= r[1] <= hi]
queues = {}
for r in sel:
    queues.setdefault(r[3], {}).setdefault(short(r[0]), 0)
    queues[r[3]][short(r[0])] += 1
print("hardware queues in the window (queue id: kernels):", file=out)
for q, ks in sorted(queues.items()):
    top = sorted(ks.items(), key=lambda kv: -kv[1])[:6]
    print("  q%d: %s" % (q, ", ".join("%s x%d" % kv for kv in top)), file=out)
marks = ("k_copy_in", "k_copy_out", "k_init_tables", "k_pointwise_head", "k_gather_points", "k_insert_points", "k_fusion_attn")
print("\n   t_us     dur_us  queue  kernel   (window: five forwards)", file=out)
for r in sel:
    if any(m in r[0] for m in marks):
        print("%9.1f %9.1f   q%-3d  %s" % ((r[1] - lo) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])), file=out)
# overlap of the copy kernels with compute kernels of other queues
def overlap(a, b): return max(0, min(a[2], b[2]) - max(a[1], b[1]))
comp = [r for r in sel if "k_copy" not in r[0]]
for name in ("k_copy_in", "k_copy_out"):
    cs = [r for r in sel if name in r[0]]
    tot = sum(r[2] - r[1] for r in cs)
    # union of compute intervals
    ev = sorted((r[1], r[2]) for r in comp)
    merged = []
    for s, e in ev:
        if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
        else: merged.append([s, e])
    ov = sum(overlap((0, c[1], c[2]), (0, m[0], m[1])) for c in cs for m in merged)
    if cs:
        print("\n%s: %d launches, avg %.1f us, %.0f %% of their time under compute kernels" % (name, len(cs), tot / len(cs) / 1e3, 100.0 * ov / max(1, tot)), file=out)
span = (rows[heads[-3]][2] - rows[heads[-8]][2]) / 5e3 if len(heads) >= 8 else 0
print("\nforward-to-forward period over the window: %.1f us per job" % span, file=out)
