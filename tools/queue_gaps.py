"""tools/queue_gaps.py <kernel_trace.csv> <t_begin_frac> <t_end_frac>: per HW queue of a rocprofv3 --kernel-trace CSV, inside the
window [begin, end] given as fractions of the trace's span (a steady-state stretch of untraced steps): busy time, idle time and
every idle gap above 2 us with the kernels on either side -- where a stream of the step is waiting."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
f0, f1 = float(sys.argv[2]), float(sys.argv[3])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
T0, T1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
a, b = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
short = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("imf::", "").replace("(anonymous namespace)::", "")[:44]
byq = collections.defaultdict(list)
for r in rows:
    if a <= int(r["Start_Timestamp"]) <= b:
        byq[r["Queue_Id"]].append(r)
n_first = sum(1 for r in rows if a <= int(r["Start_Timestamp"]) <= b and "k_conv_first" in r["Kernel_Name"])
print("window %.1f ms, %d forwards (k_conv_first launches)" % ((b - a) / 1e6, n_first))
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    gaps = collections.Counter(); gsum = collections.Counter()
    for p, n in zip(rs, rs[1:]):
        g = (int(n["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
        if g > 2.0:
            k = short(p) + "  ->  " + short(n)
            gaps[k] += 1; gsum[k] += g
    print("queue %s: %d dispatches, busy %.1f us / forward, idle %.1f us / forward" % (q, len(rs), busy / 1e3 / max(n_first, 1), (span - busy) / 1e3 / max(n_first, 1)))
    for k, s in sorted(gsum.items(), key=lambda kv: -kv[1])[:14]:
        print("    %7.1f us / forward  (%3d x %5.1f)  %s" % (s / max(n_first, 1), gaps[k], s / gaps[k], k))
