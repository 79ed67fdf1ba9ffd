"""tools/kernel_timeline.py <kernel_trace.csv> [n_tail_dispatches]: the last dispatches of a rocprofv3 --kernel-trace CSV in start
order -- queue, start offset, duration, and the gap to the previous dispatch of the SAME queue -- to see what sits between two
convolutions of a step (rocprofv3 serialises nothing here: streams overlap as in the product)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
queues = {}
for r in rows:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("imf::", "").replace("(anonymous namespace)::", "")
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("q%d %9.1f us  dur %7.1f  gap %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    last_end[q] = e
