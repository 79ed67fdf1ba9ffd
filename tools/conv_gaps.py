"""In-situ timeline of the main stream's convolutions in one step of the bench workload (all three streams running): start
of every convolution relative to the step's first event, its duration, and the GAP before it (kernel boundary + whatever
the main stream waited for: a rulebook event of the side stream, the image branch).  Uses the HIP events the library
records around each convolution (roofline trace).  usage: python tools/conv_gaps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
import bench
from imfnet_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
pts, imgs = bench.load_pair(1.7)
wl = bench.Workload(model, dev, pts, imgs, 0.025)
with torch.no_grad():
    wl.prepare_graph()
    wl.runner.use_graph = False
    for _ in range(30): wl.graph_step()
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tl = []
        with torch.cuda.stream(wl.stream):
            e0.record(wl.stream)
        wl.graph_step(tl)
        with torch.cuda.stream(wl.stream):
            e1.record(wl.stream)
        torch.cuda.synchronize()
        runs.append((e0.elapsed_time(e1) * 1e3, tl))
total, tl = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
ms = lambda a, b: L.imf_event_elapsed_ms(a, b) * 1e3
first = tl[0]["ev"].begin
print("step (with event records) %.1f us; first convolution starts at ? (pyramid + conv1 before it)" % total)
prev_end = None
print("%-18s %-18s %9s %9s %9s" % ("conv", "kernel", "start", "dur", "gap"))
gsum = 0.0
for rec in tl:
    ev = rec["ev"]
    start, dur = ms(first, ev.begin), ms(ev.begin, ev.end)
    gap = ms(prev_end, ev.begin) if prev_end is not None else 0.0
    gsum += gap
    print("%-18s %-18s %9.1f %9.1f %9.1f" % (rec["name"], rec["kernel"], start, dur, gap))
    prev_end = ev.end
print("sum of gaps %.1f us, sum of convolutions %.1f us" % (gsum, sum(ms(r["ev"].begin, r["ev"].end) for r in tl)))
