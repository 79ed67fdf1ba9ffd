"""When do the side stream (pyramid levels 1-3, rulebooks) and the image branch finish inside a step of the bench workload?
Elapsed times between the executor's own events: [7] level-0 pyramid done (main), [8] end of the side stream's chain,
[10] end of the image branch, and the end of the step.  usage: python tools/branch_times.py"""
import os, sys
os.environ["IMFNET_DIAG_EVENTS"] = "1"
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
    for _ in range(50): wl.graph_step()
    torch.cuda.synchronize()
    rows = []
    for _ in range(41):
        e0, e1 = L.imf_event_create(), L.imf_event_create()
        # events need timing enabled: the executor's are created by imf_event_create (default flags: timing on)
        with torch.cuda.stream(wl.stream):
            pass
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for _ in range(6):      # the host runs ahead of the device, as in the bench's back-to-back steps
            wl.graph_step()
        hip.hipEventRecord(ctypes.c_void_p(e0), ctypes.c_void_p(wl.stream.cuda_stream))
        wl.graph_step()
        hip.hipEventRecord(ctypes.c_void_p(e1), ctypes.c_void_p(wl.stream.cuda_stream))
        torch.cuda.synchronize()
        ev = wl.bucket.events
        ms = lambda a, b: L.imf_event_elapsed_ms(a, b) * 1e3
        # the side stream's marks: with conv1 + the level-0 map as one launch on the main stream (the default) there are
        # three (stride 2 / 4 / 8), with IMF_FIRST_AND_MAP=0 a fourth in front (the level-0 map)
        fam = os.environ.get("IMF_FIRST_AND_MAP", "1") != "0"
        marks = (float("nan"), ms(e0, ev[0]), ms(e0, ev[1]), ms(e0, ev[2])) if fam else tuple(ms(e0, ev[i]) for i in range(4))
        rows.append((ms(e0, ev[7]), ms(e0, ev[10]), ms(e0, ev[8]), ms(e0, ev[11]), ms(e0, ev[12]), ms(e0, e1)) + marks)
r = np.nanmedian(np.array(rows), axis=0) if not np.isnan(np.array(rows)[:, 6]).all() else np.concatenate([np.median(np.array(rows)[:, :6], axis=0), [float('nan')], np.median(np.array(rows)[:, 7:], axis=0)])
print("medians, us from the step's start: level-0 pyramid done %.0f | image branch done %.0f | side stream done (incl. its join "
      "with the image branch) %.0f | main stream reaches the join %.0f | fusion done %.0f | step done %.0f" % tuple(r[:6]))
print("side stream's marks: level-0 3x3x3 map ready %.0f | stride-2 level + its maps %.0f | stride-4 %.0f | stride-8 %.0f" % tuple(r[6:]))
