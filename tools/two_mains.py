"""Two forwards in flight: the bench's device-resident pair step alternating between TWO capacity buckets on TWO main
streams (side / image streams shared) against the single-bucket loop, same box.   usage: python tools/two_mains.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from bench import load_pair, build_model, Workload
from imfnet_amd import _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
model, sd = build_model(dev)
with torch.no_grad():
    wl = Workload(model, dev, pts, imgs, 0.025)
    F0 = wl.prepare_graph(replicate=True)
    r = wl.runner; r.use_graph = False
    L = _lib.lib()
    main1 = wl.stream
    raw2 = L.imf_stream_create()
    main2 = torch.cuda.ExternalStream(raw2, device=dev)
    b1 = wl.bucket
    b2 = r.bucket(b1.key, dev, main2, lane=1)
    r.stage(b2, wl.xyz, wl.starts, wl.img, main2)
    torch.cuda.synchronize()
    turn = [0]
    def step1():
        b1.io.xyz = wl.replicas[turn[0] % len(wl.replicas)].data_ptr(); turn[0] += 1
        return r.launch(b1, wl.n_points, len(wl.starts), main1)
    def step2():
        k = turn[0]; turn[0] += 1
        b, st = (b1, main1) if k % 2 == 0 else (b2, main2)
        b.io.xyz = wl.replicas[k % len(wl.replicas)].data_ptr()
        return r.launch(b, wl.n_points, len(wl.starts), st)
    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for fn in (step1, step2):
        for _ in range(30): fn()
    m = int(b1.meta[0].item())
    for rep in range(3):
        a = timed(step1, steps); b = timed(step2, steps)
        print("one bucket / one main: %.4f ms per pair   two buckets / two mains: %.4f ms per pair  (%.1f -> %.1f M desc/s)" %
              (a, b, m / a / 1e3, m / b / 1e3))
    torch.cuda.synchronize()
    Fa, Fb = b1.out[:m].clone(), b2.out[:m].clone()
    print("descriptors of the two buckets equal each other:", torch.equal(Fa, Fb), " equal the exact path:", torch.equal(Fa, F0[:m]) if F0.shape[0] >= m else None,
          "flags", int(b1.meta[1].item()), int(b2.meta[1].item()))
