"""Throughput of the bench workload (the S50k pair per step) with one lane vs two lanes: two FragmentRunners of the same
model (own capacity buckets, own side / image streams), steps issued alternately on two main streams, so that one step's
coarse levels (136-240 workgroups on 256 CUs) can run beside the other step's level-0 kernels.
usage: python tools/two_lanes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
import bench
from imfnet_amd.model.graph import FragmentRunner
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
pts, imgs = bench.load_pair(1.7)
with torch.no_grad():
    lanes = []
    for k in range(3):
        wl = bench.Workload(model, dev, pts, imgs, 0.025)
        F = wl.prepare_graph()
        if k > 0:                                          # a runner of its own: separate buckets and streams
            r = FragmentRunner(model)
            r.ratios, r.grid_words = wl.runner.ratios, wl.runner.grid_words
            key = r.caps_for(int(wl.xyz.shape[0]), len(wl.starts), int(wl.img.shape[2]), int(wl.img.shape[3]), wl.voxel, True)
            wl.stream = r.main_stream(dev)
            wl.runner, wl.bucket = r, r.bucket(key, dev, wl.stream)
            wl.n_points = r.stage(wl.bucket, wl.xyz, wl.starts, wl.img, wl.stream)
        wl.runner.use_graph = False
        lanes.append(wl)
    ref = lanes[0].graph_step().F.clone()
    for wl in lanes:
        assert torch.equal(wl.graph_step().F, ref)
    torch.cuda.synchronize()
    for n_l in (1, 2, 3, 1, 2):
        use = lanes[:n_l]
        for _ in range(30):
            for wl in use: wl.graph_step()
        torch.cuda.synchronize()
        reps = []
        for _ in range(7):
            t0 = time.perf_counter()
            for i in range(60):
                use[i % n_l].graph_step()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 60 * 1e3)
        print(f"{n_l} lane(s): median {np.median(reps):.4f} ms per step  (min {min(reps):.4f})", flush=True)
