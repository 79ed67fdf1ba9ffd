"""Step time of the pair workload: hipGraph replay vs eager capacity mode vs exact mode vs two graphs in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
import bench as B
dev = torch.device("cuda:0")
model, sd = B.build_model(dev)
batch = int(os.environ.get("BATCH", "2"))
pts, imgs = B.load_pair(1.7) if batch == 2 else ([B.load_workload(1.7, 0.025)[0]], B.load_workload(1.7, 0.025)[1])
sync = torch.cuda.synchronize
only = os.environ.get("ONLY")
with torch.no_grad():
    wl = B.Workload(model, dev, pts, imgs, 0.025)
    wl.prepare_graph()
    r = wl.runner
    for _ in range(3): wl.graph_step()
    sync()
    for rnd in range(3 if only is None else 1):
      if only in (None, "graph"):
        r.use_graph = True
        for _ in range(3): wl.graph_step()
        print("graph replay      %.4f ms/step" % (B.timed(wl.graph_step, 50, sync) * 1e3))
      if only in (None, "eager"):
        r.use_graph = False
        for _ in range(3): wl.graph_step()
        print("eager capacity    %.4f ms/step" % (B.timed(wl.graph_step, 50, sync) * 1e3))
        r.use_graph = True
      if only in (None, "exact"):
        for _ in range(3): wl.exact_step()
        print("exact             %.4f ms/step" % (B.timed(wl.exact_step, 50, sync) * 1e3))
    if only in (None, "two"):
        # two buckets (buffer sets) alternating on two streams: consecutive fragments overlap
        import copy
        key = wl.bucket.key
        from imfnet_amd.model.graph import _Bucket
        s2 = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s2):
            b2 = _Bucket(r, key, dev)
        n = r.stage(b2, wl.xyz, wl.starts, wl.img, s2)
        lanes = [(wl.bucket, wl.stream), (b2, s2)]
        cnt = [0]
        def two():
            b, s = lanes[cnt[0] % 2]; cnt[0] += 1
            return r.launch(b, n, len(wl.starts), s)
        for _ in range(4): two()
        sync()
        print("two graphs in flight %.4f ms/step" % (B.timed(two, 30, sync) * 1e3))
        r.use_graph = False
        for _ in range(4): two()
        sync()
        print("two eager capacity fragments in flight %.4f ms/step" % (B.timed(two, 30, sync) * 1e3))
        with torch.cuda.stream(torch.cuda.Stream(device=dev)) as _:
            s3 = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s3):
            b3 = _Bucket(r, key, dev)
        r.stage(b3, wl.xyz, wl.starts, wl.img, s3)
        lanes.append((b3, s3))
        def three():
            b, s = lanes[cnt[0] % 3]; cnt[0] += 1
            return r.launch(b, n, len(wl.starts), s)
        for _ in range(6): three()
        sync()
        print("three eager capacity fragments in flight %.4f ms/step" % (B.timed(three, 30, sync) * 1e3))
        r.use_graph = True
