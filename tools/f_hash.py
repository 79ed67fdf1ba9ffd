"""sha1 of the descriptors of the bench pair (capacity mode): the quick bit-identity check between two builds or
two settings of a kernel switch.  usage: [IMF_...=..] python tools/f_hash.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
import bench
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
pts, imgs = bench.load_pair(1.7)
wl = bench.Workload(model, dev, pts, imgs, 0.025)
with torch.no_grad():
    wl.prepare_graph()
    wl.runner.use_graph = False
    res = wl.graph_step()
    torch.cuda.synchronize()
    F = res.F.detach().cpu().numpy()
    flags = res.flags
print("rows", F.shape[0], "sha1", hashlib.sha1(np.ascontiguousarray(F).tobytes()).hexdigest(), "max|F|", float(np.abs(F).max()), "nan", int(np.isnan(F).sum()), "flags", flags)
