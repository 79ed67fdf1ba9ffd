"""Host-side phase timing of one step (no profiler): where does the host thread spend its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from imfnet_amd.model import load_model
from bench import load_workload

xyz, img, voxel = load_workload(1.7, 0.025)
dev = torch.device("cuda:0")
sd = O.seeded_state_dict(0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
model.load_state_dict(sd); model = model.eval().to(dev)
xyz_d = torch.as_tensor(xyz).to(dev); img_d = torch.as_tensor(img).to(dev)
T = {}
def tick(name, t0):
    T.setdefault(name, []).append(time.perf_counter() - t0)
with torch.no_grad():
    for it in range(25):
        t = time.perf_counter()
        meta = ops.new_meta(4, dev)
        lv = ops.voxelize(xyz_d, voxel, 0, meta=meta[0])
        cm = ME.CoordinateManager(lv, meta=meta)
        tick("1 voxelize queue", t); t = time.perf_counter()
        # queue downsamples (copy of build_pyramid without sync)
        cm.build_pyramid(8, before_sync=lambda: (tick("2 downsample queue", t), T.setdefault("_t", []).append(time.perf_counter()),
                                                  model.start_image_branch(img_d),
                                                  tick("3 image graph launch", T["_t"][-1]), T["_t"].append(time.perf_counter())))
        tick("4 sync wait", T["_t"][-1]); t = time.perf_counter()
        f = torch.ones((lv.n, 1), dtype=torch.float32, device=dev)
        st = ME.SparseTensor(f, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=cm); st._all_ones = True
        tick("5 tensor", t); t = time.perf_counter()
        F = model(st, img_d).F
        tick("6 forward queue", t)
    torch.cuda.synchronize()
for k in sorted(T):
    if k != "_t": print(f"{k:28s} median {np.median(T[k][5:])*1e6:8.1f} us")
