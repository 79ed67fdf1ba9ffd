"""Host-side phase timing of one step (no profiler): where does the host thread spend its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from imfnet_amd.model import load_model
from imfnet_amd.model import plan as planmod
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
# instrument plan.run phases
orig_run = planmod.FusedPlan.run
def timed_fuse_wrapper(fuse):
    def f(x):
        t = time.perf_counter(); r = fuse(x); tick("6b  fuse (torch attention) queue", t); return r
    return f
def run(self, x, fuse, after_fuse=None):
    t = time.perf_counter(); r = orig_run(self, x, timed_fuse_wrapper(fuse), after_fuse); tick("6a  plan.run total", t); return r
planmod.FusedPlan.run = run
with torch.no_grad():
    for it in range(30):
        t = time.perf_counter()
        st = {}
        def hook():
            tick("1 geometry queue (pyramid_build)", t)
            t1 = time.perf_counter()
            model.start_image_branch(img_d, inputs_ready=True)
            tick("2 image graph launch", t1)
            st["t"] = time.perf_counter()
        levels = ops.pyramid_from_points(xyz_d, voxel, 4, 0, inputs_ready=True, before_sync=hook)
        tick("3 event wait + level objects", st["t"]); t = time.perf_counter()
        cm = ME.CoordinateManager.from_levels(levels)
        f = torch.ones((levels[0].n, 1), dtype=torch.float32, device=dev)
        s = ME.SparseTensor(f, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=cm); s._all_ones = True
        tick("4 tensor objects", t); t = time.perf_counter()
        F = model(s, img_d).F
        tick("5 forward queue (total)", t)
    torch.cuda.synchronize()
    # whole-step host enqueue time vs wall time per step
    t = time.perf_counter(); n = 30
    for it in range(n):
        levels = ops.pyramid_from_points(xyz_d, voxel, 4, 0, inputs_ready=True, before_sync=lambda: model.start_image_branch(img_d, inputs_ready=True))
        cm = ME.CoordinateManager.from_levels(levels)
        f = torch.ones((levels[0].n, 1), dtype=torch.float32, device=dev)
        s = ME.SparseTensor(f, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=cm); s._all_ones = True
        F = model(s, img_d).F
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"host enqueue {t_host / n * 1e6:.1f} us/step, wall {t_all / n * 1e6:.1f} us/step")
for k in sorted(T):
    print(f"{k:36s} median {np.median(T[k][8:])*1e6:8.1f} us")
