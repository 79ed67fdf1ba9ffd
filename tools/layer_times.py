"""Per-layer sparse-conv kernel times of one S50k fragment (HIP events around each launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd.extract import sparse_tensor_from_points
from imfnet_amd.model import load_model
from bench import load_workload, load_pair
from imfnet_amd.extract import start_geometry
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
xyz_d, img_d = torch.as_tensor(xyz).to(dev), torch.as_tensor(img).to(dev)
starts = None
if os.environ.get("BATCH") == "2":
    pts, imgs = load_pair(1.7)
    xyz_d, img_d, starts = torch.as_tensor(np.concatenate(pts, 0)).to(dev), torch.as_tensor(imgs).to(dev), [0, len(pts[0])]
acc = {}
with torch.no_grad():
    for it in range(8):
        ops.TRACE = [] if it >= 3 else None
        st, _ = sparse_tensor_from_points(None, voxel, dev, geometry=start_geometry(xyz_d, voxel, dev, item_starts=starts))
        model(st, img_d).F
        torch.cuda.synchronize()
        if ops.TRACE:
            for i, r in enumerate(ops.TRACE):
                key = (i, r.get("name", "?"), r["kernel"], r["kvol"], r["cin"], r["cout"], r["rb"].n_slots, r["split"])
                acc.setdefault(key, []).append(r["ev"].elapsed_ms() * 1e3)
        ops.TRACE = None
tot = 0.0
for key, v in sorted(acc.items()):
    m = float(np.median(v)); tot += m
    print("%2d %-18s %-18s k=%3d %3d->%3d slots=%6d split=%d  %7.1f us" % (*key, m))
print("sum of conv main kernels: %.1f us" % tot)
