"""Launch + dispatch floor of the conv kernel: every workgroup returns at once (all-zero tile masks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
g = torch.Generator().manual_seed(0)
for lvl, cin, cout in ((0, 64, 64), (1, 64, 64), (2, 128, 128), (3, 256, 256)):
    rb = cm.conv_rulebook(1 << lvl, 3, 1)
    f = torch.randn(levels[lvl].n, cin, generator=g).to(dev)
    w = ops.pack_weights((torch.randn(27, cin, cout, generator=g) * 0.05).to(dev), split16=True)
    res = {}
    for name in ("real", "empty"):
        if name == "empty":
            rb.tile_mask.zero_()
        ts = []
        for _ in range(8):
            ops.TRACE = []
            ops.spconv(f, w, cout, rb, variant=6, split_k=1)
            torch.cuda.synchronize()
            ts.append(ops.TRACE[0]["ev"].elapsed_ms() * 1e3); ops.TRACE = None
        res[name] = np.median(ts[2:])
    print(f"level {lvl} {cin}->{cout}: tiles {rb.n_slots // 64} x slabs {cout // 64}: real (split 1) {res['real']:.1f} us, all workgroups return at once {res['empty']:.1f} us")
