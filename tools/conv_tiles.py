"""Time the conv kernel for a varying number of 64-row tiles (residency / scaling study)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(2.2, 0.025)          # ~80k voxels so that 1024+ tiles exist
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1, 3, 1)
print("n1 =", levels[0].n, "tiles =", rb.n_slots // 64)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[0].n, 64, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, 64, 64, generator=g) * 0.05).to(dev))
full = rb.nbr.view(27, rb.n_slots)
for variant in (0, 3):
    line = []
    for tiles in (128, 256, 512, 768, 1024, 1280):
        ns = tiles * 64
        sub = ops.Rulebook(rb.tile_rows[:ns].contiguous(), full[:, :ns].contiguous().view(-1),
                           rb.tile_mask[: tiles * 4].contiguous(), ns, ns, 27)
        ts = []
        for r in range(5):
            ops.TRACE = []
            ops.spconv(f, w, 64, sub, variant=variant, split_k=1); torch.cuda.synchronize()
            ts.append(ops.TRACE[0]["ev"].elapsed_ms() * 1e3); ops.TRACE = None
        line.append(f"{tiles}: {np.median(ts[1:]):6.1f}us")
    print(f"variant {variant}  " + "  ".join(line))
