import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from imfnet_amd import ops, sparse as ME
z = np.load(os.path.join(ROOT, "tests/golden/fixture_clouds.npz"))
xyz = torch.as_tensor(z["cloud_bin_0"][::3].astype(np.float64)).cuda()
lv = ops.voxelize(xyz, 0.05); ops.sync_levels([lv])
cm = ME.CoordinateManager(lv); rb = cm.conv_rulebook(1, 3, 1)
g = torch.Generator().manual_seed(0)
w = torch.randn(27, 32, 32, generator=g) * 0.05
wp = ops.pack_weights(w.cuda(), split16=True); w0 = ops.pack_weights(w.cuda())
for mag in (6e4, 7e4, 1e5, 1.3e5, 1.4e5, 3e5, 1e6):
    f = torch.full((lv.n, 32), float(mag)).cuda()
    y6 = ops.spconv(f, wp, 32, rb, variant=6); y0 = ops.spconv(f, w0, 32, rb, variant=0)
    print(mag, "finite", bool(torch.isfinite(y6).all()), "rel err vs fp32 %.3e" % float(((y6 - y0).abs().max() / y0.abs().max())))
