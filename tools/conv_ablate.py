import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1, 3, 1)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[0].n, 64, generator=g).to(dev)
VAR = int(os.environ.get("VAR","0"))
w = ops.pack_weights((torch.randn(27, 64, 64, generator=g) * 0.05).to(dev), split16=(VAR == 6))
out = []
for kw in (dict(variant=int(os.environ.get("VAR","0")), split_k=1),):
    ts = []
    for r in range(5):
        ops.TRACE = []
        ops.spconv(f, w, 64, rb, **kw); torch.cuda.synchronize()
        ts.append(ops.TRACE[0]["ev"].elapsed_ms() * 1e3); ops.TRACE = None
    out.append("%%s: %%.1f us" %% (kw, np.median(ts[1:])))
print("IMF_ABLATE=%%s  " %% os.environ.get("IMF_ABLATE", "0") + "  ".join(out))
''' % (ROOT, ROOT)
for ab in [int(x) for x in os.environ.get("ABLATIONS", "0,1,2,4,8,3,5,6,7,12,15").split(",")]:
    env = dict(os.environ, IMF_ABLATE=str(ab))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
