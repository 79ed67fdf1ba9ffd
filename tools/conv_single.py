"""Run one sparse-conv shape N times with a given variant (for rocprofv3 --pmc runs).
usage: conv_single.py <variant> <split> [cin cout level]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload
variant, split = int(sys.argv[1]), int(sys.argv[2])
cin, cout, lvl = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (64, 64, 0)
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1 << lvl, 3, 1)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[lvl].n, cin, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, cin, cout, generator=g) * 0.05).to(dev), split16=(variant == 6))
for _ in range(10):
    ops.spconv(f, w, cout, rb, variant=variant, split_k=split)
torch.cuda.synchronize()
