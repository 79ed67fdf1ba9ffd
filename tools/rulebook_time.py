"""Isolated timing of three rulebook builds of the S50k pair (memset of the tile masks + k_rulebook each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from bench import load_pair
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), 0.025, 4, 0, item_starts=starts).result()
for (a, b, k, name) in ((0, 0, 3, "k3@1"), (0, 1, 3, "down 1->2"), (1, 1, 3, "k3@2")):
    for _ in range(3):
        rb = ops.rulebook_conv(levels[a], levels[b], k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rb = ops.rulebook_conv(levels[a], levels[b], k)
    e1.record(); torch.cuda.synchronize()
    print("%-10s rows=%6d  %6.1f us per build" % (name, levels[b].n, e0.elapsed_time(e1) * 1e3 / 20))
