"""Isolated timing of the network's sparse-convolution shapes (nothing else on the GPU): each shape is launched
REPS times back to back on one stream and timed with one event pair (main kernel + split-K reduce).
usage: [BATCH=2] [VARIANT=6|3|0] [IMF_LIB=...] conv_iso.py [staging ...]      staging: dma (default) | regs | wave8 | wave4 | wave4h (half-tile workgroups, bf16x3)
Several stagings print one column each (a shape a staging does not serve prints "-")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload, load_pair
stagings = sys.argv[1:] or [None]
VARIANT = int(os.environ.get("VARIANT", "6"))
REPS = 20
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
starts = None
if os.environ.get("BATCH") == "2":
    pts, imgs = load_pair(1.7)
    xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), voxel, 4, 0, item_starts=starts).result()
cm = ME.CoordinateManager.from_levels(levels)
g = torch.Generator().manual_seed(0)
SHAPES = [("block1", 32, 0, 32, "k3", 0), ("conv2", 32, 0, 64, "down", 0), ("block2", 64, 0, 64, "k3", 1),
          ("conv3", 64, 0, 128, "down", 1), ("block3", 128, 0, 128, "k3", 2), ("conv4", 128, 0, 256, "down", 2),
          ("block4", 256, 0, 256, "k3", 3), ("conv4_tr", 256, 0, 128, "up", 2), ("conv3_tr", 128, 128, 64, "up", 1),
          ("conv2_tr", 64, 64, 64, "up", 0), ("block2_tr", 64, 0, 64, "k3", 0), ("conv1_tr", 64, 32, 64, "k1", 0),
          ("final", 64, 0, 32, "k1", 0)]
tot = 0.0
for name, ca, cb, cout, kind, i in SHAPES:
    if kind == "k3":
        rb, n_in = cm.conv_rulebook(1 << i, 3, 1), levels[i].n
    elif kind == "down":
        rb, n_in = cm.conv_rulebook(1 << i, 3, 2), levels[i].n
    elif kind == "up":
        rb, n_in = cm.transpose_rulebook(2 << i, 3, 2), levels[i + 1].n
    else:
        rb, n_in = cm.conv_rulebook(1, 1, 1), levels[0].n
    fa = torch.randn(n_in, ca, generator=g).to(dev)
    fb = torch.randn(n_in, cb, generator=g).to(dev) if cb else None
    w = ops.pack_weights((torch.randn(rb.kvol, ca + cb, cout, generator=g) * 0.05).to(dev), variant=VARIANT)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    out = torch.empty(rb.n_out, cout, device=dev)
    cols = []
    for staging in stagings:
        if staging in ("wave8", "wave4", "wave4h", "wave8u", "wave4u", "wave4o", "wave4h4", "wave8h4") and (rb.kvol == 1 or cout % 64):
            cols.append(None)
            continue
        kw = dict(in_b=fb, scale=sc, shift=sh, relu=True, variant=VARIANT, out=out, staging=staging)
        for _ in range(3):
            ops.spconv(fa, w, cout, rb, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            ops.spconv(fa, w, cout, rb, **kw)
        e1.record(); torch.cuda.synchronize()
        cols.append(e0.elapsed_time(e1) * 1e3 / REPS)
    split = ops._lib.lib().imf_spconv_auto_split(rb.n_slots, cout, rb.max_active) if rb.kvol > 1 else 1
    tot = [a + (b if b is not None else (cols[0] or 0.0)) for a, b in zip(tot if isinstance(tot, list) else [0.0] * len(cols), cols)]
    print("%-10s k=%2d %3d->%3d slots=%6d split=%d  " % (name, rb.kvol, ca + cb, cout, rb.n_slots, split) +
          "  ".join("%7.1f us" % c if c is not None else "      -   " for c in cols))
print("sum (missing = first column):  " + "  ".join("%7.1f us" % t for t in tot) + "   [" + ", ".join(str(s) for s in stagings) + "]")
