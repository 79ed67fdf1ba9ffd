"""A/B timing of the sparse-conv kernel variants on the S50k fragment's real rulebooks (one process,
interleaved rounds, HIP events around the main kernel).  usage: python tools/conv_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd import sparse as ME
from bench import load_workload

dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(float(os.environ.get("SCALE", 1.7)), 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
n = [l.n for l in levels]
print("levels", n)
cases = [  # name, rulebook, cin, cout, rows_in
    ("k3@1 64->64", cm.conv_rulebook(1, 3, 1), 64, 64, n[0]),
    ("k3@1 32->32", cm.conv_rulebook(1, 3, 1), 32, 32, n[0]),
    ("k1@1 96->64", cm.conv_rulebook(1, 1, 1), 96, 64, n[0]),
    ("k1@1 64->32", cm.conv_rulebook(1, 1, 1), 64, 32, n[0]),
    ("up@1 128->64", cm.transpose_rulebook(2, 3, 2), 128, 64, n[1]),
    ("dn 1->2 32->64", cm.conv_rulebook(1, 3, 2), 32, 64, n[0]),
    ("k3@2 64->64", cm.conv_rulebook(2, 3, 1), 64, 64, n[1]),
    ("k3@4 128->128", cm.conv_rulebook(4, 3, 1), 128, 128, n[2]),
    ("k3@8 256->256", cm.conv_rulebook(8, 3, 1), 256, 256, n[3]),
    ("up@4 256->128", cm.transpose_rulebook(8, 3, 2), 256, 128, n[3]),
]
variants = [("v0 auto", dict(variant=0)),
            ("v6 auto", dict(variant=6)), ("v6 s1", dict(variant=6, split_k=1)), ("v6 s2", dict(variant=6, split_k=2)),
            ("v6 s3", dict(variant=6, split_k=3)), ("v6 s4", dict(variant=6, split_k=4)), ("v6 s6", dict(variant=6, split_k=6)),
            ("v6 s8", dict(variant=6, split_k=8)), ("v6 wave8", dict(variant=6, staging="wave8")),
            ("v6 wave4", dict(variant=6, staging="wave4"))]      # (variant 6 + in-launch combine: diagnostic builds only)
g = torch.Generator().manual_seed(0)
for name, rb, cin, cout, rows in cases:
    f = torch.randn(rows, cin, generator=g).to(dev)
    wt = (torch.randn(rb.kvol, cin, cout, generator=g) * 0.05).to(dev)
    wp = {0: ops.pack_weights(wt), 6: ops.pack_weights(wt, split16=True)}
    res = {}
    outs = {}
    for rnd in range(6):
        for vn, kw in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ws_warm = ops.spconv(f, wp[kw['variant']], cout, rb, **kw)       # allocator warm
            e0.record()
            outs[vn] = ops.spconv(f, wp[kw['variant']], cout, rb, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)                                         # whole op: main kernel + split-K reduce
            if rnd: res.setdefault(vn, []).append(ms * 1e3)
    ref = outs["v0 auto"]
    line = f"{name:16s} n_slots={rb.n_slots:6d} " + "  ".join(f"{vn}: {np.median(v):7.1f}us" for vn, v in res.items())
    err = max(float((outs[vn] - ref).abs().max()) for vn, _ in variants)
    print(line, f" max|d|={err:.1e}")
