#!/usr/bin/env python3
"""Round 6: what the occupancy-sorted maps + empty-block skipping buy, per layer shape and unit shape, in isolation.  For the
pair's maps (stride-1 at the four levels, the three strided ones): time the convolution on the map as built and on its
imf_rulebook_sort_by_occupancy twin, for every workgroup shape the kernels offer; also the sort's own time.
usage: [VARIANT=3] [IMF_LIB=...] python tools/sorted_conv_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from imfnet_amd import ops, sparse as ME
from bench import load_pair
VARIANT = int(os.environ.get("VARIANT", "3"))
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), 0.025, 4, 0, item_starts=starts).result()
cm = ME.CoordinateManager.from_levels(levels)
g = torch.Generator().manual_seed(0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def waste(rb, bs):
    K, S = rb.kvol, rb.n_slots
    occ = (rb.nbr.view(K, S) >= 0)
    return float(occ.view(K, S // bs, bs).any(2).sum() * bs) / float(occ.sum())


SHAPES = [("block1    32->32  L0", 32, 32, 0, 1), ("block2_tr 64->64  L0", 64, 64, 0, 1), ("conv2     32->64  L0->1", 32, 64, 0, 2),
          ("block2    64->64  L1", 64, 64, 1, 1), ("conv3     64->128 L1->2", 64, 128, 1, 2), ("block3   128->128 L2", 128, 128, 2, 1),
          ("conv4    128->256 L2->3", 128, 256, 2, 2), ("block4   256->256 L3", 256, 256, 3, 1)]
STAGINGS = (None, "wave4", "wave4h", "wave8", "wave8u")
print("variant", VARIANT, "lib", os.environ.get("IMF_LIB", "default"))
print("%-24s %-22s | %s" % ("layer", "issued/useful b16 (b64)", "  ".join("%-19s" % (s or "dma(g)") for s in STAGINGS)))
maps = {}
for name, cin, cout, li, stride in SHAPES:
    key = (li, stride)
    if key not in maps:
        rb = cm.conv_rulebook(1 << li, 3, stride)
        rs = ops.rulebook_sorted(rb)
        t_sort = timed(lambda: ops.rulebook_sorted(rb), 10)
        maps[key] = (rb, rs, t_sort)
    rb, rs, t_sort = maps[key]
    n_in = levels[li].n
    fa = torch.randn(n_in, cin, generator=g).to(dev)
    w = ops.pack_weights((torch.randn(rb.kvol, cin, cout, generator=g) * 0.05).to(dev), variant=VARIANT)
    out = torch.empty(rb.n_out, cout, device=dev)
    cols = []
    for staging in STAGINGS:
        if staging is not None and cout % 64:
            cols.append("-")
            continue
        if staging == "wave4h" and VARIANT != 3:
            cols.append("-")
            continue
        kw = dict(variant=VARIANT, out=out, staging=staging, split_k=1)
        try:
            t0 = timed(lambda: ops.spconv(fa, w, cout, rb, **kw))
            ref = out.clone()
            t1 = timed(lambda: ops.spconv(fa, w, cout, rs, **kw))
            err = float((out - ref).abs().max()) / float(ref.abs().max())
            cols.append("%6.1f -> %6.1f%s" % (t0, t1, "" if err < 1e-5 else " ERR%.0e" % err))
        except Exception as e:
            cols.append("fail")
    print("%-24s %.2f (%.2f) -> %.2f (%.2f) | %s   [sort %.1f us]" % (name, waste(rb, 16), waste(rb, 64), waste(rs, 16), waste(rs, 64),
                                                                     "  ".join("%-19s" % c for c in cols), t_sort))
