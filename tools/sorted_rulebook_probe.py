#!/usr/bin/env python3
"""What would tiles of rows with SIMILAR neighbour-occupancy patterns buy?  The kernels walk, per 64-row tile, every offset any
of its rows has a neighbour at; in slot = row order ~99.7 % of the (tile, offset) pairs are active although only ~52 % of
the (row, offset) pairs exist.  This probe permutes the SLOTS of the pair's rulebooks on the host (rows keep their numbers:
tile_rows[slot] = row, nbr[k][slot] = the row's neighbour) so that rows are sorted by their occupancy mask -- globally, or
inside windows of consecutive rows (to keep the gathers local) -- and times the same convolutions on the permuted maps.
usage: [VARIANT=3] python tools/sorted_rulebook_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
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


def permuted(rb, window):
    K, S, n = rb.kvol, rb.n_slots, rb.n_out
    nbr = rb.nbr.view(K, S)
    rows = rb.tile_rows[:S] if rb.tile_rows is not None else torch.arange(S, device=dev, dtype=torch.int32)
    valid = (rows >= 0) if rb.tile_rows is not None else (torch.arange(S, device=dev) < n)
    occ = (nbr >= 0) & valid[None, :]                       # [K, S]
    frac = occ.float().sum(1) / max(1, int(valid.sum()))
    order_k = torch.argsort((frac - 0.5).abs())             # most informative offsets in the top bits
    key = torch.zeros(S, dtype=torch.int64, device=dev)
    for kk in order_k.tolist():
        key = (key << 1) | occ[kk].long()
    slot = torch.arange(S, device=dev)
    win = (slot // window) if window else torch.zeros_like(slot)
    full = (~valid).long() << 62 | win << 30 | key          # padding slots last; K <= 27 < 30 bits
    perm = torch.argsort(full, stable=True)                 # new slot s takes old slot perm[s]
    new_rows = torch.where(valid[perm], rows[perm].int() if rb.tile_rows is not None else perm.int(), torch.full_like(perm, -1).int())
    new_nbr = nbr[:, perm].contiguous()
    new_nbr = torch.where(valid[perm][None, :], new_nbr, torch.full_like(new_nbr, -1))
    o = (new_nbr >= 0).view(K, S // 64, 64).any(2)          # [K, tiles]
    bits = (o.long() << torch.arange(K, device=dev)[:, None]).sum(0)
    mask = torch.zeros((S // 64, 4), dtype=torch.int64, device=dev)
    mask[:, 0] = bits
    mask = mask.to(torch.int32).view(-1)                     # (bit 31 never set: K <= 27)
    rbn = ops.Rulebook(new_rows.contiguous(), new_nbr.view(-1), mask.contiguous(), S, n, K, rb.max_active)
    return rbn, float(o.float().mean()), float((new_nbr >= 0).view(K, S // 16, 16).any(2).float().mean())


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


SHAPES = [("block1   32->32  L0", 32, 32, 0, 1, None), ("block1_tr 64->64 L0", 64, 64, 0, 1, "wave4h"), ("block1_tr 64->64 L0 (g)", 64, 64, 0, 1, None),
          ("conv2    32->64  L0->1", 32, 64, 0, 2, "wave4"), ("block2   64->64  L1", 64, 64, 1, 1, "wave4"),
          ("conv3    64->128 L1->2", 64, 128, 1, 2, "wave8"), ("block3  128->128 L2", 128, 128, 2, 1, "wave8"),
          ("conv4   128->256 L2->3", 128, 256, 2, 2, "wave8u"), ("block4  256->256 L3", 256, 256, 3, 1, "wave8u")]
print("%-26s %10s | %s" % ("layer", "as built", "  ".join("%-22s" % ("window %s" % (w or "all")) for w in (None, 16384, 4096, 1024))))
for name, cin, cout, li, stride, staging in SHAPES:
    rb = cm.conv_rulebook(1 << li, 3, stride)
    n_in = levels[li].n
    fa = torch.randn(n_in, cin, generator=g).to(dev)
    w = ops.pack_weights((torch.randn(rb.kvol, cin, cout, generator=g) * 0.05).to(dev), variant=VARIANT)
    out = torch.empty(rb.n_out, cout, device=dev)
    kw = dict(variant=VARIANT, out=out, staging=staging, split_k=1)
    t0 = timed(lambda: ops.spconv(fa, w, cout, rb, **kw))
    ref = out.clone()
    cols = []
    for window in (None, 16384, 4096, 1024):
        rbn, act64, act16 = permuted(rb, window)
        t = timed(lambda: ops.spconv(fa, w, cout, rbn, **kw))
        err = float((out - ref).abs().max())
        cols.append("%6.1f us (%.2f/%.2f) %s" % (t, act64, act16, "" if err < 1e-4 else "ERR %.1e" % err))
    print("%-26s %7.1f us | %s" % (name, t0, "  ".join("%-22s" % c for c in cols)))
print("(in brackets: fraction of active (64-row tile, offset) / (16-row block, offset) pairs after the permutation)")
