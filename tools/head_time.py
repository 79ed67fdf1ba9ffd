"""Isolated timing of the pointwise head: imf_pointwise_head (one launch) against the two variant-6 convolution launches
it replaces (conv1_tr + final), at the row counts of one S50k fragment and of the pair.  usage: python tools/head_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from imfnet_amd import ops
from imfnet_amd.ops import Rulebook
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w1 = (torch.randn((1, 96, 64), generator=g) / 96 ** 0.5).to(dev); w2 = (torch.randn((1, 64, 32), generator=g) / 8).to(dev)
sc, sh, bias = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev), torch.randn(32, device=dev)
w1p, w2p = ops.pack_weights(w1, split16=True), ops.pack_weights(w2, split16=True)
for n in (51232, 103396, 200000):
    a, b = torch.randn(n, 64, device=dev), torch.randn(n, 32, device=dev)
    rb = Rulebook(None, None, None, (n + 63) // 64 * 64, n, 1)
    out1, out2, hid = torch.empty(n, 32, device=dev), torch.empty(n, 32, device=dev), torch.empty(n, 64, device=dev)
    t1, t2 = [], []
    for r in range(12):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ops.pointwise_head(a, b, w1p, w2p, scale1=sc, shift1=sh, relu1=True, shift2=bias, l2norm=True, out=out1)
        e1.record()
        ops.spconv(a, w1p, 64, rb, in_b=b, scale=sc, shift=sh, relu=True, variant=6, out=hid)
        ops.spconv(hid, w2p, 32, rb, shift=bias, l2norm=True, variant=6, out=out2)
        e2.record(); torch.cuda.synchronize()
        t1.append(e0.elapsed_time(e1) * 1e3); t2.append(e1.elapsed_time(e2) * 1e3)
    gb = n * (96 + 32) * 4 / 1e9
    print(f"n={n}: fused head {np.median(t1[2:]):6.1f} us ({gb / np.median(t1[2:]) * 1e6 / 1e3:5.2f} TB/s of rows in + out)   "
          f"two launches {np.median(t2[2:]):6.1f} us   identical {bool(torch.equal(out1, out2))}")
