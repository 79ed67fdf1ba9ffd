"""Seeded random weights in the reference's state_dict schema: what bench.py, the CLI's
`--seeded_weights` plumbing runs and the tools load when no checkpoint is reachable.

conv kernels U(+-1/sqrt(K*Cin)) (MinkowskiEngine's default init), BatchNorm gamma U(.5,1.5), beta U(-.1,.1),
running_mean N(0,.1), running_var U(.5,1.5) so that every op does something; Linear / LayerNorm / Conv2d likewise.
Deterministic in `seed`.  The oracle holds its own copy of this generator (oracle/imf_oracle.py); a test keeps the two
equal, so that product-side runs and oracle-side checks of "seed 0" mean the same network.
"""
import math

import torch

RESUNETBN2C = dict(CH=[None, 32, 64, 128, 256], TR=[None, 64, 64, 64, 128])      # model/resunet.py:309-314


def seeded_state_dict(seed=0, conv1_kernel_size=5, in_channels=1, out_channels=32,
                      cfg=None, with_unused_image_layers=False):
    """Random weights in the reference's state_dict schema (SURVEY App. B, config 2 of
    §8d): conv kernels U(+-1/sqrt(K*Cin)), BN gamma U(.5,1.5), beta U(-.1,.1),
    mean N(0,.1), var U(.5,1.5).  Deterministic in `seed`."""
    cfg = cfg or RESUNETBN2C
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def U(shape, a, b):
        return torch.rand(shape, generator=g) * (b - a) + a

    def conv(name, K, cin, cout):
        bound = 1.0 / math.sqrt(K * cin)
        sd[name + ".kernel"] = U((K, cin, cout) if K > 1 else (cin, cout), -bound, bound)

    def bn(name, c, mid=".bn"):
        sd[f"{name}{mid}.weight"] = U((c,), 0.5, 1.5)
        sd[f"{name}{mid}.bias"] = U((c,), -0.1, 0.1)
        sd[f"{name}{mid}.running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[f"{name}{mid}.running_var"] = U((c,), 0.5, 1.5)
        sd[f"{name}{mid}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(name, c):
        conv(name + ".conv1", 27, c, c); bn(name + ".norm1", c)
        conv(name + ".conv2", 27, c, c); bn(name + ".norm2", c)

    CH, TR = cfg["CH"], cfg["TR"]
    conv("conv1", conv1_kernel_size ** 3, in_channels, CH[1]); bn("norm1", CH[1]); block("block1", CH[1])
    conv("conv2", 27, CH[1], CH[2]); bn("norm2", CH[2]); block("block2", CH[2])
    conv("conv3", 27, CH[2], CH[3]); bn("norm3", CH[3]); block("block3", CH[3])
    conv("conv4", 27, CH[3], CH[4]); bn("norm4", CH[4]); block("block4", CH[4])
    p = "attention_fusion.cross_attend_blocks."
    ld, dim, inner = CH[4], 128, CH[4] // 2

    def lin(name, cout, cin, bias=True):
        b = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = U((cout, cin), -b, b)
        if bias:
            sd[name + ".bias"] = U((cout,), -b, b)

    def ln(name, c):
        sd[name + ".weight"] = U((c,), 0.5, 1.5)
        sd[name + ".bias"] = U((c,), -0.1, 0.1)

    lin(p + "0.fn.to_q", inner, ld, False); lin(p + "0.fn.to_kv", 2 * inner, dim, False)
    lin(p + "0.fn.to_out", ld, inner)
    ln(p + "0.norm", ld); ln(p + "0.norm_context", dim)
    lin(p + "1.fn.net.0", ld * 8, ld); lin(p + "1.fn.net.2", ld, ld * 4); ln(p + "1.norm", ld)
    conv("conv4_tr", 27, CH[4], TR[4]); bn("norm4_tr", TR[4]); block("block4_tr", TR[4])
    conv("conv3_tr", 27, CH[3] + TR[4], TR[3]); bn("norm3_tr", TR[3]); block("block3_tr", TR[3])
    conv("conv2_tr", 27, CH[2] + TR[3], TR[2]); bn("norm2_tr", TR[2]); block("block2_tr", TR[2])
    conv("conv1_tr", 1, CH[1] + TR[2], TR[1])
    conv("final", 1, TR[1], out_channels)
    sd["final.bias"] = U((1, out_channels), -0.1, 0.1)

    ip = "img_encoder.backbone."

    def conv2d(name, cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn2d(name, c):
        bn(name, c, mid="")

    conv2d(ip + "conv1", 64, 3, 7); bn2d(ip + "bn1", 64)
    layers = [(1, 64, 64, 3), (2, 64, 128, 4)]
    if with_unused_image_layers:
        layers += [(3, 128, 256, 6), (4, 256, 512, 3)]
    for li, cin, cout, nb in layers:
        for i in range(nb):
            q = f"{ip}layer{li}.{i}"
            conv2d(q + ".conv1", cout, cin if i == 0 else cout, 3); bn2d(q + ".bn1", cout)
            conv2d(q + ".conv2", cout, cout, 3); bn2d(q + ".bn2", cout)
            if i == 0 and li > 1:
                conv2d(q + ".downsample.0", cout, cin, 1); bn2d(q + ".downsample.1", cout)
    if with_unused_image_layers:
        lin(ip + "fc", 1000, 512)
    return sd
