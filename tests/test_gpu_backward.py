"""Training backward of the sparse convolutions (imfnet_amd/autograd.py, csrc/backward.hip) against torch
autograd through the CPU oracle's restatement (index_add / matmul): per kernel-map kind, and for the whole network."""
import numpy as np
import pytest
import torch

import imf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def geo(clouds):
    from imfnet_amd import ops
    from imfnet_amd import sparse as ME
    xyz = clouds[0][::4].astype(np.float64)
    lv = ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.05)
    ops.sync_levels([lv])
    cm = ME.CoordinateManager(lv)
    cm.build_pyramid(8)
    coords, _ = O.voxelize(xyz, 0.05)
    return cm, O.Geometry(coords), ME


def _rel(a, b):
    return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("kind", ["k3_stride1", "k1", "stride2", "transposed", "small_cin_k5"])
def test_conv_gradients_match_the_oracle_autograd(geo, kind):
    cm, g, ME = geo
    gen = torch.Generator().manual_seed(3)
    conv, convT = ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose
    if kind == "k3_stride1":
        m, ts, nbr, n_in = conv(64, 32, kernel_size=3, stride=1, dimension=3), 2, g.k3[1], len(g.levels[1])
    elif kind == "k1":
        m, ts, nbr, n_in = conv(96, 64, kernel_size=1, stride=1, dimension=3), 1, None, len(g.levels[0])
    elif kind == "stride2":
        m, ts, nbr, n_in = conv(32, 64, kernel_size=3, stride=2, dimension=3), 1, g.down[0], len(g.levels[0])
    elif kind == "transposed":
        m, ts, nbr, n_in = convT(64, 32, kernel_size=3, stride=2, dimension=3), 4, g.up[1], len(g.levels[2])
    else:
        m, ts, nbr, n_in = conv(1, 32, kernel_size=5, stride=1, dimension=3), 1, g.k_first, len(g.levels[0])
    m = m.to(DEV)
    feat = torch.randn(n_in, m.in_channels, generator=gen)
    f_gpu = feat.to(DEV).requires_grad_(m.in_channels >= 32)
    x = ME.SparseTensor(f_gpu, coordinate_map_key=ME.CoordinateMapKey(ts), coordinate_manager=cm)
    out = m(x).F
    f_ref = feat.clone().requires_grad_(True)
    w_ref = m.kernel.detach().cpu().clone().requires_grad_(True)
    out_ref = O.spconv(f_ref, w_ref, nbr)
    assert out.shape == out_ref.shape and _rel(out.detach(), out_ref.detach()) < 1e-5
    go = torch.randn(out_ref.shape, generator=gen)
    out.backward(go.to(DEV))
    out_ref.backward(go)
    assert m.kernel.grad.shape == w_ref.grad.shape and _rel(m.kernel.grad, w_ref.grad) < 1e-5
    if m.in_channels >= 32:
        assert _rel(f_gpu.grad, f_ref.grad) < 1e-5


def test_whole_network_gradients_match_the_oracle(clouds, images, seeded_sd):
    """Fine-tuning configuration: eval-mode BatchNorm statistics, gradients to every parameter.  loss = <F, T>."""
    from imfnet_amd.extract import sparse_tensor_from_points
    from imfnet_amd.model import load_model
    xyz = clouds[1][::6].astype(np.float64)
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().to(DEV)
    st, _ = sparse_tensor_from_points(xyz, 0.05, torch.device(DEV))
    F = m(st, torch.as_tensor(images[1]).to(DEV)).F
    assert F.requires_grad                                    # routed through forward_layers
    T = torch.randn(F.shape, generator=torch.Generator().manual_seed(9))
    (F * T.to(DEV)).sum().backward()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in seeded_sd.items()}
    coords, _ = O.voxelize(xyz, 0.05)
    F_ref = O.resunet_forward(sd, coords, images[1])
    assert float((F.detach().cpu() - F_ref.detach()).abs().max()) < 1e-4
    (F_ref * T).sum().backward()
    grads = dict(m.named_parameters())
    checked = 0
    for k, ref in sd.items():
        if not getattr(ref, "requires_grad", False) or ref.grad is None:
            continue
        got = grads[k].grad
        assert got is not None, k
        scale = float(ref.grad.abs().max())
        if scale < 1e-7:
            continue
        assert float((got.cpu() - ref.grad).abs().max()) < 2e-3 * scale + 1e-6, k
        checked += 1
    assert checked > 100
    # one SGD step on the GPU model lowers the loss (the path trains)
    with torch.no_grad():
        loss0 = float((F * T.to(DEV)).sum())
        for p in m.parameters():
            if p.grad is not None:
                p -= 1e-3 * p.grad
    F1 = m(st, torch.as_tensor(images[1]).to(DEV)).F
    assert float((F1.detach() * T.to(DEV)).sum()) < loss0
    # training mode proper (BatchNorm batch statistics, running stats updated): gradients reach every layer, all finite
    m.train()
    m.zero_grad()
    rm0 = m.norm3.bn.running_mean.clone()
    F2 = m(st, torch.as_tensor(images[1]).to(DEV)).F
    (F2 * T.to(DEV)).sum().backward()
    assert not torch.equal(m.norm3.bn.running_mean, rm0)
    for name, p in m.named_parameters():
        used = not any(s in name for s in ("layer3", "layer4", ".fc."))          # stored, never executed (resnet.py:205-216)
        if used:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
