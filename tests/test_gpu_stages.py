"""Per-stage goldens on the GPU (the generator's crop_tap_* vectors were only consumed on the
CPU) and the kernel-offset-order switch of tools/checkpoint_fmr.py."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(sd, normalize=True):
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=normalize, conv1_kernel_size=5, D=3, config=None)
    m.load_state_dict(sd, strict=True)
    return m.eval().to(DEV)


def test_stage_taps_match_reference_wiring_goldens(clouds, images, golden, seeded_sd, monkeypatch):
    """crop_tap_out_s8 (encoder output at stride 8, after the ReLU), crop_tap_fused (bottleneck fusion output),
    crop_tap_final (pre-normalisation descriptors), crop_tap_image_feat: goldens of the reference's own model code.
    The stride-8 buffers are observed where the production kernels hand them over (the fusion call of the arena
    executor, which is bit-identical to the native one); tolerances: 2e-5 on O(1) activations."""
    from imfnet_amd import ops
    from imfnet_amd.extract import extract_features
    monkeypatch.setenv("IMFNET_PYTHON_EXECUTOR", "1")
    monkeypatch.setenv("IMFNET_NO_FRAGMENT_GRAPH", "1")
    seen = {}
    real = ops.fusion_attention_batched

    def spy(x, items, *a, **k):
        out = real(x, items, *a, **k)
        seen["out_s8"], seen["fused"] = x.clone(), out.clone()
        return out

    monkeypatch.setattr(ops, "fusion_attention_batched", spy)
    crop = clouds[0].astype(np.float64)[golden["crop_sel_idx"]]
    m = _model(seeded_sd)
    with torch.no_grad():
        _, F = extract_features(m, crop, voxel_size=0.025, device=torch.device(DEV), skip_check=True, image=images[0])
    assert np.abs(seen["out_s8"].cpu().numpy() - golden["crop_tap_out_s8"]).max() < 2e-5
    assert np.abs(seen["fused"].cpu().numpy() - golden["crop_tap_fused"]).max() < 2e-5
    assert np.abs(F.cpu().numpy() - golden["crop_F"]).max() < 1e-4
    m2 = _model(seeded_sd, normalize=False)                  # `final` before the L2 normalisation (resunet.py:226)
    with torch.no_grad():
        _, Fraw = extract_features(m2, crop, voxel_size=0.025, device=torch.device(DEV), skip_check=True, image=images[0])
    assert np.abs(Fraw.cpu().numpy() - golden["crop_tap_final"]).max() < 2e-5
    rows, _ = m._native_image().run(torch.as_tensor(images[0]).to(DEV))
    got = rows.view(1, 15, 20, 128).permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - golden["crop_tap_image_feat"]).max() < 6e-5


def test_kernel_offset_order_switch_is_live(seeded_sd):
    """tools/checkpoint_fmr.py on seeded weights: flipping every kernel from x-fastest to z-fastest offset order must
    change the descriptors substantially (on a trained checkpoint the inlier ratio of the in-tree ground-truth pair
    then collapses, which is how the MinkowskiEngine convention gets pinned); the flip is an involution."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import checkpoint_fmr as T
    from imfnet_amd.checkpoint import Config
    res, (F1, F2) = T.pair_inlier_ratio(seeded_sd, Config(), 0.05, 2000, 0)
    flipped = T.flip_kernel_offsets(seeded_sd)
    res_f, (G1, G2) = T.pair_inlier_ratio(flipped, Config(), 0.05, 2000, 0)
    assert res["mutual_matches"] > 0 and res_f["mutual_matches"] > 0 and 0.0 <= res["inlier_ratio"] <= 1.0
    assert float(np.abs(F1 - G1).max()) > 0.05                         # unit-norm descriptors: a different network
    back = T.flip_kernel_offsets(flipped)
    assert all(torch.equal(back[k], seeded_sd[k]) for k in seeded_sd)
    # oracle agreement for the flipped network too: the permutation is applied to the weights, the kernels are unchanged
    import imf_oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
    im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
    pts = z["cloud_bin_0"][::3].astype(np.float64)
    img = np.transpose(im["image_0"], (2, 0, 1))[None].copy()
    from imfnet_amd.extract import extract_features
    with torch.no_grad():
        _, F = extract_features(_model(flipped), pts, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=img)
    _, F_ref = O.extract_features(flipped, pts, 0.05, img)
    assert float((F.cpu() - F_ref).abs().max()) < 1e-4
    assert T.main([]) == 0                                              # the CLI itself, seeded weights
