"""The f16 range of the split-f16 convolution (variant 6) is guarded, never silent (ADVICE):
every kernel whose output feeds a variant-6 convolution raises IMF_FLAG_RANGE for NaN / |y| >= 65504, and the
harness recomputes a flagged fragment on the true-fp32 matrix instructions.  Plus a trained-checkpoint-like weight
distribution (small kernels, small running variances => large folded BatchNorm scales) through the default path."""
import warnings

import numpy as np
import pytest
import torch

import imf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(sd):
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    m.load_state_dict(sd, strict=True)
    return m.eval().to(DEV)


def test_conv_flags_outputs_beyond_f16(clouds):
    from imfnet_amd import ops
    from imfnet_amd import sparse as ME
    xyz = torch.as_tensor(clouds[0][::3].astype(np.float64)).to(DEV)
    lv = ops.voxelize(xyz, 0.05)
    ops.sync_levels([lv])
    cm = ME.CoordinateManager(lv)
    rb = cm.conv_rulebook(1, 3, 1)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(27, 32, 32, generator=g) * 0.05
    f = torch.randn(lv.n, 32, generator=g)
    wp = ops.pack_weights(w.to(DEV), split16=True)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.spconv(f.to(DEV), wp, 32, rb, variant=6, flags=flags)
    assert int(flags.item()) == 0                                        # ordinary magnitudes: silent
    wbig = ops.pack_weights((w * 10).to(DEV), split16=True)
    fin = (f.clamp(-3, 3) * 1.5e4).to(DEV)                              # inputs inside the range, outputs beyond it
    for kw in ({}, {"split_k": 3}):                                      # epilogue and split-K reduce paths
        flags.zero_()
        y = ops.spconv(fin, wbig, 32, rb, variant=6, flags=flags, **kw)
        assert torch.isfinite(y).all() and float(y.abs().max()) > 65504 and int(flags.item()) == 32   # flagged for the consumer
    flags.zero_()
    bad = ops.spconv((f * 1e5).to(DEV), wp, 32, rb, variant=6, flags=flags)   # inputs beyond f16: NaN/inf out, flagged
    assert not torch.isfinite(bad).all() and int(flags.item()) == 32
    flags.zero_()
    ok = ops.spconv((f * 1e5).to(DEV), ops.pack_weights(w.to(DEV)), 32, rb, variant=0, flags=flags)
    assert torch.isfinite(ok).all()                                      # the fp32-MFMA kernel has no such limit


def _huge_bottleneck_sd(seeded_sd):
    """Fusion output ~1e5 (bias of the last feed-forward layer), conv4_tr kernel 1e-5: its inputs leave the f16
    range while every product stays O(1) -- the un-normalised residual stream the verdict asked to exercise."""
    sd = {k: v.clone() for k, v in seeded_sd.items()}
    sd["attention_fusion.cross_attend_blocks.1.fn.net.2.bias"] += 1.0e5
    sd["conv4_tr.kernel"] *= 1.0e-5
    return sd


def test_out_of_range_fragment_is_recomputed_in_fp32(clouds, images, seeded_sd, monkeypatch, fast_mode):
    from imfnet_amd.extract import extract_features
    sd = _huge_bottleneck_sd(seeded_sd)
    xyz = clouds[0][::4].astype(np.float64)
    xd_ref, F_ref = O.extract_features(sd, xyz, 0.05, images[0])
    m = _model(sd)
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        xd, F = extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[0])
    assert any("f16 range" in str(w.message) for w in rec)
    assert (xd == xd_ref).all() and torch.isfinite(F).all()
    assert float((F.cpu() - F_ref).abs().max()) < 1e-4
    # the model is back on the default kernels afterwards, and a second fragment takes the same route (also via the runner)
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        xd, F2 = extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[0])
    assert any("f16 range" in str(w.message) for w in rec) and torch.equal(F, F2)
    # red without the guard: ignoring the flag gives a silently WRONG result (the NaNs of conv4_tr are squashed to
    # zero by the next ReLU epilogue -- fmaxf(NaN, 0) = 0 -- so the descriptors even look plausible)
    monkeypatch.setattr(type(m), "take_flags", lambda self, dev: 0)
    monkeypatch.setenv("IMFNET_NO_FRAGMENT_GRAPH", "1")
    with torch.no_grad():
        _, F3 = extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[0])
    assert float((F3.cpu() - F_ref).abs().max()) > 1e-3          # 10x the 1e-4 tolerance the guarded path meets


def test_capacity_mode_raises_the_range_flag(clouds, images, seeded_sd, fast_mode):
    from imfnet_amd.model.graph import FragmentRunner
    m = _model(_huge_bottleneck_sd(seeded_sd))
    r = FragmentRunner(m)
    xyz = torch.as_tensor(clouds[0][::4].astype(np.float64)).to(DEV)
    r.ratios, r.grid_words = [0.2, 0.06, 0.02, 0.006], 1 << 16
    res = r.run(xyz, [0], torch.as_tensor(images[0]).to(DEV), 0.05, stream=torch.cuda.Stream())
    assert res.flags & 32


def test_bf16x3_needs_no_range_guard(clouds, images, seeded_sd):
    """The default arithmetic (variant 3, bf16x3) carries every fp32 value: the fragment that sends variant 6 to its fp32
    recompute (activations ~1e5 in the bottleneck) is simply computed -- no flag, no warning, no second pass -- on the
    per-layer path, the exact path and in capacity mode, within the same 1e-4 of the oracle."""
    from imfnet_amd import ops
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model.graph import FragmentRunner
    assert ops.CONV_VARIANT == 3
    sd = _huge_bottleneck_sd(seeded_sd)
    xyz = clouds[0][::4].astype(np.float64)
    xd_ref, F_ref = O.extract_features(sd, xyz, 0.05, images[0])
    m = _model(sd)
    for _ in range(2):                                   # first call: exact path (teaches the runner); second: capacity mode
        with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            xd, F = extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[0])
        assert not any("f16 range" in str(w.message) for w in rec)
        assert (xd == xd_ref).all() and float((F.cpu() - F_ref).abs().max()) < 1e-4
    st = m.fragment_runner().stats
    assert st.get("redone", 0) == 0 and st["eager"] >= 1
    r = FragmentRunner(m)
    r.ratios, r.grid_words = [0.2, 0.06, 0.02, 0.006], 1 << 16
    res = r.run(torch.as_tensor(xyz).to(DEV), [0], torch.as_tensor(images[0]).to(DEV), 0.05, stream=torch.cuda.Stream())
    assert res.flags == 0 and float((res.F.cpu() - F_ref).abs().max()) < 1e-4


def _checkpoint_like(seeded_sd):
    """Small kernels (x0.05) under small running variances (x0.0025: folded scales x20) and wide BatchNorm affine
    terms -- what a trained checkpoint looks like, unlike the O(1) seeded one."""
    g = torch.Generator().manual_seed(11)
    sd = {k: v.clone() for k, v in seeded_sd.items()}
    for k in list(sd):
        sparse = not k.startswith("img_encoder") and not k.startswith("attention_fusion")
        if k.endswith(".kernel") and sparse and k != "final.kernel":
            sd[k] *= 0.05
        elif k.endswith("running_var") and sparse:
            sd[k] *= 0.0025
        elif k.endswith("running_mean") and sparse:
            sd[k] *= 0.05
        elif k.endswith("bn.weight") and sparse:
            sd[k] = sd[k] * torch.empty_like(sd[k]).uniform_(0.2, 3.0, generator=g)
    return sd


def test_checkpoint_like_weight_distribution(clouds, images, seeded_sd):
    """Checkpoint-like weights (`_checkpoint_like`), default path vs the oracle: 1e-4, no range flag."""
    from imfnet_amd.extract import extract_features
    sd = _checkpoint_like(seeded_sd)
    xyz = clouds[1][::2].astype(np.float64)
    xd_ref, F_ref = O.extract_features(sd, xyz, 0.05, images[1])
    m = _model(sd)
    with torch.no_grad(), warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        xd, F = extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[1])
    assert not any("f16 range" in str(w.message) for w in rec)
    assert (xd == xd_ref).all() and float((F.cpu() - F_ref).abs().max()) < 1e-4


def test_in_place_parameter_edits_invalidate_the_plans(clouds, images, seeded_sd):
    """packed weights / folded BatchNorm / native plans hold raw copies; an in-place edit in eval mode
    (param.data.copy_, running statistics set by hand, submodule.load_state_dict) must be picked up."""
    from imfnet_amd.extract import extract_features
    m = _model(seeded_sd)
    xyz = clouds[0][::6].astype(np.float64)
    run = lambda: extract_features(m, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True, image=images[0])[1].clone()
    with torch.no_grad():
        F0 = run()
        assert torch.equal(run(), F0)
        m.block2.conv1.kernel.mul_(1.5)                            # a sparse-conv kernel (in place, under no_grad)
        F1 = run()
        m.norm3.bn.running_var.mul_(2.0)                           # a BatchNorm buffer
        F2 = run()
        m.img_encoder.backbone.layer1[0].conv1.weight.data.mul_(0.5)   # `.data` edits bypass torch's version counters:
        m.invalidate()                                                  # ... they need the explicit call
        F3 = run()
    sd = {k: v.clone() for k, v in seeded_sd.items()}
    sd["block2.conv1.kernel"] = sd["block2.conv1.kernel"] * 1.5
    _, R1 = O.extract_features(sd, xyz, 0.05, images[0])
    sd["norm3.bn.running_var"] = sd["norm3.bn.running_var"] * 2.0
    _, R2 = O.extract_features(sd, xyz, 0.05, images[0])
    sd["img_encoder.backbone.layer1.0.conv1.weight"] = sd["img_encoder.backbone.layer1.0.conv1.weight"] * 0.5
    _, R3 = O.extract_features(sd, xyz, 0.05, images[0])
    assert float((F1.cpu() - R1).abs().max()) < 1e-4 and float((F1 - F0).abs().max()) > 1e-3
    assert float((F2.cpu() - R2).abs().max()) < 1e-4 and float((F2 - F1).abs().max()) > 1e-3
    assert float((F3.cpu() - R3).abs().max()) < 1e-4 and not torch.equal(F3, F2)


def test_range_flag_hits_over_a_fragment_set_at_2p5cm(seeded_sd, clouds, images):
    """how often does IMF_FLAG_RANGE fire?  Twelve fragments at the benchmark's 2.5 cm voxels (slabs of the
    in-tree pair under seeded scales 1.0-1.9, the emulation's recipe) through the capacity-mode stream under checkpoint-like
    weights (small kernels, BatchNorm folded to a x20 scale, shifted means): no fragment may raise the flag -- the stream
    counts a flagged fragment as `redone` -- and the largest |activation| the network produces stays orders of magnitude
    inside the f16 range; one fragment is checked against the oracle at 1e-4."""
    import imf_oracle as O
    from imfnet_amd.extract import extract_features, extract_features_stream
    sd = _checkpoint_like(seeded_sd)
    m = _model(sd)
    rng = np.random.default_rng(11)
    frags = []
    for i in range(12):
        base = clouds[i % 2]
        d = rng.normal(size=3).astype(np.float32)
        proj = base @ (d / np.linalg.norm(d))
        frac = rng.uniform(0.3, 0.8)
        lo = np.quantile(proj, rng.uniform(0.0, 1.0 - frac))
        keep = np.sort(np.flatnonzero(proj >= lo)[: int(frac * len(base))])
        frags.append(((base[keep] * np.float32(rng.uniform(1.0, 1.9))).astype(np.float64), images[i % 2]))
    dev = torch.device(DEV)
    with torch.no_grad():
        xd0, F0 = extract_features(m, frags[0][0], voxel_size=0.025, device=dev, skip_check=True, image=frags[0][1])
        assert m.take_flags(dev) == 0
        runner = m.fragment_runner()
        before = runner.stats["redone"]
        outs = list(extract_features_stream(m, iter(frags), 0.025, dev, batch=1))
    assert runner.stats["redone"] == before, ("a fragment raised a flag (range: 32, or capacity) under checkpoint-like "
                                              "weights: flags %d" % runner.stats.get("redone_flags", -1))
    assert len(outs) == 12 and all(np.isfinite(F).all() for _, F in outs)
    assert (outs[0][0] == xd0).all() and np.abs(outs[0][1] - F0.cpu().numpy()).max() < 2e-6
    k = int(np.argmin([len(x) for x, _ in frags]))                      # the smallest fragment against the oracle
    xd_ref, F_ref = O.extract_features(sd, frags[k][0], 0.025, frags[k][1])
    assert (outs[k][0] == xd_ref).all() and np.abs(outs[k][1] - F_ref.numpy()).max() < 1e-4
