#!/usr/bin/env python3
"""The decisive convention check (VERDICT r1 missing #1), ready to run the moment a reference checkpoint appears:

    python tools/checkpoint_fmr.py -m /path/to/checkpoint.pth [--voxel 0.025] [--keypoints 5000]

runs the two in-tree 7-scenes-redkitchen fragments (files/cloud_bin_0.ply / cloud_bin_1.ply of the reference,
shipped here as tests/golden fixtures with their colour images) through generate_desc's extract_features, matches the
descriptors mutually (scripts/evaluation_3dmatch.py:207-234) under the benchmark's ground-truth pose
(benchmarks/3DMatch/7-scenes-redkitchen/gt.log:1-5) and reports the inlier ratio.  With a TRAINED checkpoint the ratio
must clear the 5 % feature-match-recall threshold (tau_2 = 0.05 at tau_1 = 10 cm) -- and it collapses when the
MinkowskiEngine conventions this build restates are wrong.  `--flip` permutes every 3-D kernel from x-fastest to
z-fastest offset order (the one convention that cannot be pinned without MinkowskiEngine): on a trained checkpoint
exactly one of the two orders clears the threshold, which pins `kernel_offsets`.

Without -m it runs on seeded random weights: the ratios are then meaningless as a pass/fail signal (both orders are
random networks) but the descriptors must differ, which proves the switch is live.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def flip_kernel_offsets(sd):
    """x-fastest <-> z-fastest: W'[x + K y + K^2 z] = W[z + K y + K^2 x] for every [K^3, cin, cout] kernel."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".kernel") and v.dim() == 3 and v.shape[0] in (27, 125):
            K = round(v.shape[0] ** (1 / 3))
            out[k] = v.reshape(K, K, K, *v.shape[1:]).transpose(0, 2).reshape(v.shape).contiguous()
        else:
            out[k] = v
    return out


def pair_inlier_ratio(sd, cfg, voxel, n_keypoints, seed, device="cuda:0"):
    from imfnet_amd.extract import extract_features
    from imfnet_amd.matching import feature_match, select_keypoints
    from imfnet_amd.model import load_model
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
    im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
    gt = np.load(os.path.join(ROOT, "tests", "golden", "redkitchen_pair_0_1_gt.npz"))
    model = load_model(cfg.model)(1, cfg.model_n_out, bn_momentum=cfg.bn_momentum, normalize_feature=cfg.normalize_feature,
                                  conv1_kernel_size=cfg.conv1_kernel_size, D=3, config=cfg)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(device)
    rng = np.random.RandomState(seed)
    data = []
    for k in (0, 1):
        pts = z[f"cloud_bin_{k}"].astype(np.float64)
        img = np.transpose(im[f"image_{k}"], (2, 0, 1))[None].copy()
        with torch.no_grad():
            xyz, F = extract_features(model, pts, voxel_size=voxel, device=torch.device(device), skip_check=True, image=img)
        F = F.cpu().numpy()
        sample = pts[rng.choice(len(pts), min(n_keypoints, len(pts)), replace=False)]      # evaluation_3dmatch.py:154-171
        sel = select_keypoints(sample, xyz, voxel)
        data.append((xyz[sel], F[sel], F))
    (k1, d1, F1), (k2, d2, F2) = data
    n_inl, ratio, m1, m2 = feature_match(k1, d1, k2, d2, gt["pose"], 0.1)
    return dict(inlier_ratio=float(ratio), inliers=int(n_inl), mutual_matches=int(len(m1)),
                keypoints=[int(len(k1)), int(len(k2))]), (F1, F2)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--model", default=None, help="reference checkpoint (.pth); default: seeded random weights")
    ap.add_argument("--voxel", type=float, default=None)
    ap.add_argument("--keypoints", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threshold", type=float, default=0.05, help="inlier-ratio threshold (FMR tau_2)")
    args = ap.parse_args(argv)
    from imfnet_amd.checkpoint import Config, load_checkpoint
    if args.model:
        sd, cfg = load_checkpoint(args.model)
        trained = True
    else:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import imf_oracle as O
        sd, cfg, trained = O.seeded_state_dict(seed=0, with_unused_image_layers=True), Config(), False
    voxel = args.voxel or float(cfg.voxel_size)
    res, (F1, F2) = pair_inlier_ratio(sd, cfg, voxel, args.keypoints, args.seed)
    res_f, (G1, G2) = pair_inlier_ratio(flip_kernel_offsets(sd), cfg, voxel, args.keypoints, args.seed)
    out = {"checkpoint": args.model or "seeded random weights", "voxel_size": voxel,
           "x_fastest (this build's reading of MinkowskiEngine's kernel_region)": res,
           "z_fastest (flipped)": res_f,
           "max_descriptor_change_under_flip": float(max(np.abs(F1 - G1).max(), np.abs(F2 - G2).max()))}
    if trained:
        ok, ok_f = res["inlier_ratio"] > args.threshold, res_f["inlier_ratio"] > args.threshold
        out["verdict"] = ("conventions PINNED: x-fastest clears the threshold, z-fastest does not" if ok and not ok_f else
                          "CONVENTION WRONG: only the flipped order clears the threshold -- flip kernel_offsets" if ok_f and not ok else
                          "inconclusive (both or neither order clear the threshold)")
    print(json.dumps(out, indent=1))
    return 0 if (not trained or "PINNED" in out.get("verdict", "")) else 1


if __name__ == "__main__":
    sys.exit(main())
