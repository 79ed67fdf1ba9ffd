#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.  Runs ONLY in the build
container (needs /root/reference); the GPU box and the test-suite use the .npz files.

What it does
  1. copies the reference's *data* fixtures (files/cloud_bin_{0,1}.ply points as float32,
     the two PNGs resized to 120x160, and the xyz rows of files/3D_head_map.ply -- the
     only result of the path the reference itself pins) into compact .npz files;
  2. imports the reference's own model/*.py and util/misc.py VERBATIM over the
     MinkowskiEngine stand-in in oracle/me_shim and a torchvision stub, loads seeded
     weights with strict=True (proves the 361-key state_dict schema), and runs
     util.misc.extract_features end to end;
  3. checks oracle/imf_oracle.py's restatement against those outputs and writes them as
     golden descriptors / per-stage taps.
No reference source text is copied; only inputs and numeric outputs are stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "me_shim"))
import imf_oracle as O  # noqa: E402


def read_ply_xyz(path):
    with open(path, "rb") as f:
        props, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                props.append(line.split()[1:])
            elif line == "end_header":
                break
        m = {"float": "<f4", "double": "<f8", "uchar": "u1"}
        dt = np.dtype([(p[1], m[p[0]]) for p in props])
        a = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return np.stack([a["x"], a["y"], a["z"]], 1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import matplotlib.image as mpimg

    # ---- 1. data fixtures -------------------------------------------------------
    clouds, images = {}, {}
    for i in (0, 1):
        xyz = read_ply_xyz(f"{REF}/files/cloud_bin_{i}.ply")
        assert xyz.dtype == np.float32
        clouds[i] = xyz
        png = mpimg.imread(f"{REF}/files/cloud_bin_{i}_0.png")            # generate_desc.py:92
        assert png.dtype == np.float32 and png.shape == (480, 640, 3)
        images[i] = O.resize_bilinear(png, 120, 160)                      # uio.py:31-40 (A.6)
    head = read_ply_xyz(f"{REF}/files/3D_head_map.ply")                   # float64 [18977,3]
    assert (head.astype(np.float32).astype(np.float64) == head).all()
    np.savez_compressed(f"{HERE}/fixture_clouds.npz", cloud_bin_0=clouds[0], cloud_bin_1=clouds[1])
    np.savez_compressed(f"{HERE}/fixture_images.npz", image_0=images[0], image_1=images[1],
                        png0_rows=mpimg.imread(f"{REF}/files/cloud_bin_0_0.png")[::60, ::80].copy())
    np.savez_compressed(f"{HERE}/head_map_xyz.npz", xyz=head.astype(np.float32))

    # the reference-pinned result: voxelise cloud_bin_0 @2.5cm == 3D_head_map rows
    c25, i25 = O.voxelize(clouds[0].astype(np.float64), 0.025)
    assert len(i25) == 18977 and (clouds[0].astype(np.float64)[i25] == head).all()
    print("voxelize == 3D_head_map.ply: OK (18977 rows)")

    # ---- 2. the reference's own model code over the stand-ins ---------------------
    sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.models.utils")
    ip = "img_encoder.backbone."
    tvu.load_state_dict_from_url = lambda *a, **k: {k_[len(ip):]: v for k_, v in sd.items() if k_.startswith(ip)}
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu})
    sys.path.insert(0, REF)
    from model import load_model                      # reference code, verbatim
    from util.misc import extract_features            # reference code, verbatim
    Model = load_model("ResUNetBN2C")
    model = Model(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    ref_keys = list(model.state_dict().keys())
    assert len(ref_keys) == 361, len(ref_keys)
    model.load_state_dict(sd, strict=True)
    model.eval()
    schema = {k: list(v.shape) for k, v in model.state_dict().items()}

    gold = {}
    with torch.no_grad():
        # config 1 (plumbing): cloud_bin_0 @ 5 cm
        img0 = np.transpose(images[0], (2, 0, 1))[None]                  # generate_desc.py:96-97
        xyz0 = clouds[0].astype(np.float64)                              # Open3D widens to f64
        xyz_down, Fref = extract_features(model, xyz=xyz0, voxel_size=0.05, device=torch.device("cpu"),
                                          skip_check=True, image=img0)
        Fref = Fref.numpy()
        assert Fref.shape == (5182, 32)
        taps = {}
        xd2, Fo = O.extract_features(sd, xyz0, 0.05, img0)
        d = np.abs(Fo.numpy() - Fref).max()
        print(f"S5: reference-wiring vs restatement max|d| = {d:.3e}")
        assert d < 1e-6 and (xd2 == xyz_down).all()
        gold["S5_F"] = Fref
        gold["S5_xyz_down_f32"] = xyz_down.astype(np.float32)

        # crop of cloud_bin_0 @ 2.5 cm (<= 2k voxels), full descriptors + taps
        sel = (np.abs(xyz0[:, 0] - 0.3) < 0.35) & (np.abs(xyz0[:, 1] + 0.2) < 0.35)
        crop = xyz0[sel]
        xdc, Fc = extract_features(model, xyz=crop, voxel_size=0.025, device=torch.device("cpu"),
                                   skip_check=True, image=img0)
        cc, ic = O.voxelize(crop, 0.025)
        Fc_o = O.resunet_forward(sd, cc, img0, taps=taps)
        d = np.abs(Fc_o.numpy() - Fc.numpy()).max()
        print(f"crop25 ({len(cc)} voxels): reference-wiring vs restatement max|d| = {d:.3e}")
        assert d < 1e-6
        gold["crop_sel_idx"] = np.nonzero(sel)[0].astype(np.int32)
        gold["crop_F"] = Fc.numpy()
        for k in ("image_feat", "out_s8", "fused", "final"):
            gold["crop_tap_" + k] = taps[k].numpy()

        # config 2: full pair @ 2.5 cm -- checksums only (2 x 2.4 MB otherwise)
        for i in (0, 1):
            im = np.transpose(images[i], (2, 0, 1))[None]
            xd, Ff = extract_features(model, xyz=clouds[i].astype(np.float64), voxel_size=0.025,
                                      device=torch.device("cpu"), skip_check=True, image=im)
            Ff = Ff.numpy().astype(np.float64)
            gold[f"S25_{i}_M"] = np.int64(len(xd))
            gold[f"S25_{i}_colsum"] = Ff.sum(0)
            gold[f"S25_{i}_rows"] = Ff[:: max(1, len(Ff) // 256)][:256].astype(np.float32)
            print(f"S25 cloud_bin_{i}: M={len(xd)}")

        # attention block and image encoder, reference modules run directly
        af_in = torch.randn(1, 413, 256, generator=torch.Generator().manual_seed(1))
        ctx = torch.randn(1, 300, 128, generator=torch.Generator().manual_seed(2))
        gold["af_in"], gold["af_ctx"] = af_in[0].numpy(), ctx[0].numpy()
        gold["af_out"] = model.attention_fusion(ctx, queries_encoder=af_in)[0].numpy()
        gold["img_out"] = model.img_encoder(torch.as_tensor(img0)).numpy()

    np.savez_compressed(f"{HERE}/golden_descriptors.npz", **gold)
    import json
    with open(f"{HERE}/state_dict_schema.json", "w") as f:
        json.dump(schema, f, indent=0)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()


def write_pair_ground_truth():
    """tests/golden/redkitchen_pair_0_1_gt.npz: ground-truth pose and covariance of the in-tree fragment pair
    (files/cloud_bin_0.ply, cloud_bin_1.ply are 7-scenes-redkitchen fragments 0 and 1): the first block of
    benchmarks/3DMatch/7-scenes-redkitchen/gt.log and gt.info -- data, used by the evaluator tests."""
    import numpy as np
    root = "/root/reference/benchmarks/3DMatch/7-scenes-redkitchen/"
    log = open(root + "gt.log").read().splitlines()[:5]
    info = open(root + "gt.info").read().splitlines()[:7]
    pose = np.array([[float(v) for v in l.split()] for l in log[1:5]])
    cov = np.array([[float(v) for v in l.split()] for l in info[1:7]])
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "redkitchen_pair_0_1_gt.npz"), pose=pose,
             covariance=cov, indices=np.array([0, 1, 60]))
