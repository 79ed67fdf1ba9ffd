"""GPU tests of the harness rows: the generate_desc CLI over a 3DMatch-layout
tree, host-array inputs, and a large (~200k voxel, KITTI-sized) fragment."""
import os

import numpy as np
import pytest
import torch

import imf_oracle as O

pytestmark = pytest.mark.gpu


def _write_ply(path, pts):
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
                b"property float y\nproperty float z\nend_header\n" % len(pts))
        f.write(np.ascontiguousarray(pts, dtype="<f4").tobytes())


def test_generate_desc_cli_matches_oracle(tmp_path, clouds, seeded_sd):
    """Same flags, directory contract and NPZ keys as scripts/generate_desc.py; descriptors vs the oracle."""
    from PIL import Image
    from imfnet_amd.checkpoint import Config
    from imfnet_amd import generate_desc as gd
    src, dst = tmp_path / "src", tmp_path / "dst"
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fixture_images.npz"))
    frags = {}
    for scene, ids in (("sceneA", (0, 1)), ("sceneB", (2,))):
        d = src / scene / "seq-01"
        d.mkdir(parents=True)
        for k in ids:
            pts = clouds[k % 2][k::7].copy()
            _write_ply(d / f"cloud_bin_{k}.ply", pts)
            img8 = np.clip(np.rint(z[f"image_{k % 2}"] * 255), 0, 255).astype(np.uint8)
            Image.fromarray(img8).save(d / f"cloud_bin_{k}_0.png")       # already 160x120: no resize
            frags[(scene, k)] = (pts.astype(np.float64), np.transpose(np.divide(img8, 255, dtype=np.float32), (2, 0, 1))[None])
    (src / "sceneA-evaluation").mkdir()                                   # skipped, as in the reference
    ckpt = tmp_path / "ckpt.pth"
    torch.save({"state_dict": seeded_sd, "config": dict(Config(voxel_size=0.05)), "epoch": 1}, ckpt)
    gd.main(["--source", str(src), "--target", str(dst), "-m", str(ckpt)])
    assert sorted(os.listdir(dst)) == ["sceneA", "sceneB"]
    dst_seq = tmp_path / "dst_sequential"                                 # the reference's sequential order
    gd.main(["--source", str(src), "--target", str(dst_seq), "-m", str(ckpt), "--workers", "0"])
    dst_grp = tmp_path / "dst_grouped"                                    # opt-in: fragments share forwards (batched call)
    gd.main(["--source", str(src), "--target", str(dst_grp), "-m", str(ckpt), "--batch_points", "-1"])
    for scene, k in frags:
        a = np.load(dst / scene / "seq-01" / f"cloud_bin_{k}.npz")
        b = np.load(dst_seq / scene / "seq-01" / f"cloud_bin_{k}.npz")
        c = np.load(dst_grp / scene / "seq-01" / f"cloud_bin_{k}.npz")
        assert all((a[key] == b[key]).all() for key in ("points", "xyz", "feature"))       # threads change nothing
        assert (c["points"] == b["points"]).all() and (c["xyz"] == b["xyz"]).all()
        assert np.abs(c["feature"] - b["feature"]).max() < 2e-6                              # batching: rounding only
    for (scene, k), (pts, img) in frags.items():
        out = np.load(dst / scene / "seq-01" / f"cloud_bin_{k}.npz")
        assert sorted(out.files) == ["feature", "points", "xyz"]
        assert out["points"].dtype == np.float64 and (out["points"] == pts).all()
        xyz_ref, F_ref = O.extract_features(seeded_sd, pts, 0.05, img)
        assert out["xyz"].dtype == np.float64 and (out["xyz"] == xyz_ref).all()
        assert out["feature"].dtype == np.float32 and out["feature"].shape == (len(xyz_ref), 32)
        assert np.abs(out["feature"] - F_ref.numpy()).max() < 1e-4


def test_generate_desc_cli_resizes_a_640x480_png(tmp_path, clouds, seeded_sd):
    """The data set's images are 640 x 480 (scripts/generate_desc.py:92-97 -> util/uio.py:33-40 resizes them to the
    checkpoint's 160 x 120): the CLI's decode -> resize -> forward on such a PNG (content that the 4:1 resize really mixes)
    against the oracle's resize (O.resize_bilinear) + forward."""
    from PIL import Image
    from imfnet_amd.checkpoint import Config
    from imfnet_amd import generate_desc as gd
    src, dst = tmp_path / "src", tmp_path / "dst"
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fixture_images.npz"))
    rng = np.random.default_rng(5)
    d = src / "scene" / "seq-01"
    d.mkdir(parents=True)
    frags = {}
    for k in (0, 1):
        big = np.kron(z[f"image_{k}"], np.ones((4, 4, 1), np.float32)) + rng.uniform(-0.08, 0.08, (480, 640, 3)).astype(np.float32)
        img8 = np.clip(np.rint(big * 255), 0, 255).astype(np.uint8)
        Image.fromarray(img8).save(d / f"cloud_bin_{k}_0.png")
        pts = clouds[k][k::6].copy()
        _write_ply(d / f"cloud_bin_{k}.ply", pts)
        small = O.resize_bilinear(np.divide(img8, 255, dtype=np.float32), 120, 160)           # util/uio.py:33-40
        frags[k] = (pts.astype(np.float64), np.transpose(small, (2, 0, 1))[None].copy())
    ckpt = tmp_path / "ckpt.pth"
    torch.save({"state_dict": seeded_sd, "config": dict(Config(voxel_size=0.05)), "epoch": 1}, ckpt)
    gd.main(["--source", str(src), "--target", str(dst), "-m", str(ckpt)])
    for k, (pts, img) in frags.items():
        out = np.load(dst / "scene" / "seq-01" / f"cloud_bin_{k}.npz")
        xyz_ref, F_ref = O.extract_features(seeded_sd, pts, 0.05, img)
        assert (out["points"] == pts).all() and (out["xyz"] == xyz_ref).all()
        assert np.abs(out["feature"] - F_ref.numpy()).max() < 1e-4


def test_extract_features_with_rgb_and_checks(clouds, images, seeded_sd):
    """util/misc.py:48-79: optional rgb input (3 channels) and the argument checks."""
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    sd = O.seeded_state_dict(seed=3, in_channels=3, with_unused_image_layers=True)
    m = load_model("ResUNetBN2C")(3, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    xyz = clouds[0][::9].astype(np.float64)
    rgb = np.random.default_rng(0).random((len(xyz), 3))
    with torch.no_grad():
        xd, F = extract_features(m, xyz, rgb=rgb, voxel_size=0.05, device=torch.device("cuda:0"), image=images[0])
    coords, inds = O.voxelize(xyz, 0.05)
    Fr = O.resunet_forward(sd, coords, images[0], feats=(rgb - 0.5)[inds].astype(np.float32))
    assert (xd == xyz[inds]).all() and (F.cpu() - Fr).abs().max() < 1e-4
    with pytest.raises(ValueError):
        extract_features(m, xyz, rgb=rgb * 3, voxel_size=0.05, device=torch.device("cuda:0"), image=images[0])


def test_large_fragment_200k_voxels(clouds, images, seeded_sd):
    """BASELINE.json configs[4] shape: a ~200k-voxel fragment (the fixture scaled x3.4 @ 2.5 cm is
    geometrically the 30 cm KITTI case scaled down).  GPU vs oracle (C geometry + torch-CPU convs)."""
    import imf_oracle_cbind as OC
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    xyz = clouds[0].astype(np.float64) * 3.4
    with torch.no_grad():
        xd, F = extract_features(m, xyz, voxel_size=0.025, device=torch.device("cuda:0"), skip_check=True,
                                 image=images[0])
    coords, inds = OC.voxelize(xyz, 0.025)
    assert len(coords) > 150_000 and F.shape == (len(coords), 32)
    assert (xd == xyz[inds]).all()
    Fr = O.resunet_forward(seeded_sd, coords, images[0], geometry=OC.Geometry(coords))
    assert (F.cpu() - Fr).abs().max() < 1e-4


def test_bench_workload_pair_capacity_mode_vs_oracle(seeded_sd):
    """THE bench.py workload itself (BASELINE.json configs[1] at the size the metric is quoted on): the S50k fragment
    pair (cloud_bin_0 + cloud_bin_1 scaled x1.7 @ 2.5 cm, 103 k voxels) as ONE batched forward in capacity mode --
    built by bench.py's own Workload class, so exactly the launches the headline times -- against the oracle
    (C geometry + torch-CPU convolutions, each fragment with its own image): voxel sets / order exact, descriptors to
    the north_star's 1e-4, capacity mode == exact mode bit for bit."""
    import bench
    import imf_oracle_cbind as OC
    dev = torch.device("cuda:0")
    model, sd = bench.build_model(dev)
    pts, imgs = bench.load_pair(1.7)
    wl = bench.Workload(model, dev, pts, imgs, 0.025)
    with torch.no_grad():
        F_exact = wl.prepare_graph().clone()
        wl.runner.use_graph = False
        res = wl.graph_step()
        torch.cuda.synchronize()
        assert res.flags == 0
        F = res.F.clone()
    assert torch.equal(F, F_exact)
    row0 = 0
    for k in (0, 1):
        coords, inds = OC.voxelize(pts[k], 0.025)
        m = len(coords)
        Fr = O.resunet_forward(sd, coords, imgs[k:k + 1], geometry=OC.Geometry(coords))
        assert (F[row0:row0 + m].cpu() - Fr).abs().max() < 1e-4
        row0 += m
    assert row0 == F.shape[0] == res.counts[0] and row0 > 100_000


def test_bench_pipelined_steps_equal_the_one_bucket_steps():
    """bench.py's pipelined mode (two capacity buckets, a step's head on the side stream under the previous step's decoder --
    imf_fragment_io.head_on_side with a reuse event, the order the streaming pipeline uses): every step of a back-to-back
    run returns the exact path's descriptors bit for bit, from both buckets, with a different replica of the points each
    step, and traced / one-bucket steps can follow pipelined ones on the same workload."""
    import bench
    dev = torch.device("cuda:0")
    model, sd = bench.build_model(dev)
    pts, imgs = bench.load_pair(1.0)
    wl = bench.Workload(model, dev, pts, imgs, 0.025)
    with torch.no_grad():
        F_exact = wl.prepare_graph(replicate=True, pipelined=True).clone()
        wl.runner.use_graph = False
        assert len(wl.buckets) == 2 and wl.buckets[0] is not wl.buckets[1]
        wl.pipelined = True
        outs = [wl.graph_step() for _ in range(9)]
        torch.cuda.synchronize()
        assert {id(r.bucket) for r in outs[-2:]} == {id(b) for b in wl.buckets}       # both buckets took steps
        for r in outs[-2:]:
            assert r.flags == 0 and torch.equal(r.F, F_exact)
        tr = []
        r = wl.graph_step(tr)                                   # a traced step (bucket 0, head on the main stream) right behind
        torch.cuda.synchronize()
        assert len(tr) > 15 and torch.equal(r.F, F_exact)
        wl.pipelined = False
        r = wl.graph_step()
        torch.cuda.synchronize()
        assert torch.equal(r.F, F_exact)


def test_extract_features_stream_equals_extract_features(clouds, images, seeded_sd):
    """extract_features_stream (pinned staging, several forwards in flight through the capacity buckets) returns, in order,
    what extract_features returns fragment by fragment -- for fragments of different sizes (different buckets), float32 and
    float64 points, and from a cold runner (first fragment on the exact path): exactly with one fragment per forward
    (batch=1); with two per forward (the default) the voxels exactly and the descriptors to rounding (2e-6: a row's partial
    sums are grouped by its tile's active offsets, and its tile-mates differ in a batch)."""
    from imfnet_amd.extract import extract_features, extract_features_stream
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    dev = torch.device("cuda:0")
    frs = []
    for i, (k, sc, dt) in enumerate([(0, 1.0, np.float64), (1, 1.0, np.float64), (0, 1.3, np.float64), (1, 1.0, np.float32),
                                     (0, 1.0, np.float64), (1, 1.3, np.float64), (0, 1.3, np.float64)]):
        frs.append(((clouds[k].astype(np.float64) * sc).astype(dt), images[k]))
    got = list(extract_features_stream(m, iter(frs), 0.05, dev, depth=3, batch=1))
    assert len(got) == len(frs)
    with torch.no_grad():
        for (xyz, img), (xd, F) in zip(frs, got):
            xr, Fr = extract_features(m, xyz, voxel_size=0.05, device=dev, skip_check=True, image=img)
            assert xd.dtype == np.float64 and F.dtype == np.float32
            assert xd.shape == xr.shape and (xd == xr).all()
            assert (F == Fr.cpu().numpy()).all()
    again = list(extract_features_stream(m, iter(frs[:3]), 0.05, dev, depth=2, batch=1))   # warm runner: all through the buckets
    for a, b in zip(again, got[:3]):
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    runner = m.fragment_runner()
    before = dict(runner.stats)
    for rep in range(2):                               # the first pass may size the pair buckets' grids (observe_batch)
        pairs = list(extract_features_stream(m, iter(frs), 0.05, dev, depth=3))            # two fragments per forward
    assert len(pairs) == len(frs)
    for a, b in zip(pairs, got):
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all()
        assert np.abs(a[1] - b[1]).max() < 2e-6
    assert runner.stats["eager"] + runner.stats["graph"] > before["eager"] + before["graph"]   # (went through the buckets)
    views = list(extract_features_stream(m, iter(frs[:4]), 0.05, dev, depth=2, copy=False))
    assert len(views) == 4


def kitti_like_cloud(n_points=2_000_000, seed=0):
    """seeded union of 64 random planar patches in a 120 m cube, 2 cm jitter (LiDAR-like surfaces
    at KITTI extents), tuned to ~200 k voxels at 0.3 m."""
    rng = np.random.default_rng(seed)
    per = n_points // 64
    parts = []
    for _ in range(64):
        c = rng.uniform(-60, 60, 3)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        v = np.cross(u, rng.normal(size=3)); v /= np.linalg.norm(v)
        ext = rng.uniform(5, 12, 2)
        ab = rng.uniform(-1, 1, (per, 2)) * ext
        parts.append(c + ab[:, :1] * u + ab[:, 1:] * v + rng.normal(0, 0.02, (per, 3)))
    return np.clip(np.concatenate(parts, 0), -60, 60)


def test_kitti_like_fragment_voxel_30cm(seeded_sd):
    """BASELINE.json configs[4]: KITTI extents, voxel 0.3 m (config_kitti.py:118), ~200 k voxels, random 120x160 image.
    Exact voxel set / order vs the C oracle, descriptors vs the oracle to 1e-4; second call through the capacity-mode
    runner bit-identical."""
    import imf_oracle_cbind as OC
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    xyz = kitti_like_cloud()
    img = np.random.default_rng(0).random((1, 3, 120, 160)).astype(np.float32)
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    with torch.no_grad():
        xd, F = extract_features(m, xyz, voxel_size=0.3, device=torch.device("cuda:0"), skip_check=True, image=img)
        xd2, F2 = extract_features(m, xyz, voxel_size=0.3, device=torch.device("cuda:0"), skip_check=True, image=img)
    coords, inds = OC.voxelize(xyz, 0.3)
    assert 120_000 < len(coords) < 400_000 and F.shape == (len(coords), 32)
    assert (xd == xyz[inds]).all() and (xd2 == xd).all() and torch.equal(F, F2)
    assert m.fragment_runner().stats["eager"] + m.fragment_runner().stats["graph"] >= 1
    Fr = O.resunet_forward(seeded_sd, coords, img, geometry=OC.Geometry(coords))
    assert (F.cpu() - Fr).abs().max() < 1e-4


def test_extract_features_numpy_points_with_a_device_image(clouds, images, seeded_sd):
    """util/misc.py:97 takes a host point array next to a device image (torch.as_tensor); on a warm runner this
    mix goes through the device-staging branch of the capacity path."""
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    model_s = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    model_s.load_state_dict(seeded_sd, strict=True)
    model_s = model_s.eval().cuda()
    dev = torch.device("cuda:0")
    xyz = clouds[0][::3].astype(np.float64)
    with torch.no_grad():
        xd0, F0 = extract_features(model_s, xyz, voxel_size=0.05, device=dev, skip_check=True, image=images[0])
        xd1, F1 = extract_features(model_s, xyz, voxel_size=0.05, device=dev, skip_check=True,
                                   image=torch.as_tensor(images[0]).to(dev))      # warm runner, CUDA image
        xd2, F2 = extract_features(model_s, torch.as_tensor(xyz).to(dev), voxel_size=0.05, device=dev, skip_check=True,
                                   image=images[0])                                # CUDA points, host image
    assert (xd0 == xd1).all() and (xd0 == xd2).all()
    assert torch.equal(F0, F1) and torch.equal(F0, F2)


@pytest.mark.parametrize("sdma,head", [(False, True), (True, True), (True, False)])
def test_pipeline_transfer_modes_agree(clouds, images, seeded_sd, sdma, head):
    """(head: a job's table reset / level-0 pyramid / image fork on the side stream, under the previous job's last
    convolutions -- imf_fragment_io.head_on_side -- or on the main stream.)  The streaming pipeline's two transfer mechanisms -- copy kernels that address the pinned blocks directly (the
    fallback when the HIP runtime was started without ROC_CPU_WAIT_FOR_SIGNAL=0) and the copy engines (hipMemcpyAsync) --
    deliver the same bytes: xyz_down and descriptors of streamed fragments equal the per-fragment calls', for float64
    points that are float32 values (uploaded narrowed), arbitrary float64 and float32 inputs."""
    from imfnet_amd.extract import extract_features, extract_features_stream
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    dev = torch.device("cuda:0")
    frs = [(clouds[0][::3].astype(np.float64), images[0]),                 # float32-valued float64: narrowed on upload
           (clouds[1][::3].astype(np.float64) * 1.1, images[1]),           # arbitrary float64
           (clouds[0][1::4].copy(), images[0]),                            # float32
           (clouds[1][2::3].astype(np.float64), images[1])]
    with torch.no_grad():
        ref = [extract_features(m, x, voxel_size=0.05, device=dev, skip_check=True, image=i) for x, i in frs]
        ref = [(xd, F.cpu().numpy()) for xd, F in ref]
        runner = m.fragment_runner()
        runner._streamers.pop(dev, None)                                   # (a streamer of the default mode may exist already)
        st = runner.streamer(dev, sdma_copies=sdma, head_on_side=head)
        assert st.sdma_copies == sdma and st.head_on_side == head
        got = list(extract_features_stream(m, iter(frs), 0.05, dev, depth=2, batch=1))
        got += list(extract_features_stream(m, iter(frs), 0.05, dev, depth=3, batch=1))
    assert runner.stats["eager"] >= 6                  # (a fragment that outgrows its bucket is redone exactly: also equal)
    for (xd, F), (xr, Fr) in zip(got, ref + ref):
        assert (xd == xr).all() and xd.dtype == np.float64
        assert (F == Fr).all()


def test_direct_launches_after_a_stream_keep_the_main_stream_order(clouds, images, seeded_sd):
    """The streaming pipeline runs a job's upload and head on the image / side streams (imf_fragment_io.head_on_side), which
    do not wait for the main stream; the direct capacity-mode launches (device tensors in) follow the main stream's order
    only.  Round 4 had them SHARE the lane-0 bucket of a capacity key (a pipeline job after an unsynchronised
    direct launch could overwrite its inputs); since round 5 the streamer owns lanes 1 .. n of a key and the direct path
    lane 0, so the two never meet on one bucket.  Checked: after a stream pass and ten back-to-back direct launches (no
    synchronisation in between) the direct bucket has seen exactly the direct launches, carries head_on_side = 0 and no
    stale events, the streamer's lanes are other objects, and every result equals the first."""
    from imfnet_amd.extract import extract_features, extract_features_stream
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    dev = torch.device("cuda:0")
    xyz = clouds[0].astype(np.float64) * 1.3
    with torch.no_grad():
        xd0, F0 = extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=images[0])
        F0 = F0.cpu()
        for _ in extract_features_stream(m, ((xyz, images[0]) for _ in range(6)), 0.025, dev, batch=1):
            pass
        runner = m.fragment_runner()
        assert runner.streamer(dev).head_on_side
        xyz_d, img_d = torch.as_tensor(xyz).to(dev), torch.as_tensor(images[0]).to(dev)
        eager0 = runner.stats["eager"]
        outs = [extract_features(m, xyz_d, voxel_size=0.025, device=dev, skip_check=True, image=img_d)[1] for _ in range(10)]
        torch.cuda.synchronize()
    assert runner.stats["eager"] - eager0 == 10                      # capacity-mode launches on the direct bucket
    key = runner.caps_for(len(xyz), 1, images[0].shape[2], images[0].shape[3], 0.025, True)
    b = runner.buckets[key]                                          # lane 0: the direct launches' own bucket
    assert b.launches == 10 and b.io.head_on_side == 0 and not b.io.inputs_event and not b.io.reuse_event
    st = runner.streamer(dev)
    lanes = [runner.buckets[(k, lane)] for k in st._made for lane in sorted(st._made[k])]
    assert lanes and all(l is not b for l in lanes) and sum(l.launches for l in lanes) >= 6
    assert all(torch.equal(F.cpu(), F0) for F in outs)
    assert m.take_flags(dev) == 0


def test_extract_features_can_leave_the_descriptors_on_the_device(clouds, images, seeded_sd):
    """util/misc.py:100-104 returns F as a device tensor; extract_features(host_descriptors=False) does exactly that -- the
    download stops in front of the descriptor block (xyz_down + counts only cross PCIe) -- and the rows equal the default
    call's, on the device and in its pinned host copy."""
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    dev = torch.device("cuda:0")
    xyz = clouds[1].astype(np.float64) * 1.2
    with torch.no_grad():
        extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=images[1])     # exact path: teaches the runner
        xd1, F1 = extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=images[1])
        host1 = np.array(F1.host)
        xd2, F2 = extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=images[1], host_descriptors=False)
    assert hasattr(F1, "host") and not hasattr(F2, "host") and F2.is_cuda
    assert (xd1 == xd2).all() and torch.equal(F1, F2) and (F2.cpu().numpy() == host1).all()
    assert m.fragment_runner().stats["eager"] >= 2
