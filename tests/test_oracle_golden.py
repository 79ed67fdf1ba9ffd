"""CPU tests: the oracle (oracle/imf_oracle.py) against every golden vector the reference offers
for this path and against outputs of the reference's own model code
(tests/golden/gen_golden.py).  No GPU, no /root/reference at run time."""
import json
import os

import numpy as np
import torch

import imf_oracle as O
from conftest import GOLDEN


def test_voxelize_matches_reference_head_map(clouds, head_map):
    """files/3D_head_map.ply == xyz[inds] of sparse_quantize(floor(xyz/0.025)) -- the one result of
    the path the reference itself pins (first-occurrence order, fp64 division)."""
    xyz = clouds[0].astype(np.float64)
    coords, inds = O.voxelize(xyz, 0.025)
    assert coords.shape == (18977, 4) and coords.dtype == np.int32
    assert (np.diff(inds) > 0).all()
    assert (xyz[inds].astype(np.float32) == head_map).all()
    # fp32 arithmetic is NOT equivalent: guards against "optimising" the division
    c32 = np.floor(clouds[0] / np.float32(0.025)).astype(np.int64)
    assert (c32 != np.floor(xyz / 0.025).astype(np.int64)).any()


def test_voxel_counts_of_survey(clouds):
    for i, vs, m in ((0, 0.05, 5182), (1, 0.05, 5140), (0, 0.025, 18977), (1, 0.025, 19082)):
        assert len(O.voxelize(clouds[i].astype(np.float64), vs)[0]) == m
    assert len(O.voxelize(clouds[0].astype(np.float64) * 1.7, 0.025)[0]) == 51232


def test_pyramid_and_rulebook_counts_of_survey(clouds):
    """SURVEY Appendix C pair counts (S5)."""
    coords, _ = O.voxelize(clouds[0].astype(np.float64), 0.05)
    g = O.Geometry(coords)
    assert [len(l) for l in g.levels] == [5182, 1453, 413, 112]
    assert (g.k_first >= 0).sum() == 208506
    assert [(r >= 0).sum() for r in g.k3] == [71848, 20117, 5803, 1544]
    assert [(r >= 0).sum() for r in g.down] == [14244, 4077, 1141]
    # transposed maps are the forward maps with in/out swapped (same pair multiset)
    for lv in range(3):
        fwd = {(int(r), o, k) for o, row in enumerate(g.down[lv]) for k, r in enumerate(row) if r >= 0}
        tr = {(f, int(c), k) for f, row in enumerate(g.up[lv]) for k, c in enumerate(row) if c >= 0}
        assert fwd == tr


def test_kernel_offset_order():
    o = O.kernel_offsets(3)
    assert o[0].tolist() == [-1, -1, -1] and o[1].tolist() == [0, -1, -1] and o[13].tolist() == [0, 0, 0]
    assert o[3].tolist() == [-1, 0, -1] and o[9].tolist() == [-1, -1, 0]
    assert len(O.kernel_offsets(5)) == 125


def test_downsample_floor_semantics():
    c = np.array([[0, -1, -2, -3], [0, 1, 2, 3], [0, -4, 0, 5], [0, -1, -1, -3]], np.int32)
    coarse, parent = O.downsample(c, 2)
    assert coarse.tolist() == [[0, -2, -2, -4], [0, 0, 2, 2], [0, -4, 0, 4]]
    assert parent.tolist() == [0, 1, 2, 0]


def test_restatement_matches_reference_wiring_S5(clouds, images, golden, seeded_sd):
    """Full forward of the restatement vs the reference's model/*.py run verbatim (golden)."""
    xyz = clouds[0].astype(np.float64)
    xyz_down, F = O.extract_features(seeded_sd, xyz, 0.05, images[0])
    assert F.shape == (5182, 32)
    assert np.abs(F.numpy() - golden["S5_F"]).max() < 2e-6
    assert (xyz_down.astype(np.float32) == golden["S5_xyz_down_f32"]).all()
    assert np.allclose(np.linalg.norm(F.numpy(), axis=1), 1.0, atol=1e-5)


def test_restatement_attention_and_image_encoder(golden, images, seeded_sd):
    sd = {k: torch.as_tensor(v) for k, v in seeded_sd.items()}
    out = O.attention_fusion(torch.as_tensor(golden["af_ctx"]), torch.as_tensor(golden["af_in"]), sd)
    assert np.abs(out.numpy() - golden["af_out"]).max() < 2e-5
    img = O.image_encoder(images[0], sd)
    assert img.shape == (1, 128, 15, 20)
    assert np.abs(img.numpy() - golden["img_out"]).max() < 1e-5


def test_state_dict_schema_matches_reference(seeded_sd):
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_schema.json")))
    assert len(ref) == 361
    assert set(seeded_sd) == set(ref)
    assert all(list(seeded_sd[k].shape) == ref[k] for k in ref)


def test_resize_is_centre_average_for_4x(golden):
    """640x480 -> 160x120 INTER_LINEAR == mean of the centre 2x2 of each 4x4 block."""
    rng = np.random.default_rng(0)
    img = rng.random((480, 640, 3), dtype=np.float32)
    out = O.resize_bilinear(img, 120, 160)
    ref = 0.25 * (img[1::4, 1::4] + img[1::4, 2::4] + img[2::4, 1::4] + img[2::4, 2::4])
    assert np.abs(out - ref).max() < 1e-6


def test_fnv_hash_known_answer():
    # FNV-1a-64 of the single column value 0 and of (1,2,3), by hand
    h0 = (14695981039346656037 * 1099511628211) % 2 ** 64
    assert int(O.fnv_hash_vec(np.array([[0]]))[0]) == h0
    h = 14695981039346656037
    for v in (1, 2, 3):
        h = ((h * 1099511628211) % 2 ** 64) ^ v
    assert int(O.fnv_hash_vec(np.array([[1, 2, 3]]))[0]) == h
    neg = O.fnv_hash_vec(np.array([[-1, 0, 5]]))
    assert neg.dtype == np.uint64


# ---- descriptor matching ----------------------------------------------------------
def test_knn_restatement_equals_independent_kdtree():
    """The reference's KD-tree (Open3D) is absent; an exact 1-NN is unique up to exact ties, so the
    oracle's brute force must equal scipy's exact KD-tree."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    a = rng.standard_normal((700, 32)).astype(np.float32)
    b = rng.standard_normal((900, 32)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    assert (O.knn_search(b, a) == cKDTree(a.astype(np.float64)).query(b.astype(np.float64), k=1)[1]).all()
    assert O.knn_search(b, a).dtype == np.int32


def test_mutual_match_and_inlier_ratio_hand_case():
    nn21 = np.array([2, 0, 0, 1], dtype=np.int32)                # frag2 -> frag1
    nn12 = np.array([1, 3, 0], dtype=np.int32)                   # frag1 -> frag2
    assert list(O.mutual_match_indices(nn21, nn12)) == [0, 1, 3]
    T = np.eye(4)
    T[:3, 3] = [1.0, 0.0, 0.0]
    assert np.allclose(O.transform_points(np.array([[0.0, 2.0, 3.0]]), T), [[1.0, 2.0, 3.0]])
    T[3, 3] = 2.0
    assert np.allclose(O.transform_points(np.array([[0.0, 2.0, 3.0]]), T), [[0.5, 1.0, 1.5]])
    d = np.eye(32, dtype=np.float32)[:3]
    k1 = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=np.float64)
    k2 = k1 + np.array([[0.0, 0.05, 0], [0.0, 0.2, 0], [0.0, 0.099, 0]])
    n_inl, ratio, m2, nn = O.feature_match(k1, d, k2, d, np.eye(4), 0.1)
    assert (n_inl, list(m2), list(nn)) == (2, [0, 1, 2], [0, 1, 2]) and abs(ratio - 2 / 3) < 1e-15


def test_select_keypoints_restatement_hand_case():
    """scripts/evaluation_3dmatch.py:162-171 -- FNV keys of floor(p / voxel), membership, ascending."""
    c = np.array([[-0.05, 0.0, 0.05], [-1e-9, -0.05 - 1e-9, 0.1], [0.049999, -0.1, 0.0]])
    s = np.array([[-0.04, -0.09, 0.14], [-0.01, 0.01, 0.09]])
    assert list(O.select_keypoints(s, c, 0.05)) == [0, 1]
    assert O.fnv_hash_vec(np.array([[-1.0, 0.0, 1.0]]))[0] == O.fnv_hash_vec(np.array([[2 ** 64 - 1, 0, 1]], dtype=np.uint64))[0]


# ---- RANSAC registration ----------------------------------------------------------
def test_ransac_restatement_pieces():
    """Known-answer for the shared generator (published splitmix64 vector), the rigid fit against a
    constructed motion (also a reflection-prone planar sample), and the error metrics of util/uio.py."""
    assert int(O._splitmix64(0)) == 0xE220A8397B1DCDAF and int(O._splitmix64(0x9E3779B97F4A7C15)) == 0x6E789E6AA1B965F4
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    t = np.array([0.3, -0.2, 0.5])
    s = rng.standard_normal((6, 3, 3))
    T = O.rigid_fit(s, s @ q.T + t)
    assert np.abs(T[:, :3, :3] - q).max() < 1e-12 and np.abs(T[:, :3, 3] - t).max() < 1e-12
    assert np.allclose(np.linalg.det(T[:, :3, :3]), 1.0)
    Tg = np.eye(4)
    Tg[:3, :3], Tg[:3, 3] = q, t
    rre0, rte0 = O.compute_registration_error(Tg, Tg)
    assert rre0 < 1e-4 and rte0 == 0.0                     # arccos of 1 - 1e-16
    assert abs(O.compute_transform_error(Tg, np.eye(6), Tg)) < 1e-24
    Te = Tg.copy()
    Te[:3, 3] += [0.03, 0.0, 0.04]
    rre, rte = O.compute_registration_error(Tg, Te)
    assert rre < 1e-4 and abs(rte - 0.05) < 1e-12


def test_ransac_restatement_recovers_known_motion():
    rng = np.random.default_rng(1)
    n = 1500
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    Tg = np.eye(4)
    Tg[:3, :3], Tg[:3, 3] = q, [0.4, -0.1, 0.2]
    src = rng.uniform(-1.5, 1.5, (n, 3))
    dst = src @ q.T + Tg[:3, 3] + rng.normal(0, 0.005, (n, 3))
    corres = np.arange(n)
    bad = rng.random(n) < 0.6
    corres[bad] = rng.integers(0, n, bad.sum())
    T, it, inl, nvalid, fit, rmse = O.ransac_registration(src, dst, corres, 3, 0.075, 0.9, 20000, seed=3)
    rre, rte = O.compute_registration_error(Tg, T)
    assert it >= 0 and inl >= 0.35 * n and nvalid > 50 and rre < 3.0 and rte < 0.05 and rmse < 0.075
    # no hypothesis can survive when every correspondence is wrong by metres
    T0, it0, inl0, *_ = O.ransac_registration(src, dst + 50 * rng.standard_normal((n, 3)), corres, 3, 0.075, 0.9, 2000, seed=3)
    assert it0 == -1 and inl0 == 0 and (T0 == np.eye(4)).all()


def test_cpu_conv_twin_matches_the_torch_restatement(clouds):
    """imf_cpu_spconv_fwd (C / OpenMP, the reported CPU baseline's convolution) == the torch gather-GEMM-scatter
    restatement, for k3, k1 (no table) and a strided map, and through the whole network."""
    import imf_oracle_cbind as OC
    xyz = clouds[0][::5].astype(np.float64)
    coords, _ = OC.voxelize(xyz, 0.05)
    g = OC.Geometry(coords)
    gen = torch.Generator().manual_seed(0)
    for nbr, cin, cout, kvol in ((g.k3[0], 32, 64, 27), (None, 96, 64, 1), (g.down[0], 32, 64, 27)):
        n_in = len(g.levels[0])
        f = torch.randn(n_in, cin, generator=gen)
        w = torch.randn(kvol, cin, cout, generator=gen) * 0.1
        ref = O.spconv(f, w if kvol > 1 else w[0], nbr)
        got = OC.spconv(f, w if kvol > 1 else w[0], nbr)
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
    img = np.random.default_rng(0).random((1, 3, 120, 160)).astype(np.float32)
    F_t = O.resunet_forward(sd, coords, img, geometry=g)
    O.SPCONV_IMPL = "c"
    try:
        F_c = O.resunet_forward(sd, coords, img, geometry=g)
    finally:
        O.SPCONV_IMPL = "torch"
    assert float((F_t - F_c).abs().max()) < 2e-6


def test_product_side_seeded_weights_equal_the_oracles():
    """bench.py and the tools take their seeded weights from imfnet_amd/seeded.py (the product side never imports the
    oracle); the checks take theirs from the oracle: "seed s" must be the same network on both sides."""
    import torch
    from imfnet_amd.seeded import seeded_state_dict
    for kw in (dict(seed=0, with_unused_image_layers=True), dict(seed=3, conv1_kernel_size=3)):
        a, b = O.seeded_state_dict(**kw), seeded_state_dict(**kw)
        assert list(a) == list(b)
        assert all(a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) for k in a)


def test_native_png_decode_and_resize_on_the_reference_image(images):
    """The reference's own 480x640 image (files/cloud_bin_0_0.png) through the native PNG decoder and bilinear resize
    (csrc/codecs.hip) -- the path scripts/generate_desc.py:92-97 + util/uio.py:33-40 take for every fragment: the decoded
    pixels equal matplotlib's at the committed sample grid (`png0_rows` = imread(...)[::60, ::80], tests/golden/gen_golden.py),
    and the resized image equals the committed 120x160 fixture (O.resize_bilinear of matplotlib's decode).  The PNG itself
    lives in the reference tree: where that is absent (the GPU box) only the committed samples' consistency is checked."""
    z = np.load(os.path.join(GOLDEN, "fixture_images.npz"))
    rows = z["png0_rows"]
    assert rows.shape == (8, 8, 3) and rows.dtype == np.float32 and 0.0 <= rows.min() and rows.max() <= 1.0
    assert (np.rint(rows * 255) / 255 - rows).__abs__().max() < 1e-7          # 8-bit PNG values / 255, as matplotlib returns them
    path = "/root/reference/files/cloud_bin_0_0.png"
    if not os.path.exists(path):
        pytest.skip("the reference's PNG is not on this machine")
    from imfnet_amd.dataio import process_image, read_image, image_to_nchw
    img = read_image(path)
    assert img.dtype == np.float32 and img.shape == (480, 640, 3)
    assert (img[::60, ::80] == rows).all()
    small = process_image(image=img, aim_H=120, aim_W=160)
    assert small.shape == (120, 160, 3)
    assert np.abs(small - z["image_0"]).max() < 2e-7
    assert np.abs(image_to_nchw(small) - images[0]).max() < 2e-7
