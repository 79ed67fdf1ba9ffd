"""GPU parity of the descriptor-matching row: imf_nn_search / imf_mutual_inliers
through the C ABI against the oracle's restatement of util/uio.py:245-258 and
scripts/evaluation_3dmatch.py:207-234.  Index work: bit-exact."""
import json

import numpy as np
import pytest
import torch

import imf_oracle as O

pytestmark = pytest.mark.gpu


def _descs(rng, n, dim=32):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)          # L2-normalised like the model's output


def _rigid(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    T = np.eye(4)
    T[:3, :3] = q
    T[:3, 3] = rng.uniform(-2, 2, 3)
    return T


@pytest.mark.parametrize("nq,nd,dim", [(1, 1, 32), (5, 3, 32), (16, 16, 32), (17, 63, 32), (64, 65, 32),
                                       (129, 4097, 32), (1000, 777, 32), (333, 500, 16), (200, 321, 64)])
def test_nn_search_matches_oracle(nq, nd, dim):
    from imfnet_amd.matching import knn_search, nn_search
    rng = np.random.default_rng(nq * 131 + nd)
    q, d = _descs(rng, nq, dim), _descs(rng, nd, dim)
    ref = O.knn_search(q, d)
    got = knn_search(q, d)
    assert got.dtype == np.int32 and got.shape == (nq,)
    assert (got == ref).all()
    nn, d2 = nn_search(torch.as_tensor(q).cuda(), torch.as_tensor(d).cuda(), return_dist2=True)
    exact = ((q.astype(np.float64) - d.astype(np.float64)[ref]) ** 2).sum(1)
    assert np.abs(d2.cpu().numpy() - exact).max() < 1e-12


def test_nn_search_unnormalised_and_scipy_kdtree():
    """Arbitrary (not unit-length) descriptors; cross-check with an independent exact KD-tree."""
    from scipy.spatial import cKDTree
    from imfnet_amd.matching import knn_search
    rng = np.random.default_rng(5)
    q = (rng.standard_normal((700, 32)) * rng.uniform(0.1, 30, (700, 1))).astype(np.float32)
    d = (rng.standard_normal((900, 32)) * rng.uniform(0.1, 30, (900, 1))).astype(np.float32)
    got = knn_search(q, d)
    assert (got == cKDTree(d.astype(np.float64)).query(q.astype(np.float64), k=1)[1]).all()
    assert (got == O.knn_search(q, d)).all()


def test_nn_search_ties_take_lowest_index():
    from imfnet_amd.matching import knn_search
    rng = np.random.default_rng(9)
    base = _descs(rng, 150)
    d = np.concatenate([base, base[::-1], base[:40]], 0)         # every row appears 2-3 times
    q = np.concatenate([base[10:90], _descs(rng, 33)], 0)
    got = knn_search(q, d)
    assert (got == O.knn_search(q, d)).all()
    assert (got[:80] == np.arange(10, 90)).all()                 # exact duplicates: first occurrence


def test_nn_search_rejects_bad_arguments():
    from imfnet_amd.matching import knn_search, nn_search
    from imfnet_amd._lib import ImfError
    with pytest.raises(ImfError):
        knn_search(np.zeros((4, 32), np.float32), np.zeros((0, 32), np.float32))
    with pytest.raises(ImfError):
        knn_search(np.zeros((4, 32), np.float32), np.zeros((4, 16), np.float32))
    with pytest.raises(ImfError):
        knn_search(np.zeros((4, 24), np.float32), np.zeros((4, 24), np.float32))   # width not in {16,32,64}
    with pytest.raises(NotImplementedError):
        knn_search(np.zeros((4, 32), np.float32), np.zeros((4, 32), np.float32), k=2)
    empty = nn_search(torch.zeros((0, 32), device="cuda"), torch.zeros((3, 32), device="cuda"))
    assert empty.shape == (0,)


@pytest.mark.parametrize("n1,n2,noise", [(400, 300, 0.02), (1500, 2100, 0.05), (5000, 5000, 0.08)])
def test_feature_match_matches_oracle(n1, n2, noise):
    """A synthetic overlapping pair: frag2's descriptors / keypoints are noisy, transformed copies of
    part of frag1's plus distractors; ground-truth pose known."""
    from imfnet_amd.matching import feature_match
    rng = np.random.default_rng(n1 + n2)
    T = _rigid(rng)
    k1 = rng.uniform(-1.5, 1.5, (n1, 3))
    d1 = _descs(rng, n1)
    n_shared = min(n1, n2) * 2 // 3
    pick = rng.permutation(n1)[:n_shared]
    d2 = np.concatenate([d1[pick] + noise * rng.standard_normal((n_shared, 32)).astype(np.float32),
                         _descs(rng, n2 - n_shared)], 0).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    k2_in_1 = np.concatenate([k1[pick] + rng.normal(0, 0.06, (n_shared, 3)), rng.uniform(-1.5, 1.5, (n2 - n_shared, 3))], 0)
    Tinv = np.linalg.inv(T)
    k2 = k2_in_1 @ Tinv[:3, :3].T + Tinv[:3, 3]
    perm = rng.permutation(n2)
    d2, k2 = d2[perm], k2[perm]
    ref = O.feature_match(k1, d1, k2, d2, T, 0.1)
    got = feature_match(k1, d1, k2, d2, T, 0.1)
    assert (got[3] == ref[3]).all()                              # frag21_nnindices
    assert (got[2] == ref[2]).all() and got[2].dtype == np.int32  # frag2_match_indices, ascending
    assert got[0] == ref[0] and got[1] == ref[1]
    assert 0 < got[0] < len(got[2])                              # the case exercises both outcomes


def test_mutual_inliers_edge_cases():
    from imfnet_amd.matching import feature_match, mutual_inliers
    # no mutual match at all: a 3-cycle nn21 = [1,2,0], nn12 = [1,2,0]
    nn21 = torch.tensor([1, 2, 0], dtype=torch.int32, device="cuda")
    nn12 = torch.tensor([1, 2, 0], dtype=torch.int32, device="cuda")
    m, n_matches, n_inl = mutual_inliers(nn21, nn12)
    assert n_matches == 0 and n_inl == 0 and m.shape == (0,)
    # single keypoint each side: always mutual; inlier iff within the threshold after the pose
    T = np.eye(4)
    T[:3, 3] = [0.05, 0, 0]
    one = np.ones((1, 32), np.float32)
    n_inl, ratio, m2, nn = feature_match(np.zeros((1, 3)), one, np.zeros((1, 3)), one, T, 0.1)
    assert (n_inl, ratio, list(m2), list(nn)) == (1, 1.0, [0], [0])
    n_inl, ratio, _, _ = feature_match(np.zeros((1, 3)), one, np.zeros((1, 3)), one, T, 0.05)   # strict <
    assert (n_inl, ratio) == (0, 0.0)
    # projective row is honoured (divide by w), as Open3D's transform does
    T2 = np.eye(4)
    T2[3, 3] = 2.0
    k = np.array([[0.3, 0.0, 0.0]])
    assert feature_match(k / 2, one, k, one, T2, 1e-9)[0] == 1


def test_self_match_properties_full_size():
    """Size-independent properties at the evaluation's full size (5 000 keypoints): matching a set
    against itself is the identity, every keypoint is mutual, distances are ~0."""
    from imfnet_amd.matching import mutual_inliers, nn_search
    rng = np.random.default_rng(77)
    d = torch.as_tensor(_descs(rng, 5000)).cuda()
    nn, d2 = nn_search(d, d, return_dist2=True)
    assert torch.equal(nn.cpu(), torch.arange(5000, dtype=torch.int32))
    assert float(d2.max()) < 1e-12
    m, n_matches, _ = mutual_inliers(nn, nn)
    assert n_matches == 5000 and torch.equal(m.cpu(), torch.arange(5000, dtype=torch.int32))


def test_fixture_pair_descriptors_match(clouds, images, seeded_sd):
    """End to end on the in-tree pair: descriptors of both fixture fragments from the HIP path
    (5 cm), then matching on the GPU vs the oracle on the same descriptors."""
    from imfnet_amd.extract import extract_features
    from imfnet_amd.matching import feature_match
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    out = []
    for i in (0, 1):
        xyz, F = extract_features(m, xyz=clouds[i].astype(np.float64), voxel_size=0.05, device="cuda",
                                  skip_check=True, image=torch.as_tensor(images[i]))
        out.append((xyz, F.cpu().numpy()))
    (k1, d1), (k2, d2) = out
    T = np.eye(4)                                               # both fragments share the scene frame here
    ref = O.feature_match(k1, d1, k2, d2, T, 0.1)
    got = feature_match(k1, d1, k2, d2, T, 0.1)
    assert (got[3] == ref[3]).all() and (got[2] == ref[2]).all() and got[:2] == ref[:2]


# ---- keypoint -> voxel selection --------------------------------------------------
@pytest.mark.parametrize("voxel,n_keypoints", [(0.05, 5000), (0.025, 5000), (0.025, 300), (0.1, 100000)])
def test_select_keypoints_matches_oracle(clouds, voxel, n_keypoints):
    """The evaluator's selection on the fixture fragment: xyz_down from the voxeliser, 5 000 random raw
    points (scripts/evaluation_3dmatch.py:154-171)."""
    from imfnet_amd.matching import select_keypoints
    pts = clouds[0].astype(np.float64)
    _, inds = O.voxelize(pts, voxel)
    xyz_down = pts[inds]
    rng = np.random.RandomState(3)
    pick = rng.choice(len(pts), min(len(pts), n_keypoints), replace=False)
    ref = O.select_keypoints(pts[pick], xyz_down, voxel)
    got = select_keypoints(pts[pick], xyz_down, voxel)
    assert got.dtype == np.int64 and (got == ref).all()
    assert 0 < len(got) <= min(n_keypoints, len(xyz_down))


def test_select_keypoints_edge_cases():
    from imfnet_amd.matching import select_keypoints
    rng = np.random.default_rng(0)
    coords = rng.uniform(-3, 3, (1500, 3))
    assert len(select_keypoints(np.zeros((0, 3)), coords, 0.05)) == 0          # no samples
    assert len(select_keypoints(coords[:5], np.zeros((0, 3)), 0.05)) == 0      # no voxels
    got = select_keypoints(coords, coords, 0.05)                               # everything selected
    assert (got == np.arange(1500)).all()
    far = coords + 100.0
    assert len(select_keypoints(far, coords, 0.05)) == 0                       # disjoint
    # negative coordinates and exact voxel boundaries hash like numpy's float -> uint64 cast
    c = np.array([[-0.05, 0.0, 0.05], [-1e-9, -0.05 - 1e-9, 0.1], [0.049999, -0.1, 0.0]])
    s = np.array([[-0.01, 0.01, 0.09], [-0.04, -0.09, 0.14]])
    assert (select_keypoints(s, c, 0.05) == O.select_keypoints(s, c, 0.05)).all()
    assert list(O.select_keypoints(s, c, 0.05)) == [0, 1]


# ---- RANSAC registration ------------------------------------------------------------
def _ransac_case(rng, n, outlier_frac, noise):
    T = _rigid(rng)
    src = rng.uniform(-1.5, 1.5, (n, 3))
    dst = (src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, noise, (n, 3)))
    perm = rng.permutation(n)
    dst = dst[perm]
    corres = np.argsort(perm).astype(np.int32)
    bad = rng.random(n) < outlier_frac
    corres[bad] = rng.integers(0, n, bad.sum())
    return src, dst, corres, T


@pytest.mark.parametrize("n,ransac_n,iters,seed", [(300, 3, 4000, 0), (2000, 3, 50000, 1), (5000, 3, 50000, 7),
                                                   (1200, 4, 20000, 2)])
def test_ransac_matches_oracle(n, ransac_n, iters, seed):
    """Same draws, same checkers, same scoring as the restatement: the winning hypothesis, its inlier
    count and the number of hypotheses that pass the checkers are exact; the transformation to 1e-9."""
    from imfnet_amd.matching import ransac_registration
    rng = np.random.default_rng(100 + n)
    src, dst, corres, Tg = _ransac_case(rng, n, 0.65, 0.01)
    ref = O.ransac_registration(src, dst, corres, ransac_n, 0.075, 0.9, iters, seed)
    got = ransac_registration(src, dst, corres, ransac_n, 0.075, 0.9, iters, seed)
    assert got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3]
    assert np.abs(got[0] - ref[0]).max() < 1e-9
    assert abs(got[4] - ref[4]) < 1e-15 and abs(got[5] - ref[5]) < 1e-12
    rre, rte = O.compute_registration_error(Tg, got[0])
    assert rre < 3.0 and rte < 0.06


def test_ransac_edge_cases_and_run_ransac():
    from imfnet_amd.matching import ransac_registration, run_ransac
    rng = np.random.default_rng(5)
    src, dst, corres, Tg = _ransac_case(rng, 800, 0.5, 0.005)
    # nothing survives: identity, iteration -1 (Open3D's default RegistrationResult)
    T, it, inl, nvalid, fit, rmse = ransac_registration(src, dst + 100.0 * rng.standard_normal(dst.shape), corres,
                                                        3, 0.075, 0.9, 3000, 0)
    assert it == -1 and inl == 0 and fit == 0.0 and (T == np.eye(4)).all()
    # determinism, and a different seed is a different draw sequence
    a = ransac_registration(src, dst, corres, 3, 0.075, 0.9, 8000, 11)
    b = ransac_registration(src, dst, corres, 3, 0.075, 0.9, 8000, 11)
    c = ransac_registration(src, dst, corres, 3, 0.075, 0.9, 8000, 12)
    assert a[1:] == b[1:] and (a[0] == b[0]).all() and a[1] != c[1]
    # the reference's call shape: features in, transformation out (benchmark_util.py:16-34)
    feat1 = _descs(rng, 800)
    feat0 = feat1[corres] + 0.01 * rng.standard_normal((800, 32)).astype(np.float32)
    T = run_ransac(src, dst, feat0, feat1, 0.05, ransac_n=3)
    rre, rte = O.compute_registration_error(Tg, T)
    assert T.shape == (4, 4) and rre < 3.0 and rte < 0.06


# ---- scene-level evaluator on the in-tree pair (scripts/evaluation_3dmatch.py flow) ---------------------
def test_evaluator_on_the_redkitchen_pair(tmp_path, clouds, images, seeded_sd):
    """Descriptor files of the two in-tree fragments (7-scenes-redkitchen 0 and 1) + their ground-truth pose
    and covariance from the benchmark -> `imfnet_amd.evaluate`; every number of the result line is
    re-derived with the oracle from the same files."""
    import os
    from imfnet_amd import evaluate as E
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model
    gt = np.load(os.path.join(os.path.dirname(__file__), "golden", "redkitchen_pair_0_1_gt.npz"))
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m = m.eval().cuda()
    desc = tmp_path / "desc" / "7-scenes-redkitchen" / "seq-01"
    bench = tmp_path / "bench" / "7-scenes-redkitchen"
    desc.mkdir(parents=True)
    bench.mkdir(parents=True)
    data = {}
    for k in (0, 1):
        pts = clouds[k].astype(np.float64)
        xyz, F = extract_features(m, xyz=pts, voxel_size=0.05, device="cuda", skip_check=True,
                                  image=torch.as_tensor(images[k]))
        data[k] = dict(points=pts, xyz=xyz, feature=F.cpu().numpy())
        np.savez(desc / f"cloud_bin_{k}.npz", **data[k])
    with open(bench / "gt.log", "w") as fh:
        fh.write("0\t 1\t 60\t\n" + "".join("\t".join(f"{v:.8e}" for v in row) + "\n" for row in gt["pose"]))
    with open(bench / "gt.info", "w") as fh:
        fh.write("0\t 1\t 60\t\n" + "".join("\t".join(f"{v:.8e}" for v in row) + "\n" for row in gt["covariance"]))
    out = tmp_path / "out"
    assert E.main(["--desc_root", str(tmp_path / "desc"), "--benchmark_root", str(tmp_path / "bench"),
                   "--out_root", str(out), "--voxel_size", "0.05", "--num_rand_keypoints", "2000", "--seed", "4"]) == 0
    line = open(out / "IMFNet" / "7-scenes-redkitchen-seq-01-0.10.txt").read().split()
    assert line[:2] == ["cloud_bin_0", "cloud_bin_1"] and line[4] == "1"
    # oracle re-derivation from the same files and the cached keypoint draw
    kp = np.load(out / "IMFNet_keypoints" / "7-scenes-redkitchen_seq-01_0_1_keypoints.npz")
    pose = E.read_log(bench / "gt.log")[0].transformation
    cov = E.read_info_file(bench / "gt.info")[0]["covariance"]
    sel = [O.select_keypoints(data[k]["points"][kp["inds_i" if k == 0 else "inds_j"]], data[k]["xyz"], 0.05) for k in (0, 1)]
    k1, d1 = data[0]["xyz"][sel[0]], data[0]["feature"][sel[0]]
    k2, d2 = data[1]["xyz"][sel[1]], data[1]["feature"][sel[1]]
    n_inl, ratio, _, _ = O.feature_match(k1, d1, k2, d2, pose, 0.1)
    assert int(line[2]) == n_inl and abs(float(line[3]) - ratio) < 1e-8
    if len(k1) < len(k2):
        trans = O.ransac_registration(k1, k2, O.knn_search(d1, d2), 3, 0.075, 0.9, 50000, 4)[0]
    else:
        trans = np.linalg.inv(O.ransac_registration(k2, k1, O.knn_search(d2, d1), 3, 0.075, 0.9, 50000, 4)[0])
    es_T = np.linalg.inv(trans)
    accepted = O.compute_transform_error(pose, cov, es_T) < 0.04
    assert int(line[5]) == int(accepted)
    if accepted:
        rre, rte = O.compute_registration_error(pose, es_T)
        assert abs(float(line[6]) - rre) < 1e-6 and abs(float(line[7]) - rte) < 1e-8
    summary = json.load(open(out / "IMFNet-metrics-0.10.json"))
    assert summary["scenes"]["7-scenes-redkitchen"]["pairs"] == 1
