"""Descriptor matching for the feature-match-recall evaluation, host side.

Mirrors the reference's names: `knn_search` is util/uio.py:245-258, the mutual check and the inlier
ratio are scripts/evaluation_3dmatch.py:207-234.  Everything runs in libimfnet_hip.so
(`imf_nn_search`, `imf_mutual_inliers`); there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ImfError, check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _descs(x, device, name):
    t = torch.as_tensor(x)
    if t.dim() != 2:
        raise ImfError(f"{name} must be [n, dim], got {tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32).contiguous()


def nn_search(query, db, return_dist2=False):
    """Device tensors in, device tensors out: nn int32 [n_query] (and squared fp64 distances)."""
    if query.shape[1] != db.shape[1]:
        raise ImfError(f"descriptor widths differ: {query.shape[1]} vs {db.shape[1]}")
    if db.shape[0] == 0:
        raise ImfError("knn_search on an empty destination set")
    nq, nd, dim = query.shape[0], db.shape[0], query.shape[1]
    dev = query.device
    L = _lib.lib()
    ws_bytes = L.imf_nn_workspace_bytes(nq, nd)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nn = torch.empty(nq, dtype=torch.int32, device=dev)
    d2 = torch.empty(nq, dtype=torch.float64, device=dev) if return_dist2 else None
    check(L.imf_nn_search(query.data_ptr(), nq, db.data_ptr(), nd, dim, nn.data_ptr(),
                          d2.data_ptr() if return_dist2 else None, ws.data_ptr(), ws_bytes, _stream()),
          "imf_nn_search")
    return (nn, d2) if return_dist2 else nn


def knn_search(points_src, points_dst, k=1, device="cuda"):
    """util/uio.py:245-258: for every row of points_src the index of its nearest row of points_dst
    (exact, fp64 distances).  Returns int32 numpy [len(points_src)]; only k=1 (the only value the
    evaluation uses, scripts/evaluation_3dmatch.py:207-210)."""
    if k != 1:
        raise NotImplementedError("knn_search: the evaluation path uses k=1 only")
    nn = nn_search(_descs(points_src, device, "points_src"), _descs(points_dst, device, "points_dst"))
    return nn.cpu().numpy()


def select_keypoints(sample_points, coords, voxel_size, device="cuda"):
    """scripts/evaluation_3dmatch.py:162-171: indices (ascending, int64 numpy like `np.where`) of the
    rows of `coords` (the descriptor file's `xyz`) whose voxel key occurs among the voxel keys of
    `sample_points` (the randomly drawn raw points)."""
    s = torch.as_tensor(sample_points).to(device=device, dtype=torch.float64).contiguous()
    c = torch.as_tensor(coords).to(device=device, dtype=torch.float64).contiguous()
    if s.dim() != 2 or c.dim() != 2 or s.shape[1] != 3 or c.shape[1] != 3:
        raise ImfError(f"expected [n,3] arrays, got {tuple(s.shape)} and {tuple(c.shape)}")
    ns, m = s.shape[0], c.shape[0]
    L = _lib.lib()
    ws_bytes = L.imf_keypoint_workspace_bytes(ns, m)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s.device)
    inds = torch.empty(max(m, 1), dtype=torch.int32, device=s.device)
    count = torch.zeros(1, dtype=torch.int32, device=s.device)
    check(L.imf_select_keypoints(s.data_ptr(), ns, c.data_ptr(), m, float(voxel_size), inds.data_ptr(),
                                 count.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "imf_select_keypoints")
    return inds[:int(count.item())].cpu().numpy().astype(np.int64)


def mutual_inliers(nn21, nn12, kpts1=None, kpts2=None, pose=None, inlier_thresh=0.1):
    """scripts/evaluation_3dmatch.py:212-233 on device tensors.  Returns (frag2_match_indices int32
    device tensor, n_matches, n_inliers); n_inliers is 0 when no geometry is given."""
    dev = nn21.device
    n2, n1 = nn21.shape[0], nn12.shape[0]
    match2 = torch.empty(max(n2, 1), dtype=torch.int32, device=dev)
    meta = torch.zeros(2, dtype=torch.int32, device=dev)
    geo = kpts1 is not None and kpts2 is not None and pose is not None
    if geo:
        kpts1 = torch.as_tensor(kpts1).to(device=dev, dtype=torch.float64).contiguous()
        kpts2 = torch.as_tensor(kpts2).to(device=dev, dtype=torch.float64).contiguous()
        if kpts1.shape != (n1, 3) or kpts2.shape != (n2, 3):
            raise ImfError(f"keypoints must be [{n1},3] and [{n2},3], got {tuple(kpts1.shape)}, {tuple(kpts2.shape)}")
        T = np.ascontiguousarray(np.asarray(pose, dtype=np.float64).reshape(4, 4))
        pose_p = T.ctypes.data_as(C.c_void_p)
    check(_lib.lib().imf_mutual_inliers(nn21.data_ptr(), n2, nn12.data_ptr(), n1,
                                        kpts1.data_ptr() if geo else None, kpts2.data_ptr() if geo else None,
                                        pose_p if geo else None, float(inlier_thresh), match2.data_ptr(),
                                        meta.data_ptr(), _stream()), "imf_mutual_inliers")
    n_matches, n_inliers = meta.tolist()
    return match2[:n_matches], n_matches, n_inliers


def feature_match(frag1_kpts, frag1_descs, frag2_kpts, frag2_descs, gt_pose, inlier_thresh=0.1,
                  device="cuda"):
    """The FMR part of `register_fragment_pair` (scripts/evaluation_3dmatch.py:207-234): both
    nearest-neighbour searches, the mutual check, the ground-truth transform and the inlier count.
    Returns (num_inliers, inlier_ratio, frag2_match_indices, frag21_nnindices); inlier_ratio is nan
    when there is no mutual match (the reference divides 0 by 0)."""
    d1 = _descs(frag1_descs, device, "frag1_descs")
    d2 = _descs(frag2_descs, device, "frag2_descs")
    nn21 = nn_search(d2, d1)
    nn12 = nn_search(d1, d2)
    match2, n_matches, n_inliers = mutual_inliers(nn21, nn12, frag1_kpts, frag2_kpts, gt_pose, inlier_thresh)
    ratio = n_inliers / n_matches if n_matches else float("nan")
    return n_inliers, ratio, match2.cpu().numpy(), nn21.cpu().numpy()


def ransac_registration(src, dst, corres, ransac_n=3, max_corr_dist=0.075, edge_similarity=0.9, max_iter=50000,
                        seed=0, device="cuda"):
    """Device RANSAC on given correspondences (imf_ransac_registration).  Returns (T 4x4 numpy
    source->target, winning iteration, inliers, hypotheses that passed the checkers, fitness, rmse)."""
    s = torch.as_tensor(src).to(device=device, dtype=torch.float64).contiguous()
    d = torch.as_tensor(dst).to(device=device, dtype=torch.float64).contiguous()
    c = torch.as_tensor(corres).to(device=s.device, dtype=torch.int32).contiguous()
    if s.dim() != 2 or s.shape[1] != 3 or d.dim() != 2 or d.shape[1] != 3 or c.shape != (s.shape[0],):
        raise ImfError("ransac_registration: src [n,3], dst [m,3], corres [n]")
    L = _lib.lib()
    nbytes = L.imf_ransac_workspace_bytes(int(max_iter))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    T = torch.empty(16, dtype=torch.float64, device=s.device)
    meta = torch.empty(3, dtype=torch.int32, device=s.device)
    stats = torch.empty(2, dtype=torch.float64, device=s.device)
    check(L.imf_ransac_registration(s.data_ptr(), s.shape[0], d.data_ptr(), d.shape[0], c.data_ptr(), int(ransac_n),
                                    float(max_corr_dist), float(edge_similarity), int(max_iter), int(seed),
                                    T.data_ptr(), meta.data_ptr(), stats.data_ptr(), ws.data_ptr(), nbytes, _stream()),
          "imf_ransac_registration")
    it, inl, nvalid = meta.tolist()
    fit, rmse = stats.tolist()
    return T.cpu().numpy().reshape(4, 4), it, inl, nvalid, fit, rmse


def run_ransac(xyz0, xyz1, feat0, feat1, voxel_size, ransac_n=4, seed=0, device="cuda"):
    """scripts/benchmark_util.py:16-34: feature correspondences (nearest xyz1 feature of every xyz0
    point), then RANSAC with the edge-length (0.9) and distance (1.5 voxel) checkers, 50 000
    hypotheses.  Returns the 4x4 transformation xyz0 -> xyz1 like `result_ransac.transformation`."""
    f0, f1 = _descs(feat0, device, "feat0"), _descs(feat1, device, "feat1")
    corres = nn_search(f0, f1)
    return ransac_registration(xyz0, xyz1, corres, ransac_n=ransac_n, max_corr_dist=voxel_size * 1.5,
                               edge_similarity=0.9, max_iter=50000, seed=seed, device=device)[0]
