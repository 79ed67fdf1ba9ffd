"""Time the descriptor-matching row at the evaluation's size: 5 000 x 5 000 x 32,
both directions + mutual check + inlier count, on the GPU; the oracle's brute force and an exact
KD-tree (what the reference uses, per-row queries) on the host beside it."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "oracle")
sys.path.insert(0, ".")
import imf_oracle as O  # noqa: E402
from imfnet_amd.matching import mutual_inliers, nn_search  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(0)
d1 = rng.standard_normal((n, 32)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
d2 = (d1[rng.permutation(n)] + 0.05 * rng.standard_normal((n, 32))).astype(np.float32)
d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
k1 = rng.uniform(-1, 1, (n, 3)); k2 = rng.uniform(-1, 1, (n, 3))
a, b = torch.as_tensor(d1).cuda(), torch.as_tensor(d2).cuda()
ka, kb = torch.as_tensor(k1).cuda(), torch.as_tensor(k2).cuda()


def pair():
    nn21 = nn_search(b, a)
    nn12 = nn_search(a, b)
    return mutual_inliers(nn21, nn12, ka, kb, np.eye(4), 0.1)


for _ in range(3):
    pair()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
t0 = time.perf_counter()
e0.record()
for _ in range(reps):
    nn21 = nn_search(b, a)
    nn12 = nn_search(a, b)
e1.record()
torch.cuda.synchronize()
gpu_search_ms = e0.elapsed_time(e1) / reps
t0 = time.perf_counter()
for _ in range(reps):
    pair()
wall_ms = (time.perf_counter() - t0) / reps * 1e3

t0 = time.perf_counter(); r21 = O.knn_search(d2, d1); r12 = O.knn_search(d1, d2); cpu_bf = time.perf_counter() - t0
from scipy.spatial import cKDTree  # noqa: E402
t0 = time.perf_counter()
kd = cKDTree(d1.astype(np.float64)).query(d2.astype(np.float64), k=1)[1]
kd2 = cKDTree(d2.astype(np.float64)).query(d1.astype(np.float64), k=1)[1]
cpu_kd = time.perf_counter() - t0
assert (nn21.cpu().numpy() == r21).all() and (nn12.cpu().numpy() == r12).all() and (kd == r21).all()
# RANSAC at the evaluation's size: n correspondences (35 % correct), 50 000 hypotheses
from imfnet_amd.matching import ransac_registration  # noqa: E402
q, _ = np.linalg.qr(rng.standard_normal((3, 3))); q *= np.sign(np.linalg.det(q))
src = rng.uniform(-1.5, 1.5, (n, 3)); dst = src @ q.T + [0.3, -0.2, 0.5] + rng.normal(0, 0.01, (n, 3))
cor = np.arange(n, dtype=np.int32); bad = rng.random(n) < 0.65; cor[bad] = rng.integers(0, n, bad.sum())
srcd, dstd, cord = torch.as_tensor(src).cuda(), torch.as_tensor(dst).cuda(), torch.as_tensor(cor).cuda()
for _ in range(2):
    got = ransac_registration(srcd, dstd, cord, 3, 0.075, 0.9, 50000, 1)
t0 = time.perf_counter()
for _ in range(10):
    got = ransac_registration(srcd, dstd, cord, 3, 0.075, 0.9, 50000, 1)
ransac_gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter(); ref = O.ransac_registration(src, dst, cor, 3, 0.075, 0.9, 50000, 1); ransac_cpu_s = time.perf_counter() - t0
assert got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3]
flops = 2 * 2.0 * n * n * 32
print(json.dumps({"n": n, "gpu_two_searches_ms": round(gpu_search_ms, 4),
                  "gpu_fp64_tflops": round(flops / gpu_search_ms / 1e9, 2),
                  "pair_wall_ms_incl_host_sync": round(wall_ms, 4),
                  "cpu_oracle_bruteforce_s": round(cpu_bf, 3), "cpu_exact_kdtree_s": round(cpu_kd, 3),
                  "pairs_per_s_gpu": round(1e3 / wall_ms, 1),
                  "ransac_50k_hypotheses_gpu_ms": round(ransac_gpu_ms, 3), "ransac_valid_hypotheses": got[3],
                  "ransac_oracle_numpy_s": round(ransac_cpu_s, 3)}))
