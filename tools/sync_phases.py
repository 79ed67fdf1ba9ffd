#!/usr/bin/env python3
"""The synchronous extract_features call (SURVEY 8d's span, one S50k fragment) against its parts timed alone: staging copy /
narrow into the pinned block, H2D, the forward, D2H.  usage: python tools/sync_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from imfnet_amd import _lib
from imfnet_amd.extract import extract_features
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev)
xyz, img, voxel = bench.load_workload(1.7, 0.025)
xyz = xyz.astype(np.float64)
xyz32v = xyz.astype(np.float32).astype(np.float64)          # float32-valued, as a PLY's points
L = _lib.lib()


def med(fn, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


pin64 = torch.empty(xyz.shape, dtype=torch.float64).pin_memory()
pin32 = torch.empty(xyz.shape, dtype=torch.float32).pin_memory()
d64 = torch.empty(xyz.shape, dtype=torch.float64, device=dev)
d32 = torch.empty(xyz.shape, dtype=torch.float32, device=dev)
n64, n32 = pin64.numpy(), pin32.numpy()
print("points %d: %.1f MB as float64" % (len(xyz), xyz.nbytes / 1e6))
print("stage  np.copyto float64 -> pinned            %.3f ms" % med(lambda: np.copyto(n64, xyz)))
print("stage  imf_host_narrow_points f64 -> f32 pinned %.3f ms (rc %d)" % (med(lambda: L.imf_host_narrow_points(xyz32v.ctypes.data, xyz32v.size, n32.ctypes.data)),
                                                                           L.imf_host_narrow_points(xyz32v.ctypes.data, xyz32v.size, n32.ctypes.data)))
def h2d(dst, src):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
print("H2D    float64 points                         %.3f ms" % med(lambda: h2d(d64, pin64)))
print("H2D    float32 points                         %.3f ms" % med(lambda: h2d(d32, pin32)))
Fd = torch.empty((52000, 32), dtype=torch.float32, device=dev); Fh = torch.empty((52000, 32), dtype=torch.float32).pin_memory()
xd, xh = torch.empty((52000, 3), dtype=torch.float64, device=dev), torch.empty((52000, 3), dtype=torch.float64).pin_memory()
def d2h():
    Fh.copy_(Fd, non_blocking=True); xh.copy_(xd, non_blocking=True); torch.cuda.synchronize()
print("D2H    descriptors + xyz_down                 %.3f ms" % med(d2h))
with torch.no_grad():
    for pts, name in ((xyz, "arbitrary float64"), (xyz32v, "float32-valued float64")):
        for _ in range(4):
            extract_features(model, pts, voxel_size=voxel, device=dev, skip_check=True, image=img)
        print("extract_features, %-24s host F    %.3f ms" % (name, med(lambda: extract_features(model, pts, voxel_size=voxel, device=dev, skip_check=True, image=img))))
        print("extract_features, %-24s device F  %.3f ms" % (name, med(lambda: extract_features(model, pts, voxel_size=voxel, device=dev, skip_check=True, image=img, host_descriptors=False))))
    pd, idv = torch.as_tensor(xyz).to(dev), torch.as_tensor(img).to(dev)
    for _ in range(4):
        extract_features(model, pd, voxel_size=voxel, device=dev, skip_check=True, image=idv)
    def devcall():
        extract_features(model, pd, voxel_size=voxel, device=dev, skip_check=True, image=idv); torch.cuda.synchronize()
    print("extract_features, device-resident inputs, F on the device, + synchronize   %.3f ms" % med(devcall))
st = {k: v for k, v in model.fragment_runner().stats.items()}
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if not isinstance(v, list)})
