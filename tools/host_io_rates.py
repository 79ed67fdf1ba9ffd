"""Rates of the host-side stages of generate_desc on this box, each alone: PLY + PNG decode, NPZ write (level x threads x
concurrent writers), so that the pipelined loop's fragments/s can be set against its parts.
usage: python tools/host_io_rates.py"""
import os, sys, time, tempfile, shutil
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from PIL import Image
from imfnet_amd import dataio
from emulate_3dmatch import write_ply
z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
root = tempfile.mkdtemp(prefix="imf_io_")
n = 64
for k in range(n):
    write_ply(os.path.join(root, f"c{k}.ply"), z[f"cloud_bin_{k % 2}"] * 1.5)
    Image.fromarray(np.clip(np.rint(im[f"image_{k % 2}"] * 255), 0, 255).astype(np.uint8)).save(os.path.join(root, f"c{k}_0.png"))
print("cpus", os.cpu_count())
def load(k):
    return dataio.read_ply_points(os.path.join(root, f"c{k}.ply")), dataio.read_image(os.path.join(root, f"c{k}_0.png"))
for w in (1, 8, 16, 32):
    with ThreadPoolExecutor(w) as ex:
        t = time.time(); list(ex.map(load, range(n))); dt = time.time() - t
    print("decode PLY+PNG: %2d threads %7.1f fragments/s" % (w, n / dt))
pts = z["cloud_bin_0"].astype(np.float64) * 1.5
rng = np.random.default_rng(0)
M = 45000
xyz = pts[:M].copy(); F = rng.normal(size=(M, 32)).astype(np.float32); F /= np.linalg.norm(F, axis=1, keepdims=True)
def write(args):
    k, level, th = args
    dataio.save_npz(os.path.join(root, f"o{k}.npz"), level=level, threads=th, points=pts, xyz=xyz, feature=F)
for level, w, th in ((0, 8, 1), (1, 1, 1), (1, 1, 8), (1, 1, 32), (1, 8, 8), (1, 16, 8), (1, 16, 16), (1, 32, 8), (1, 64, 4), (1, 128, 1), (6, 16, 16)):
    with ThreadPoolExecutor(w) as ex:
        t = time.time(); list(ex.map(write, [(k, level, th) for k in range(n)])); dt = time.time() - t
    print("NPZ write level %d: %3d writers x %2d deflate threads %7.1f files/s (%.1f ms per file per writer)" % (level, w, th, n / dt, dt / n * w * 1e3))
shutil.rmtree(root)
