"""End-to-end rate of the generate_desc CLI (decode -> GPU -> zlib write) on a synthetic 3DMatch-layout
tree, with and without the loader / writer threads.  usage: python tools/cli_throughput.py [n_fragments]"""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from imfnet_amd import generate_desc as gd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
root = tempfile.mkdtemp(prefix="imf_cli_")
src = os.path.join(root, "src", "scene", "seq-01")
os.makedirs(src)
rng = np.random.default_rng(0)
for k in range(n):
    pts = (z[f"cloud_bin_{k % 2}"] * rng.uniform(1.0, 1.9)).astype("<f4")      # 19 k .. 64 k voxels @ 2.5 cm
    with open(os.path.join(src, f"cloud_bin_{k}.ply"), "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                b"property float z\nend_header\n" % len(pts))
        f.write(pts.tobytes())
    Image.fromarray(np.clip(np.rint(im[f"image_{k % 2}"] * 255), 0, 255).astype(np.uint8)).save(
        os.path.join(src, f"cloud_bin_{k}_0.png"))
res = {}
for workers in (0, 4, 16):
    dst = os.path.join(root, f"dst{workers}")
    t = time.time()
    gd.main(["--source", os.path.join(root, "src"), "--target", dst, "--seeded_weights", "0", "--workers", str(workers)])
    res[workers] = time.time() - t
    nd = sum(np.load(os.path.join(dst, "scene", "seq-01", f)).get("feature").shape[0] for f in os.listdir(os.path.join(dst, "scene", "seq-01")))
print({f"workers={w}": f"{s:.2f} s total (incl. model build), {n / s:.1f} fragments/s" for w, s in res.items()}, "descriptors:", nd)
shutil.rmtree(root)
