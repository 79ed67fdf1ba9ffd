"""End-to-end rate of the generate_desc batch loop (PLY + PNG decode -> GPU -> NPZ write) on a synthetic 3DMatch-layout
tree, for several loader / writer thread counts and NPZ compression levels; model build excluded.
usage: python tools/cli_throughput.py [n_fragments]"""
import json, os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from PIL import Image
from imfnet_amd import dataio, generate_desc as gd
from imfnet_amd.checkpoint import Config
from imfnet_amd.model import load_model
from emulate_3dmatch import write_ply

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
root = tempfile.mkdtemp(prefix="imf_cli_")
src = os.path.join(root, "src", "scene", "seq-01")
os.makedirs(src)
rng = np.random.default_rng(0)
big = os.environ.get("BIG_IMAGES")                       # 640x480 colour images as in the data set (resized on load)
for k in range(n):
    write_ply(os.path.join(src, f"cloud_bin_{k}.ply"), z[f"cloud_bin_{k % 2}"] * rng.uniform(1.0, 1.9))   # 19 k .. 64 k voxels @ 2.5 cm
    img = np.clip(np.rint(im[f"image_{k % 2}"] * 255), 0, 255).astype(np.uint8)
    if big:
        img = np.kron(img, np.ones((4, 4, 1), dtype=np.uint8))
    Image.fromarray(img).save(os.path.join(src, f"cloud_bin_{k}_0.png"))
cfg = Config()
torch.manual_seed(0)
model = load_model(cfg.model)(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=cfg).eval().cuda()
dev = torch.device("cuda:0")
res = {}
with torch.no_grad():
    gd.extract_features_batch(model, cfg, os.path.join(root, "src"), os.path.join(root, "warm"), cfg.voxel_size, dev, workers=8)
    for level, workers, threads, bp in ((1, 0, 1, 0), (1, 8, 1, 0), (1, 8, 8, 0), (1, 16, 8, 0), (1, 16, 16, 0), (1, 16, 8, -1), (6, 16, 16, 0), (0, 8, 1, 0), (0, 16, 1, 0), (0, 16, 1, -1)):
        dataio.NPZ_LEVEL, dataio.NPZ_THREADS = level, threads
        dst = os.path.join(root, f"dst{level}_{workers}_{threads}" + ("_grouped" if bp else ""))
        t = time.time()
        times, _ = gd.extract_features_batch(model, cfg, os.path.join(root, "src"), dst, cfg.voxel_size, dev, workers=workers, batch_points=bp)
        dt = time.time() - t
        res[f"npz_level={level} workers={workers} npz_threads={threads}" + (" grouped forwards" if bp else "")] = {
            "fragments_per_s": round(n / dt, 1), "wall_s": round(dt, 2), "gpu_ms_per_fragment": round(float(np.mean(times)) * 1e3, 3)}
a = np.load(os.path.join(root, "dst1_0_1", "scene", "seq-01", "cloud_bin_3.npz"))
b = np.load(os.path.join(root, "dst0_16_1", "scene", "seq-01", "cloud_bin_3.npz"))
c = np.load(os.path.join(root, "dst1_16_16", "scene", "seq-01", "cloud_bin_3.npz"))
assert all((a[k] == c[k]).all() for k in ("points", "xyz", "feature"))
assert open(os.path.join(root, "dst1_0_1", "scene", "seq-01", "cloud_bin_3.npz"), "rb").read() == \
       open(os.path.join(root, "dst1_16_16", "scene", "seq-01", "cloud_bin_3.npz"), "rb").read(), "bytes depend on the thread count"
g = np.load(os.path.join(root, "dst1_16_8_grouped", "scene", "seq-01", "cloud_bin_3.npz"))
assert all((a[k] == g[k]).all() for k in ("points", "xyz")) and np.abs(a["feature"] - g["feature"]).max() < 2e-6
same = all((a[k] == b[k]).all() for k in ("points", "xyz", "feature"))
print(json.dumps({"fragments": n, "images": "640x480" if big else "160x120", "identical_arrays_across_settings": bool(same), **res}, indent=1))
shutil.rmtree(root)
