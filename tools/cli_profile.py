"""cProfile of the generate_desc batch loop (workers=0, stored NPZ) + runner statistics."""
import cProfile, pstats, os, sys, tempfile, shutil, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from PIL import Image
from imfnet_amd import dataio, generate_desc as gd
from imfnet_amd.checkpoint import Config
from imfnet_amd.model import load_model
from emulate_3dmatch import write_ply
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz")); im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
root = tempfile.mkdtemp(prefix="imf_cli_"); src = os.path.join(root, "src", "scene", "seq-01"); os.makedirs(src)
rng = np.random.default_rng(0)
for k in range(n):
    write_ply(os.path.join(src, f"cloud_bin_{k}.ply"), z[f"cloud_bin_{k % 2}"] * rng.uniform(1.0, 1.9))
    Image.fromarray(np.clip(np.rint(im[f"image_{k % 2}"] * 255), 0, 255).astype(np.uint8)).save(os.path.join(src, f"cloud_bin_{k}_0.png"))
cfg = Config(); torch.manual_seed(0)
model = load_model(cfg.model)(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=cfg).eval().cuda()
dev = torch.device("cuda:0"); dataio.NPZ_LEVEL = 0
with torch.no_grad():
    gd.extract_features_batch(model, cfg, os.path.join(root, "src"), os.path.join(root, "warm"), cfg.voxel_size, dev, workers=0)
    r = model.fragment_runner(); print("after warm pass:", r.stats, "buckets", len(r.buckets))
    pr = cProfile.Profile(); pr.enable()
    gd.extract_features_batch(model, cfg, os.path.join(root, "src"), os.path.join(root, "d"), cfg.voxel_size, dev, workers=0)
    pr.disable()
    print("after timed pass:", r.stats, "buckets", len(r.buckets))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
shutil.rmtree(root)
