"""Run-to-run reproducibility of extract_features with a FRESH model per iteration (plans, packs, image plan rebuilt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd.extract import extract_features
from imfnet_amd.model import load_model
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests/golden/fixture_clouds.npz"))
img = np.transpose(np.load(os.path.join(ROOT, "tests/golden/fixture_images.npz"))["image_0"], (2, 0, 1))[None].copy()
scale = float(os.environ.get("SCALE", "3.4"))
xyz = z["cloud_bin_0"].astype(np.float64) * scale
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
F0 = None
for it in range(int(os.environ.get("N", "12"))):
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(sd, strict=True); m = m.eval().cuda()
    with torch.no_grad():
        xd, F = extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=img)
        xd, Fb = extract_features(m, xyz, voxel_size=0.025, device=dev, skip_check=True, image=img)
    torch.cuda.synchronize()
    if F0 is None: F0 = F.clone()
    print(it, "first-call diff vs run0 %.3e   second call (runner) diff %.3e  mode %s" % (float((F - F0).abs().max()), float((Fb - F0).abs().max()), m.image_branch_mode), flush=True)
