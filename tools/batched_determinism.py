import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import imf_oracle as O
if os.environ.get('DET'): torch.backends.cudnn.deterministic = True
from imfnet_amd.extract import extract_features_batch
from imfnet_amd.model import load_model
z = np.load("/root/repo/tests/golden/fixture_clouds.npz"); im = np.load("/root/repo/tests/golden/fixture_images.npz")
pts = [z["cloud_bin_0"].astype(np.float64), z["cloud_bin_1"].astype(np.float64)]
imgs = np.stack([np.transpose(im[f"image_{i}"], (2, 0, 1)) for i in (0, 1)]).copy()
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().cuda()
res = {}
with torch.no_grad():
    for mode in ("native", "native", "py", "py", "native"):
        if mode == "py": os.environ["IMFNET_PYTHON_EXECUTOR"] = "1"
        else: os.environ.pop("IMFNET_PYTHON_EXECUTOR", None)
        out = extract_features_batch(model, pts, float(sys.argv[1]), "cuda", imgs)
        res.setdefault(mode, []).append(torch.cat([o[1] for o in out]).clone())
a = res["native"][0]
for m, lst in res.items():
    for i, t in enumerate(lst):
        print(m, i, float((t - a).abs().max()))
# which stage varies?  image branch (K/V of both images) across replays
with torch.no_grad():
    img_d = torch.as_tensor(imgs).cuda()
    ks = []
    for _ in range(4):
        model.start_image_branch(img_d, inputs_ready=True)
        _, feat, kv, ev, packed = model._pending_image
        torch.cuda.synchronize()
        ks.append((feat.clone(), kv.clone()))
    for i in range(1, 4):
        print("image branch replay", i, float((ks[i][0] - ks[0][0]).abs().max()), float((ks[i][1] - ks[0][1]).abs().max()))
