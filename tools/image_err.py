"""Error of the native image trunk (variants 6 / 0) and of torch-on-GPU against golden img_out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd.model import load_model
from imfnet_amd.model.image_plan import ImagePlan
dev = "cuda:0"
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
m.load_state_dict(sd, strict=True); m = m.eval().to(dev)
g = dict(np.load(os.path.join(ROOT, "tests/golden/golden_descriptors.npz")))
img = np.transpose(np.load(os.path.join(ROOT, "tests/golden/fixture_images.npz"))["image_0"], (2, 0, 1))[None].copy()
x = torch.as_tensor(img).to(dev)
with torch.no_grad():
    ref = m.img_encoder(x)
    ref64 = m.img_encoder.double()(x.double()); m.img_encoder.float()
print("torch-gpu fp32 vs golden", float(np.abs(ref.cpu().numpy() - g["img_out"]).max()))
print("golden vs fp64", float(np.abs(ref64.cpu().numpy() - g["img_out"]).max()))
print("torch-gpu fp32 vs fp64", float((ref.double() - ref64).abs().max()))
for v in (6, 0):
    plan = ImagePlan(m.img_encoder, m.attention_fusion.cross_attend_blocks[0], v)
    rows, packed = plan.run(x)
    got = rows.view(1, 15, 20, 128).permute(0, 3, 1, 2)
    print("variant", v, "vs golden", float(np.abs(got.cpu().numpy() - g["img_out"]).max()), "vs fp64", float((got.double() - ref64).abs().max()))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): plan.run(x)
    torch.cuda.synchronize(); print("  %.1f us per image branch (B=1)" % ((time.perf_counter() - t0) / 50 * 1e6))
    x2 = torch.cat([x, x], 0)
    plan.run(x2); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): plan.run(x2)
    torch.cuda.synchronize(); print("  %.1f us per image branch (B=2)" % ((time.perf_counter() - t0) / 50 * 1e6))
