"""Isolated time of the image branch (imf_image_branch: trunk + LayerNorm + K/V projection) on the bench pair's images,
and its largest deviation from the fp32-MFMA arithmetic.  IMF_IMG_WAVES="<layer1>,<layer2>" picks the kernel of the 3x3
convolutions (0 = k_spconv_g with split-K + reduce, 4 / 8 = the wave-split kernel).  usage: python tools/image_branch_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
import bench
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
pts, imgs = bench.load_pair(1.7)
img = torch.as_tensor(np.asarray(imgs, dtype=np.float32)).to(dev).contiguous()
with torch.no_grad():
    model._ensure_image_plan() if hasattr(model, "_ensure_image_plan") else None
    plan = model._img_plan
    if plan is None:
        wl = bench.Workload(model, dev, pts, imgs, 0.025); wl.prepare_graph(); plan = model._img_plan
    print("image batch", tuple(img.shape))
    for _ in range(5): feat, packed = plan.run(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        e0.record()
        for _ in range(10): feat, packed = plan.run(img)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print("IMF_IMG_WAVES=%s: %.1f us per image branch (median of 15 x 10 back-to-back runs)" % (os.environ.get("IMF_IMG_WAVES", "0,0"), float(np.median(ts))))
    print("feat checksum %.9e  |feat|max %.4f" % (float(feat.double().sum()), float(feat.abs().max())))
