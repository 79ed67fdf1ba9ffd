import sys, time, cProfile, pstats
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd.extract import extract_features
from imfnet_amd.model import load_model
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
xyz = xyz.astype(np.float64)
if len(sys.argv) > 1 and sys.argv[1] == "device":
    xyz = torch.as_tensor(xyz).to(dev); img = torch.as_tensor(img).to(dev)
with torch.no_grad():
    for _ in range(3): extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
    pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
