import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd.extract import sparse_tensor_from_points
from imfnet_amd.model import load_model
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
xyz_d, img_d = torch.as_tensor(xyz).to(dev), torch.as_tensor(img).to(dev)
outs = []
with torch.no_grad():
    for it in range(6):
        st, _ = sparse_tensor_from_points(xyz_d, voxel, dev)
        outs.append(model(st, img_d).F.clone())
        if os.environ.get("SYNC"): torch.cuda.synchronize()
torch.cuda.synchronize()
for i in range(1, 6):
    d = (outs[i] - outs[0]).abs()
    print(i, float(d.max()), int((d > 0).sum()), "nan", int(torch.isnan(outs[i]).sum()))
