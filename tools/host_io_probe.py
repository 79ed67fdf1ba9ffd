import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops
from imfnet_amd.extract import extract_features, _as_device_points
from imfnet_amd.model import load_model
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
xyz = xyz.astype(np.float64)
with torch.no_grad():
    for _ in range(3): extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(10): xd, F = extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img); torch.cuda.synchronize()
    print("extract_features host arrays: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    t=time.perf_counter()
    for _ in range(10): p=_as_device_points(xyz, dev); torch.cuda.synchronize()
    print("upload pageable 6.2 MB: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    pin = torch.from_numpy(xyz).pin_memory()
    t=time.perf_counter()
    for _ in range(10): p=pin.to(dev, non_blocking=True); torch.cuda.synchronize()
    print("upload pinned: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    x32 = torch.from_numpy(xyz.astype(np.float32)).pin_memory()
    t=time.perf_counter()
    for _ in range(10): p=x32.to(dev, non_blocking=True); torch.cuda.synchronize()
    print("upload pinned f32: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    inds = torch.randint(0, len(xyz), (51232,), device=dev, dtype=torch.int32)
    t=time.perf_counter()
    for _ in range(10): ih = inds.cpu().numpy().astype(np.int64); rc = xyz[ih]
    print("inds D2H + host gather: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    t=time.perf_counter()
    for _ in range(10): im = torch.as_tensor(img, dtype=torch.float32).to(dev); torch.cuda.synchronize()
    print("image upload: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
    xd_dev = torch.as_tensor(xyz).to(dev); imd = torch.as_tensor(img).to(dev)
    t=time.perf_counter()
    for _ in range(10): xd, F = extract_features(model, xd_dev, voxel_size=voxel, device=dev, skip_check=True, image=imd); torch.cuda.synchronize()
    print("extract_features device tensors: %.2f ms" % ((time.perf_counter()-t)/10*1e3))
