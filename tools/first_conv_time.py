import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from imfnet_amd import ops
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
w = (torch.randn(125, 1, 32) * 0.1).to(dev)
from imfnet_amd import _lib
out = torch.empty(levels[0].n, 32, device=dev)
for relu in (0, 2):
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.lib().imf_conv_first_fused(levels[0].table.data_ptr(), levels[0].capacity,
                                        levels[0].coords_buf.data_ptr(), levels[0].n, 1, 5, None, 1, w.data_ptr(), 32,
                                        None, None, relu, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("conv_first_fused k5 1->32 relu=%d (2 = phase 1 only):" % relu, np.median(ts[2:]), "us")

ts = []
for r in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o = ops.conv_first_bitgrid(levels[0], w, 5); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("conv_first_bitgrid (memset + fill + conv):", np.median(ts[2:]), "us", "bbox", levels[0].bbox)
