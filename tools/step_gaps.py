"""GPU-side structure of consecutive steps without a profiler: spans and gaps from the HIP events the
library records around every sparse-conv kernel (ops.TRACE), all relative to one reference event."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd import ops, _lib
from imfnet_amd.extract import sparse_tensor_from_points, start_geometry
from imfnet_amd.model import load_model
from bench import load_workload
xyz, img, voxel = load_workload(1.7, 0.025)
dev = torch.device("cuda:0")
sd = O.seeded_state_dict(0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
model.load_state_dict(sd); model = model.eval().to(dev)
xyz_d = torch.as_tensor(xyz).to(dev); img_d = torch.as_tensor(img).to(dev)
prefetch = len(sys.argv) > 1 and sys.argv[1] == "prefetch"
queued = []
def pre():
    queued.append(start_geometry(xyz_d, voxel, dev, inputs_ready=True)); model.start_image_branch(img_d, inputs_ready=True)
def step():
    if not queued: pre()
    st, _ = sparse_tensor_from_points(None, voxel, dev, geometry=queued.pop(0))
    if prefetch: model.after_fusion_hook = pre
    return model(st, img_d).F
L = _lib.lib()
with torch.no_grad():
    for _ in range(5): step()
    torch.cuda.synchronize()
    ref = ops._Ev(); 
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventRecord(ctypes.c_void_p(ref.begin), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    ops.TRACE = []
    marks = []
    t0 = time.perf_counter()
    for i in range(12):
        marks.append(len(ops.TRACE)); step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 12 * 1e3
    tr, ops.TRACE = ops.TRACE, None
marks.append(len(tr))
print("wall ms/step", round(wall, 3), "prefetch", prefetch)
for i in range(3, 9):
    recs = tr[marks[i]:marks[i + 1]]
    s = [L.imf_event_elapsed_ms(ref.begin, r["ev"].begin) * 1e3 for r in recs]
    e = [L.imf_event_elapsed_ms(ref.begin, r["ev"].end) * 1e3 for r in recs]
    busy = sum(b - a for a, b in zip(s, e))
    gaps = [(recs[j + 1]["name"], s[j + 1] - e[j]) for j in range(len(recs) - 1)]
    big = sorted(gaps, key=lambda g: -g[1])[:4]
    prev_end = L.imf_event_elapsed_ms(ref.begin, tr[marks[i] - 1]["ev"].end) * 1e3
    print(f"step {i}: first conv start {s[0]:9.1f}  span {e[-1] - s[0]:7.1f}  conv busy {busy:7.1f}  "
          f"gap since prev step's last conv {s[0] - prev_end:6.1f}  biggest inner gaps:", [(n, round(g)) for n, g in big])
