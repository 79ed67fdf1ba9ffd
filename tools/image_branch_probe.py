"""Is the image branch (captured hipGraph on a side stream) on the step's critical path?
Times (a) the graph replay alone, (b) the whole step, (c) the step with the replay skipped
(outputs of an earlier replay reused -- measurement only, not a valid product configuration)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import imf_oracle as O
from imfnet_amd.extract import sparse_tensor_from_points, start_geometry
from imfnet_amd.model import load_model
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
xyz_d, img_d = torch.as_tensor(xyz).to(dev), torch.as_tensor(img).to(dev)

def step():
    g = start_geometry(xyz_d, voxel, dev, inputs_ready=True)
    model.start_image_branch(img_d, inputs_ready=True)
    st, _ = sparse_tensor_from_points(None, voxel, dev, geometry=g)
    return model(st, img_d).F

def timed(n=40):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

with torch.no_grad():
    full = timed()
    key = (img_d.device, tuple(img_d.shape))
    graph, static_in, outs = model._img_graph[key]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): graph.replay()
    torch.cuda.synchronize(); alone = (time.perf_counter() - t) / 50 * 1e3
    orig = model._run_image_graph
    model._run_image_graph = lambda image, side: outs          # skip the replay
    skipped = timed()
    model._run_image_graph = orig
print(f"image graph alone {alone:.3f} ms/replay; step {full:.3f} ms; step without the replay {skipped:.3f} ms")
