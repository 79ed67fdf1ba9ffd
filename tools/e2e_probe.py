"""Where the host-array span of extract_features goes: every host-side step timed alone (MI355X box).
usage: python tools/e2e_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
dev = torch.device("cuda:0")
from bench import load_workload
xyz, img, voxel = load_workload(1.7, 0.025)
xyz = xyz.astype(np.float64)
n = len(xyz)
def t(fn, k=10, sync=True):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    if sync: torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
pin = torch.empty((n + 1024, 3), dtype=torch.float64).pin_memory()
src = torch.from_numpy(xyz)
print("torch copy_ pageable -> pinned 6.2 MB: %.3f ms" % t(lambda: pin[:n].copy_(src), sync=False))
pn = pin.numpy()
print("np.copyto pageable -> pinned:          %.3f ms" % t(lambda: np.copyto(pn[:n], xyz), sync=False))
pg = torch.empty((n + 1024, 3), dtype=torch.float64)
print("torch copy_ pageable -> pageable:      %.3f ms" % t(lambda: pg[:n].copy_(src), sync=False))
for nt in (1, 4, 16):
    torch.set_num_threads(nt)
    print("  threads=%d torch copy_ -> pinned:    %.3f ms" % (nt, t(lambda: pin[:n].copy_(src), sync=False)))
d = torch.empty((n + 1024, 3), dtype=torch.float64, device=dev)
print("H2D pinned 6.2 MB async + sync:        %.3f ms" % t(lambda: d[:n].copy_(pin[:n], non_blocking=True)))
print("H2D pageable 6.2 MB:                   %.3f ms" % t(lambda: d[:n].copy_(src)))
F = torch.randn(62000, 32, device=dev)
Fp = torch.empty((62000, 32), dtype=torch.float32).pin_memory()
print("D2H 7.9 MB -> pinned async + sync:     %.3f ms" % t(lambda: Fp.copy_(F, non_blocking=True)))
print("D2H 7.9 MB -> pageable (.cpu()):       %.3f ms" % t(lambda: F.cpu()))
print("pinned -> numpy .copy() 6.5 MB:        %.3f ms" % t(lambda: Fp[:51232].numpy().copy(), sync=False))
inds = torch.randint(0, n, (51232,), dtype=torch.int32).pin_memory()
print("host gather xyz[inds pinned int32]:    %.3f ms" % t(lambda: xyz[inds.numpy()], sync=False))
i64 = inds.numpy().astype(np.int64)
print("host gather xyz[inds int64 pageable]:  %.3f ms" % t(lambda: xyz[i64], sync=False))
ev = torch.cuda.Event()
print("event record + synchronize:            %.3f ms" % t(lambda: (ev.record(), ev.synchronize()), sync=False))

import imf_oracle as O
from imfnet_amd.extract import extract_features, extract_features_stream
from imfnet_amd.model import load_model
sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
model.load_state_dict(sd, strict=True); model = model.eval().to(dev)
torch.set_num_threads(8)
with torch.no_grad():
    for _ in range(4): extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): xd, Fd = extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    print("extract_features host arrays (F on the device + F.host): %.3f ms" % t(lambda: extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)))
    def stream(k):
        for _ in extract_features_stream(model, ((xyz, img) for _ in range(k)), voxel, dev, copy=COPY): pass
    COPY = False
    stream(4)
    pr = cProfile.Profile(); pr.enable()
    stream(10)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
    t0 = time.perf_counter(); stream(30); print("stream copy=False: %.3f ms / fragment" % ((time.perf_counter() - t0) / 30 * 1e3))
    COPY = True
    stream(4)
    t0 = time.perf_counter(); stream(30); print("stream copy=True: %.3f ms / fragment" % ((time.perf_counter() - t0) / 30 * 1e3))
