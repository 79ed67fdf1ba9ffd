"""The host-array stream (extract_features_stream over the library pipeline) under its knobs, on an MI355X box.
usage: python tools/stream_probe.py [--sdma] [--blocks N] [--buckets N] [--depth N] [--batch N] [--n 60]
       GPU_MAX_HW_QUEUES=8 python tools/stream_probe.py ...   (read by the HIP runtime at start-up)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--sdma", type=int, default=-1, help="1 copy engines, 0 copy kernels, -1 the package default"); ap.add_argument("--blocks", type=int, default=0)
ap.add_argument("--buckets", type=int, default=3); ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--batch", default="2"); ap.add_argument("--n", type=int, default=60)
ap.add_argument("--f32", action="store_true", help="hand float32 point arrays in (no narrowing pass)")
ap.add_argument("--check", action="store_true")
ap.add_argument("--head", type=int, default=1, help="1: a job's level-0 head on the side stream (default), 0: on the main stream")
args = ap.parse_args()
args.batch = args.batch if args.batch == "auto" else int(args.batch)
dev = torch.device("cuda:0")
from bench import load_workload, build_model
import imf_oracle as O
from imfnet_amd.extract import extract_features, extract_features_stream
xyz, img, voxel = load_workload(1.7, 0.025)
xyz = xyz.astype(np.float32 if args.f32 else np.float64)
model, sd = build_model(dev)
with torch.no_grad():
    model.fragment_runner().streamer(dev, n_buckets=args.buckets, sdma_copies=None if args.sdma < 0 else bool(args.sdma), copy_blocks=args.blocks, head_on_side=bool(args.head))
    xd0, F0 = extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)   # exact path: teaches the runner
    F0 = F0.cpu().numpy()
    r = model.fragment_runner()
    def run(k, copy=False):
        out = None
        for xd, Fh in extract_features_stream(model, ((xyz, img) for _ in range(k)), voxel, dev, depth=args.depth, copy=copy,
                                              batch=args.batch):
            out = (xd, Fh)
        return out
    import imfnet_amd
    print("SDMA_ASYNC", imfnet_amd.SDMA_ASYNC, "ROC_CPU_WAIT_FOR_SIGNAL", os.environ.get("ROC_CPU_WAIT_FOR_SIGNAL"), "sdma copies:", model.fragment_runner().streamer(dev).sdma_copies)
    run(16)
    xd, Fh = run(4, copy=True)
    print("max |F_stream - F_exact| = %.3g, xyz_down equal: %s, M = %d" % (np.abs(Fh - F0).max(), bool((xd == xd0).all()), len(Fh)))
    for k in list(r.stats):
        if k.startswith("stream_"): r.stats.pop(k)
    r.stats["stream_trace"] = []
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(args.n); dt = (time.perf_counter() - t0) / args.n
    print("stream: %.3f ms / fragment wall (%.1f M desc/s); per-job event sum / fragment %.3f ms; stats %s" %
          (dt * 1e3, len(Fh) / dt / 1e6, r.stats.get("stream_gpu_ms", 0) / max(1, r.stats.get("stream_n", 1)), {k: v for k, v in r.stats.items() if k != "stream_trace"}))
    tr = r.stats.pop("stream_trace")
    nj = max(1, r.stats.get("stream_jobs", 1))
    print("  job: device[up0 up1 fwd0 fwd1 down1] host[submit popped up_issue0 up_issue1 fwd_issued dl_issue0 dl_issue1 returned]  (ms since pipeline creation, two clocks)")
    for row in tr[:12]:
        h = [row[5], row[8], row[9], row[10], row[6], row[11], row[12], row[7]]
        print("   ", " ".join("%9.3f" % v for v in row[:5]), " | ", " ".join("%9.3f" % v for v in h))
    print("  per job, host ms:", {k: round(v / nj, 3) for k, v in r.stats.items() if k.endswith("_ms") and k != "stream_gpu_ms"})
    # one job's three legs
    st = r.streamer(dev)
    from imfnet_amd.model.graph import HostSlot
    slots = [HostSlot() for _ in range(4)]
    jobs = [st.submit([(xyz, img)] * (4 if args.batch == 'auto' else args.batch), voxel, slots[i], more_follow=True) for i in range(3)]
    for j in jobs:
        j.wait()
        print("  job legs (upload, forward incl. queueing, download) ms:", ["%.3f" % v for v in j.ms], "host", ["%.3f" % v for v in j.host_ms])
    # synchronous extract_features (host arrays in, F.host out)
    ts = []
    for _ in range(25):
        t0 = time.perf_counter()
        xd, F = extract_features(model, xyz, voxel_size=voxel, device=dev, skip_check=True, image=img)
        Fh = F.host
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("extract_features sync: median %.3f ms (min %.3f)" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
