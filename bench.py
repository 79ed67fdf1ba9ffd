#!/usr/bin/env python3
"""bench.py -- descriptors/sec of IMFNet's descriptor-generation hot path on MI355X.

A "step" = one full pass of the path over one fragment PAIR whose raw points and images are already
resident in HBM: voxelise (fp64 quantise + hash + first-occurrence unique) -> 4-level pyramid ->
8 rulebooks -> image encoder -> 23 sparse convolutions + fusion attention -> L2-normalised
[M,32] descriptors in HBM.  Nothing is cached between steps (every fragment is new geometry in
the real workload).  Workload at N=1: BASELINE.json configs[1] ("single 3DMatch fragment pair, voxel
2.5 cm") at the 3DMatch-shaped size its metric is quoted on (SURVEY §8d "S50k"): the in-tree pair
cloud_bin_0 / cloud_bin_1 x1.7 @ 2.5 cm = 51,232 + 52,164 voxels, run as ONE batched sparse tensor
(the model's batched call, model/resunet.py:241-250).  --batch 1 runs one fragment per step.
Weights are seeded random (no checkpoint is reachable), data says so.

Default execution (--mode capacity): one imf_fragment_forward call per step in capacity mode -- device-side row
counts, no host readback, ~150 launches issued natively on three streams; bit-identical to the exact path
(asserted below).  --mode graph replays the same call as ONE captured hipGraph: also bit-identical, but ROCm 7.2
executes a graph's independent branches one after the other (profiles/r02_graph_replay_kernel_stats.txt), which
puts the image branch, the coarse pyramid levels and the rulebook builds on the critical path -- measured slower,
reported in config.graph_replay.  --mode exact is round 1's path (count readback + native executor).  Every
--trace-every'th timed step records HIP events around each convolution for the live roofline block.

  python bench.py [--gpus N --steps K --warmup W]        one JSON line on rank 0
With --gpus N > 1 and no torchrun environment the script launches its own N ranks.
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
PMC_FILE = "r03_pmc_traffic.json"


def load_pair(scale):
    """Both in-tree fragments (7-scenes-redkitchen 0 and 1) scaled to the 3DMatch-shaped size, with their images."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
    im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
    pts = [z[f"cloud_bin_{k}"].astype(np.float64) * scale for k in (0, 1)]
    imgs = np.stack([np.transpose(im[f"image_{k}"], (2, 0, 1)) for k in (0, 1)]).copy()
    return pts, imgs


def load_workload(scale, voxel):
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
    xyz = z["cloud_bin_0"].astype(np.float64) * scale
    img = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))["image_0"]
    img = np.transpose(img, (2, 0, 1))[None].copy()
    return xyz, img, voxel


def algorithmic_bytes(rec):
    """pairs*(Cin+Cout)*4 + pairs*8 (rulebook index) + kvol*Cin*Cout*4 (weights)."""
    rb = rec["rb"]
    if "res" in rec:                                     # capacity mode: only the slots of the actual rows were written
        rows = rec["res"].counts[rec["level"]]
        pairs = rb.count_pairs(rec["arena"], min(rb.n_slots, (rows + 63) // 64 * 64 + rec["slots_extra"]), rows)
    elif "arena" in rec:
        pairs = rb.count_pairs(rec["arena"])
    else:
        pairs = int((rb.nbr >= 0).sum().item()) if rb.nbr is not None else rb.n_out
    return pairs * (rec["cin"] + rec["cout"]) * 4 + pairs * 8 + rec["kvol"] * rec["cin"] * rec["cout"] * 4, pairs


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r03_pmc_traffic.json, produced by tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in
    separate runs; read side doubled per the gfx950 FETCH_SIZE correction).  None if absent."""
    for name in (PMC_FILE, "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None, "no PMC profile committed"
    ks = json.load(open(path))["kernels"]
    # `kernel` names a template FAMILY (k_spconv_g<4, 0> = every (CAT, NB, RB) instance of the 64-column sparse
    # launches, which is what the live timing above groups too): launch-weighted average over its symbols
    key = "imf::" + kernel
    if kernel.startswith("k_spconv_w<"):              # k_spconv_w<W> = the symbols k_spconv_w<CAT, W>
        # the ResUNet's launches read operand images (third argument true); the two-argument symbols of the same kernel
        # are the image trunk's small dense convolutions, which the live timing above does not group either
        w = kernel[len("k_spconv_w<"):-1]
        fam = [v for k, v in ks.items() if re.match(r"imf::k_spconv_w<(true|false), %s, true>" % w, k)] or \
              [v for k, v in ks.items() if re.match(r"imf::k_spconv_w<(true|false), %s>" % w, k)]
    else:
        fam = [v for k, v in ks.items() if k == key or k.startswith(key[:-1] + ",") or k.startswith(key + "<")]
    if not fam:
        return None, f"{key} not in {os.path.basename(path)}"
    n = sum(v["launches"] for v in fam)
    avg = lambda f: sum(v[f] * v["launches"] for v in fam) / n
    hit = sum(v["l2_hit_rate"] * v["launches"] for v in fam) / n
    return int(avg("hbm_bytes_fetch_x2")), (f"bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (raw FETCH {int(avg('fetch_bytes_raw'))} B, "
                                            f"WRITE {int(avg('write_bytes'))} B, L2 hit {hit:.4f}; {n} launches of {len(fam)} "
                                            f"symbol(s) of the family) from profiles/{os.path.basename(path)}; below the algorithmic "
                                            f"bytes because feature rows are re-gathered from L2 / Infinity Cache, not HBM")


def rocprof_avg_us(kernel, name="r03_kernel_stats.txt"):
    """Launch-weighted average duration of `kernel`'s symbols in the committed `rocprofv3 --kernel-trace --stats` summary of
    this same command (profiles/r03_kernel_stats.txt, tools/profile_round.sh) -- printed beside the live HIP-event timing so
    that roofline.frac can be recomputed from profiles/ alone.  (n, avg_us) or None."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    rows = []
    for line in open(path):
        m = re.match(r"(?:void )?imf::(.*?)\(.*\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s*$", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
    if kernel.startswith("k_spconv_w<"):   # (operand-image symbols if the profile has them: see pmc_traffic)
        w = kernel[len("k_spconv_w<"):-1]
        fam = [r for r in rows if re.match(r"k_spconv_w<(true|false), %s, true>$" % w, r[0])] or \
              [r for r in rows if re.match(r"k_spconv_w<(true|false), %s>$" % w, r[0])]
    else:
        fam = [r for r in rows if r[0] == kernel or r[0].startswith(kernel[:-1] + ",") or r[0].startswith(kernel + "<")]
    n, tot = sum(r[1] for r in fam), sum(r[2] for r in fam)
    return (n, tot / n) if n else None


def cpu_baseline(xyz, img, voxel, sd, seconds_budget=12.0):
    """The oracle timed on this box's host cores over a bounded sample of the same workload: C hash-map geometry
    (MinkowskiEngine's CPU coordinate-map algorithm restated) + the convolutions either as the C / OpenMP twin
    `imf_cpu_spconv_fwd` (output-stationary gather-FMA) or as torch-CPU per-offset gather-GEMM-scatter
    (ME's CPU algorithm) -- whichever is faster here is `value`, the other is reported beside it; dense parts
    (image trunk, attention) are the torch-CPU ops the reference itself would run.  Thread counts are picked by a short
    probe (torch's intra-op pool does not scale to all cores of a large host: 256 threads are 6x SLOWER than 16 on the
    EPYC GPU box) and reported as `cores`; the single-thread rate of the faster implementation is reported too
    ("1 thread and all cores")."""
    import imf_oracle as O
    import imf_oracle_cbind as OC
    ncpu = os.cpu_count() or 1

    def once():
        t0 = time.perf_counter()
        coords, inds = OC.voxelize(xyz, voxel)
        geom = OC.Geometry(coords)
        F = O.resunet_forward(sd, coords, img, geometry=geom)
        return time.perf_counter() - t0, F.shape[0]

    def set_threads(impl, nt):
        O.SPCONV_IMPL = impl
        torch.set_num_threads(min(nt, 32) if impl == "c" else nt)     # dense parts: torch's pool stays small
        OC.set_threads(nt)

    best = {}
    for impl, cands in (("c", (16, 32, 64, 128, 256)), ("torch", (8, 16, 32))):
        for nt in sorted({min(ncpu, c) for c in cands}):
            set_threads(impl, nt)
            once()                                     # warm-up (page-in, thread pools)
            dt, _ = once()
            if impl not in best or dt < best[impl][1]:
                best[impl] = (nt, dt)
    impl = min(best, key=lambda k: best[k][1])
    other = "torch" if impl == "c" else "c"
    set_threads(impl, best[impl][0])
    times, m, spent = [], 0, 0.0
    while spent < seconds_budget and len(times) < 12:
        dt, m = once()
        times.append(dt)
        spent += dt
    med = statistics.median(times)
    set_threads(impl, 1)                               # one thread: one run
    one_t, _ = once()
    set_threads("torch", 16)
    O.SPCONV_IMPL = "torch"
    names = {"c": "C / OpenMP twin imf_cpu_spconv_fwd (output-stationary gather-FMA)",
             "torch": "torch-CPU per-offset gather-GEMM-scatter"}
    return {"value": round(m / med, 1), "unit": "descriptors/s", "cores": best[impl][0], "kind": "port",
            "value_1_thread": round(m / one_t, 1),
            "convolutions": names[impl],
            "other_implementation": {"convolutions": names[other], "value": round(m / best[other][1], 1), "cores": best[other][0]},
            "sample": f"the same fragment (M={m}) end to end on the host, median of {len(times)} runs "
                      f"({med * 1e3:.0f} ms each), {best[impl][0]} threads (best of a probe on a {ncpu}-cpu host; one run "
                      f"on 1 thread: {one_t * 1e3:.0f} ms): C hash-map voxelise / pyramid / rulebooks (OpenMP) + the "
                      f"convolutions as {names[impl]} + torch-CPU image encoder and attention"}


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def build_model(O, dev, variant=None):
    from imfnet_amd import ops
    from imfnet_amd.model import load_model
    prev = ops.CONV_VARIANT
    if variant is not None:
        ops.CONV_VARIANT = variant
    try:
        sd = O.seeded_state_dict(seed=0, with_unused_image_layers=True)
        model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True,
                                          conv1_kernel_size=5, D=3, config=None)
        model.load_state_dict(sd, strict=True)
        model = model.eval().to(dev)
        if variant is not None:                        # plans pack their weights lazily: force it under this variant
            from imfnet_amd.model.plan import FusedPlan
            model._plan = FusedPlan(model)
            model._native_image()
    finally:
        ops.CONV_VARIANT = prev
    return model, sd


class Workload:
    """Inputs resident in HBM + the two ways of running one step on them."""

    def __init__(self, model, dev, pts_list, imgs, voxel):
        from imfnet_amd.extract import sparse_tensor_from_points, start_geometry
        self.model, self.dev, self.voxel = model, dev, voxel
        self.xyz = torch.as_tensor(np.concatenate(pts_list, 0)).to(dev)
        self.img = torch.as_tensor(imgs).to(dev)
        self.starts, n = [], 0
        for p in pts_list:
            self.starts.append(n)
            n += len(p)
        self.item_starts = self.starts if len(pts_list) > 1 else None
        self._sp, self._geo = sparse_tensor_from_points, start_geometry
        self.runner = self.bucket = None
        self.stream = torch.cuda.Stream(device=dev)

    def exact_step(self):
        """Round 1's path: geometry stream + one count readback + native executor."""
        with torch.cuda.stream(self.stream):
            fut = self._geo(self.xyz, self.voxel, self.dev, inputs_ready=True, item_starts=self.item_starts)
            st, _ = self._sp(None, self.voxel, self.dev, geometry=fut)
            self.last_st = st
            return self.model(st, self.img).F

    def prepare_graph(self):
        """Capacity bucket from one exact step's counts; inputs staged in the bucket's static buffers."""
        F = self.exact_step()
        torch.cuda.synchronize()
        cm = self.last_st.coordinate_manager
        lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
        r = self.runner = self.model.fragment_runner()
        assert r is not None, "this model configuration is not covered by the fragment graph"
        r.observe(int(self.xyz.shape[0]), [l.n for l in lv], lv[0].bbox)
        key = r.caps_for(int(self.xyz.shape[0]), len(self.starts), int(self.img.shape[2]), int(self.img.shape[3]),
                         self.voxel, self.xyz.dtype == torch.float64)
        self.stream.synchronize()
        self.stream = r.main_stream(self.dev)            # capacity-mode forwards run on the runner's own main stream
        b = self.bucket = r.bucket(key, self.dev, self.stream)
        self.n_points = r.stage(b, self.xyz, self.starts, self.img, self.stream)
        return F

    def graph_step(self, trace_list=None):
        res = self.runner.launch(self.bucket, self.n_points, len(self.starts), self.stream, trace_list=trace_list)
        self.last_res = res
        return res


def timed(fn, steps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.7)
    ap.add_argument("--voxel", type=float, default=0.025)
    ap.add_argument("--batch", type=int, default=2, choices=(1, 2),
                    help="fragments per forward: 2 = the in-tree fragment PAIR (cloud_bin_0 + cloud_bin_1, one image "
                         "each) as ONE batched sparse tensor, the batched call of model/resunet.py:241-250")
    ap.add_argument("--mode", default="auto", choices=("auto", "capacity", "graph", "exact"),
                    help="capacity: imf_fragment_forward per step (device-side counts, no host readback); graph: the same "
                         "as one hipGraph replay; auto (default): both are timed after the clocks have settled and the faster "
                         "one runs the timed region (config.capacity_mode.picked says which); exact: count readback + native "
                         "executor")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is repeated this many times; "
                         "ms_per_step / value are the MEDIAN repeat, min and max are reported beside it")
    ap.add_argument("--settle-ms", type=float, default=600.0,
                    help="untimed steps run for at least this long before the first timed region (clock / power settle)")
    ap.add_argument("--trace-steps", type=int, default=3,
                    help="steps AFTER the timed regions that carry HIP events around every convolution (live roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-fragment / end-to-end / fp32-MFMA legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from imfnet_amd import dist as idist
    from imfnet_amd import ops
    import imf_oracle as O                              # seeded weights + cpu_baseline only

    # test hooks (single-GPU box): IMF_DIST_BACKEND=gloo IMF_FORCE_DEVICE=0 run N ranks on one device
    backend = os.environ.get("IMF_DIST_BACKEND", "nccl")
    if "IMF_FORCE_DEVICE" in os.environ:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        torch.cuda.set_device(int(os.environ["IMF_FORCE_DEVICE"]))
    rank, world, local = idist.init_from_env(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if "IMF_FORCE_DEVICE" in os.environ:
        local = int(os.environ["IMF_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    xyz1, img1, voxel = load_workload(args.scale, args.voxel)
    model, sd = build_model(O, dev)
    if args.batch == 2:
        pts, imgs = load_pair(args.scale)
    else:
        pts, imgs = [xyz1], img1
    wl = Workload(model, dev, pts, imgs, voxel)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()

    sync = torch.cuda.synchronize
    graph_info = None
    with torch.no_grad():
        dyn = args.mode != "exact"
        F_exact = wl.prepare_graph().clone() if dyn else None
        step = (lambda tl=None: wl.graph_step(tl)) if dyn else (lambda tl=None: wl.exact_step())
        if dyn:
            wl.runner.use_graph = args.mode == "graph"
        for _ in range(args.warmup):
            out = step()
        sync()

        # clock / power settle: untimed steps for >= --settle-ms (the first tens of ms after start-up run at other
        # clocks than the steady state: r02's driver line was a 28 ms sample taken 7 ms after the warm-up)
        settle_steps, t_s = 0, time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            for _ in range(10):
                out = step()
            sync()
            settle_steps += 10

        picked = {"mode": args.mode}
        if args.mode == "auto":                           # eager capacity-mode launches vs ONE hipGraph replay: measure, pick
            probe = {}
            for rnd in range(2):                          # interleaved rounds; the capture happens in round 0's warm-up
                for name, ug in (("capacity", False), ("graph", True)):
                    wl.runner.use_graph = ug
                    for _ in range(3):
                        step()
                    probe.setdefault(name, []).append(timed(step, max(10, args.steps), sync) * 1e3)
            best = min(probe, key=lambda k: min(probe[k]))
            wl.runner.use_graph = best == "graph"
            picked = {"mode": "auto", "picked": best,
                      "probe_ms_per_step": {k: [round(v, 4) for v in vs] for k, vs in probe.items()}}
            for _ in range(3):
                out = step()
            sync()
        run_mode = picked.get("picked", args.mode)

        if dyn:
            out = step()
            sync()
            assert out.flags == 0, f"capacity flags {out.flags}"
            M = out.counts[0]
            F_ref = out.F.clone()
            same = bool(torch.equal(F_ref, F_exact))
            assert float((F_ref - F_exact).abs().max()) < 1e-5, "graph path differs from the exact path"
            graph_info = {"hipgraph_replay": bool(wl.runner.use_graph), "graph_nodes": wl.bucket.n_nodes,
                          "equals_exact_path_bitwise": same, "host_readbacks_per_step": 0,
                          "capacities": {"points": wl.bucket.caps.n_points, "rows": list(wl.bucket.caps.rows)},
                          **picked}
        else:
            out = step()
            sync()
            M = out.shape[0]
            F_ref = out.clone()

        # ---- the timed regions: EXACTLY --steps steps each, barrier + synchronize on both sides, nothing else inside
        # (no event records, no readbacks).  Fragments are independent units: no data-path collective (each rank
        # would write its own <frag>.npz; the optional --gather of generate_desc is not the path).
        rep = []
        for _ in range(max(1, args.repeats)):
            barrier()
            sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = step()
            sync()
            barrier()
            rep.append(time.perf_counter() - t0)
        F_last = out.F if dyn else out

        # every step recomputes the same input: the last timed step must reproduce the warm-up step bit for bit
        drift = float((F_last - F_ref).abs().max())
        assert drift < 1e-5, f"descriptors of the last timed step differ from the warm-up step by {drift}"
        bit_reproducible = drift == 0.0

        # ---- live roofline: --trace-steps further steps, in situ (all three streams), HIP events around every
        # convolution on its launch stream -- AFTER the timed regions
        trace_all = []
        traced_steps = max(1, args.trace_steps)
        for i in range(traced_steps):
            if dyn:
                step(trace_all)
            else:
                ops.TRACE = trace_all
                step()
        ops.TRACE = None
        sync()
        trace = trace_all

    t = torch.tensor(rep, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per repeat: the slowest rank
        m = torch.tensor([M], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        total_m = int(m.item())
    else:
        total_m = M
    rep = sorted(float(v) for v in t.tolist())
    elapsed = rep[len(rep) // 2] if len(rep) % 2 else 0.5 * (rep[len(rep) // 2 - 1] + rep[len(rep) // 2])

    if rank == 0:
        # ---- live roofline of the dominant kernel (HIP events on the launch stream) -------------
        def group(records):
            groups, cache = {}, {}
            for rec in records:
                ms = rec["ev"].elapsed_ms()
                assert ms >= 0.0
                key = id(rec["rb"]) if "arena" not in rec else (rec["rb"].nbr, rec["rb"].n_slots), rec["cin"], rec["cout"]
                if key not in cache:
                    cache[key] = algorithmic_bytes(rec)
                g = groups.setdefault(rec["kernel"], {"ms": 0.0, "bytes": 0, "n": 0, "flops": 0})
                g["ms"] += ms
                g["bytes"] += cache[key][0]
                g["flops"] += 2 * cache[key][1] * rec["cin"] * rec["cout"]
                g["n"] += 1
            return groups

        groups = group(trace)
        dom = max(groups, key=lambda k: groups[k]["ms"])
        g = groups[dom]
        achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        conv_ms = sum(v["ms"] for v in groups.values()) / traced_steps
        traffic, traffic_note = pmc_traffic(dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_note": traffic_note,
                    "launches_per_step": g["n"] // traced_steps,
                    "avg_launch_us": round(g["ms"] * 1e3 / g["n"], 2),
                    "algorithmic_bytes_per_launch": g["bytes"] // g["n"],
                    "achieved_tflops_useful": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 2),
                    "all_sparse_conv_ms_per_step": round(conv_ms, 3),
                    "per_kernel": {k: {"launches_per_step": v["n"] // traced_steps, "avg_launch_us": round(v["ms"] * 1e3 / v["n"], 2),
                                       "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                   for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
                    "timing": "HIP events around each launch on its launch stream, in situ (other streams' kernels of the same "
                              "step overlap), %d steps after the timed regions" % traced_steps}
        rp = rocprof_avg_us(dom)
        if rp:
            roofline["rocprof_avg_launch_us"] = round(rp[1], 2)
            roofline["rocprof_frac"] = round(roofline["algorithmic_bytes_per_launch"] / (rp[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["rocprof_note"] = ("%d launches of the family in profiles/r03_kernel_stats.txt (rocprofv3 --kernel-trace --stats of "
                                        "this command; kernel tracing serialises the streams and carries no event records)" % rp[0])
        extras = {}
        with torch.no_grad():
            if dyn and world == 1:
                # the same launches with nothing else on the GPU: every stream collapsed onto one (serialised)
                iso = []
                for _ in range(3):
                    wl.bucket.io.serialize = 1
                    try:
                        wl.graph_step(iso)
                    finally:
                        wl.bucket.io.serialize = 0
                    sync()
                gi = group(iso).get(dom)
                if gi:
                    roofline["isolated_avg_launch_us"] = round(gi["ms"] * 1e3 / gi["n"], 2)
                    roofline["isolated_frac"] = round(gi["bytes"] / (gi["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if world == 1 and not args.no_extras:
                extras = extra_legs(O, model, dev, args, sync)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(xyz1, img1, voxel, sd)
        n_pts = int(wl.xyz.shape[0])
        out = {
            "metric": "descriptors/sec (32-D) on 3DMatch fragments",
            "value": round(total_m * args.steps / elapsed, 1),
            "unit": "descriptors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "timing": {"repeats": len(rep), "statistic": "median over the repeats of one timed region = exactly --steps steps "
                                                        "between barrier + synchronize (max over ranks per repeat)",
                       "ms_per_step_min": round(rep[0] / args.steps * 1e3, 4),
                       "ms_per_step_max": round(rep[-1] / args.steps * 1e3, 4),
                       "ms_per_step_all": [round(v / args.steps * 1e3, 4) for v in rep],
                       "settle_steps": settle_steps, "settle_ms": args.settle_ms,
                       "traced_steps_in_timed_region": 0},
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic (reference fixture fragment%s scaled x%.2f, seeded random weights)"
                     % (" PAIR cloud_bin_0 + cloud_bin_1" if args.batch == 2 else " cloud_bin_0", args.scale)),
            "config": {"workload": (f"3DMatch-shaped fragment pair: {n_pts} points -> {M} voxels @ "
                                    f"{voxel * 100:.1f} cm, one 120x160 image each, ResUNetBN2C 32-D, conv1 k5; the pair is "
                                    f"ONE batched forward per step per GPU, geometry rebuilt every step"
                                    if args.batch == 2 else
                                    f"3DMatch-shaped fragment: {n_pts} points -> {M} voxels @ "
                                    f"{voxel * 100:.1f} cm, image 120x160, ResUNetBN2C 32-D, conv1 k5; "
                                    f"one fragment per step per GPU, geometry rebuilt every step"),
                       "voxels_per_step_per_gpu": M, "points_per_step_per_gpu": n_pts,
                       "last_step_equals_warmup_bitwise": bit_reproducible,
                       "fragments_per_step": world * args.batch,
                       "execution": {"capacity": "one imf_fragment_forward call per step in capacity mode: device-side row counts, "
                                                 "no host readback, launches issued natively on three streams",
                                     "graph": "one hipGraph replay per step of imf_fragment_forward in capacity mode",
                                     "exact": "exact mode: row-count readback + native executor (imf_resunet_forward)"}[run_mode],
                       "capacity_mode": graph_info,
                       "image_branch": model.image_branch_mode if not dyn else wl.runner.image_branch_mode,
                       "conv_arithmetic": ("fp32 operands split into f16 hi+lo (weights pre-scaled by a power of two), "
                                           "3x v_mfma_f32_16x16x32_f16 with fp32 accumulation (fp32-class: max |dF| 3e-7 vs "
                                           "an fp64-accumulated network)" if ops.CONV_VARIANT == 6 else "fp32 MFMA"),
                       **extras},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def extra_legs(O, model, dev, args, sync):
    """Numbers the headline does not show (each a few hundred ms of GPU time):
      single_fragment        one S50k fragment per step (the reference harness is one fragment per call)
      e2e_extract_features   SURVEY §8(d)'s span: host arrays in -> voxelise + forward -> sync -> F copied back to the host
      fp32_mfma_variant0     the same pair with true-fp32 matrix instructions (v_mfma_f32_16x16x4_f32) everywhere"""
    from imfnet_amd.extract import extract_features
    xyz1, img1, voxel = load_workload(args.scale, args.voxel)
    out = {}
    wl1 = Workload(model, dev, [xyz1], img1, voxel)
    wl1.prepare_graph()
    wl1.runner.use_graph = False
    for _ in range(3):
        r = wl1.graph_step()
    dt = timed(wl1.graph_step, 20, sync)
    m1 = wl1.last_res.counts[0]
    out["single_fragment"] = {"descriptors_per_s": round(m1 / dt, 1), "ms_per_fragment": round(dt * 1e3, 4), "voxels": m1,
                              "execution": "capacity mode, one fragment per forward"}
    dt = timed(wl1.exact_step, 20, sync)
    out["single_fragment"]["exact_mode_ms_per_fragment"] = round(dt * 1e3, 4)

    xyz_host = xyz1.astype(np.float64)
    def e2e():
        xd, F = extract_features(model, xyz_host, voxel_size=voxel, device=dev, skip_check=True, image=img1)
        Fh = getattr(F, "host", None)                 # the descriptors as they came back with the counts (pinned block)
        return xd, (Fh if Fh is not None else F.cpu().numpy())
    for _ in range(3):
        xd, Fh = e2e()
    per_call = []
    for _ in range(25):                               # synchronous calls, each timed on its own: the MEDIAN is reported (a
        t0 = time.perf_counter()                      # single allocator / GC hiccup of tens of ms otherwise decides a 10-call mean)
        xd, Fh = e2e()
        per_call.append(time.perf_counter() - t0)
    per_call.sort()
    dt = per_call[len(per_call) // 2]
    out["e2e_extract_features"] = {"descriptors_per_s": round(Fh.shape[0] / dt, 1), "ms_per_fragment": round(dt * 1e3, 3),
                                   "ms_per_fragment_min_max": [round(per_call[0] * 1e3, 3), round(per_call[-1] * 1e3, 3)],
                                   "span": "extract_features(host float64 points [%d,3] + host image) -> xyz_down on the host, "
                                           "and F on the host (PCIe both ways: one pinned H2D block scalars | image | points, one "
                                           "pinned D2H block counts | xyz_down | F -- F.host of the returned tensor), one "
                                           "fragment at a time, synchronous" % len(xyz_host),
                                   "runner": dict(model.fragment_runner().stats) if model.fragment_runner() else None}

    from imfnet_amd.extract import extract_features_stream
    def frags(k):
        for _ in range(k):
            yield xyz_host, img1
    for _ in extract_features_stream(model, frags(6), voxel, dev):
        pass
    sync()
    n_it = 40
    m_tot = 0
    prof = None
    if os.environ.get("IMF_BENCH_PROFILE_STREAM"):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for xd, Fh in extract_features_stream(model, frags(n_it), voxel, dev, copy=False):
        m_tot += Fh.shape[0]
    dt = (time.perf_counter() - t0) / n_it
    st = model.fragment_runner().stats
    print("stream leg: %.3f ms wall per fragment, %.3f ms H2D + forward + D2H on the stream per fragment (%d)" %
          (dt * 1e3, st.get("stream_gpu_ms", 0.0) / max(1, st.get("stream_n", 0)), st.get("stream_n", 0)), file=sys.stderr)
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(12)
    out["e2e_extract_features_stream"] = {"descriptors_per_s": round(m_tot / n_it / dt, 1), "ms_per_fragment": round(dt * 1e3, 3),
                                          "span": "the same span over a stream of %d host fragments through extract_features_stream: "
                                                  "two fragments per forward (the model's batched call), one pinned H2D block and one pinned D2H block "
                                                  "per forward queued under the neighbouring forwards' kernels, xyz_down and F delivered "
                                                  "as host arrays (views of the pinned slots, copy=False), in order" % n_it}

    pts2, imgs2 = load_pair(args.scale)
    wl2 = Workload(model, dev, pts2, imgs2, voxel)
    wl2.prepare_graph()
    r = wl2.runner
    prev = r.use_graph
    r.use_graph = True
    for _ in range(3):
        res = wl2.graph_step()
    dt = timed(wl2.graph_step, 20, sync)
    out["graph_replay"] = {"descriptors_per_s": round(wl2.last_res.counts[0] / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                           "graph_nodes": wl2.bucket.n_nodes,
                           "note": "the pair as ONE hipGraph replay per step (same launches, bit-identical descriptors): ROCm 7.2 "
                                   "runs the graph's independent branches back to back, so the image branch, coarse pyramid "
                                   "levels and rulebook builds no longer overlap the convolutions"}
    r.use_graph = prev
    del wl2

    m0, _ = build_model(O, dev, variant=0)
    pts, imgs = load_pair(args.scale)
    wl0 = Workload(m0, dev, pts, imgs, voxel)
    for _ in range(3):
        F0 = wl0.exact_step()
    dt = timed(wl0.exact_step, 10, sync)
    out["fp32_mfma_variant0"] = {"descriptors_per_s": round(F0.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                                 "note": "the pair, exact mode, every convolution on v_mfma_f32_16x16x4_f32 (IMF_CONV_VARIANT=0)"}
    del m0, wl0
    return out


if __name__ == "__main__":
    main()
