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
reported in config.graph_replay.  --mode exact is round 1's path (count readback + native executor).

`value` / `ms_per_step` follow the bench contract (inputs resident in HBM when the timed region starts; the points of
every step come from a different one of 26 replicas, 330 MB in all, so that no step finds its input in the 256 MiB
Infinity Cache).  The SAME JSON line carries `host_span`: SURVEY 8(d)'s span -- host float64 arrays in, xyz_down and the
descriptors back on the host, PCIe both ways -- timed the same way (exactly --steps pair-steps between barrier +
synchronize, median of the repeats) through the library's streaming pipeline (imf_pipeline_*: csrc/pipeline.hip).

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
import imfnet_amd  # noqa: E402,F401  (before torch touches the HIP runtime: sets ROC_CPU_WAIT_FOR_SIGNAL=0, see its __init__)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md: aggregate L2 bandwidth
# The three arithmetics the convolutions can run on (imfnet_amd/ops.py CONV_VARIANT; imf_conv_args.variant).  `peak_tf` = the
# dense matrix peak of MI355X_MICROARCH.md for the instruction used, divided by the matrix instructions one fp32
# multiply-add block costs: the fp32-EQUIVALENT matrix roofline of the arithmetic.
ARITH = {
    "bf16x3": {"variant": 3, "suffix": "/b3", "ar": "3", "peak_tf": 2500.0 / 6, "mfma_flops": 16384,
               "dtype": "f32 (each fp32 operand split EXACTLY into 3 bf16 parts = 24 significant bits, fp32 exponent range; "
                        "6 x v_mfma_f32_16x16x32_bf16 per 32 channels, fp32 accumulate)",
               "arithmetic": "a = a0 + a1 + a2, w = w0 + w1 + w2 exactly (bf16 parts, round-to-nearest residuals); products a0w2 + "
                             "a1w1 + a2w0 + a0w1 + a1w0 + a0w0 with fp32 accumulation in the MFMA; the three dropped terms are "
                             "together <= 2^-26 |a||w| (a quarter of an fp32 ulp of the product); fp32 buffers between layers, no "
                             "range restriction, no recompute path"},
    "f16x2": {"variant": 6, "suffix": "", "ar": "(0|1)", "peak_tf": 2500.0 / 3, "mfma_flops": 16384,
              "dtype": "f32 (2xf16-split operands, 22-bit, 3 x v_mfma_f32_16x16x32_f16; fp32 accumulate) -- FAST mode, narrower "
                       "operands than the reference's fp32",
              "arithmetic": "fp32 operands split into f16 hi + lo (weights pre-scaled by a power of two), lo*hi + hi*lo + hi*hi "
                            "with fp32 accumulation; activations outside the f16 range raise IMF_FLAG_RANGE and the fragment "
                            "is redone on fp32 MFMA"},
    "f32": {"variant": 0, "suffix": "/f32", "ar": "2", "peak_tf": 157.3, "mfma_flops": 2048,
            "dtype": "f32 (v_mfma_f32_16x16x4_f32: fp32 operands and accumulation, the reference's arithmetic)",
            "arithmetic": "fp32 rows and fp32 weights through the same LDS-DMA kernels into v_mfma_f32_16x16x4_f32"},
}
VARIANT_NAME = {v["variant"]: k for k, v in ARITH.items()}
PROFILE_TAGS = ("r06", "r05")  # profiles/<tag>_kernel_stats_<arith>.txt, <tag>_pmc_traffic_<arith>.json, <tag>_pmc_counters_<arith>.txt: newest first


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


def delivered_bytes(rec, arith, pairs):
    """Bytes that enter the CUs through their vector-memory path (L2 -> L1 / LDS / registers: 64 B per clock and CU, 34.5 TB/s
    chip-wide, MI355X_MICROARCH.md) for one launch of the LDS-DMA convolution kernels -- the resource that binds them:
    every unit of rows (64-row tile; half tile: kernel_tag 64; 48-row unit: 128) fetches the weight block of each active (offset,
    32-channel) sub-stage of its tile once per 64-column slab (12 KiB bf16x3, 8 KiB split-f16 / fp32), plus the gathered input
    rows of the occupied pairs, once per slab (a missing neighbour is a zero fill without traffic).  None for 1x1x1 layers."""
    rb = rec["rb"]
    if rec["kvol"] <= 1 or "arena" not in rec or not rb.nbr:
        return None
    arena = rec["arena"]
    start = (rb.nbr - arena.data_ptr()) // 4 + rb.kvol * rb.n_slots      # the tile masks sit behind the neighbour table
    n_tiles = rb.n_slots // 64
    if "res" in rec:
        rows = rec["res"].counts[rec["level"]]
        n_tiles = min(n_tiles, ((rows + 63) // 64 * 64 + rec["slots_extra"]) // 64)
    words = arena[start:start + 4 * n_tiles].view(n_tiles, 4)[:, 0]
    m = words.to(torch.int64) & 0x7FFFFFF
    active = 0
    for k in range(27):
        active += int(((m >> k) & 1).sum().item())
    tag = rec.get("kernel_tag", 0)
    units = 2.0 if tag & 64 else (4.0 / 3.0 if tag & 128 else 1.0)
    cin, cout = rec["cin"], rec["cout"]
    wide = cout % 64 == 0
    slabs = cout // 64 if wide else cout // 32
    sub = (12288 if arith == "bf16x3" else 8192) // (1 if wide else 2)
    return int(active * units * (cin // 32) * slabs * sub + pairs * cin * 4 * slabs)


def _family_regex(label, arith):
    """Regex over demangled kernel symbols (without the imf:: prefix and the argument list) for a bench label.  Labels come
    from ops.conv_kernel_name: `k_spconv_w<W>` / `k_spconv_g<CB, 0>` (+ `/b3`, `/f32`) name a template FAMILY -- every
    (CAT, NB, RB, WS1) instance of the arithmetic AR (template argument before the last of k_spconv_g, third of k_spconv_w: 0 / 1 split-f16 without / with operand images,
    2 fp32, 3 bf16x3); anything else is a plain symbol."""
    ar = ARITH[arith]["ar"]
    base = label.split("/")[0]
    if arith == "f16x2":
        ar = "1"                                          # the ResUNet's launches read operand images; 0 = the image trunk's
    m = re.match(r"k_spconv_w<(\d)>$", base)
    if m:
        return r"k_spconv_w<(true|false), %s, %s, 0(, \d)*>$" % (m.group(1), ar)   # (label 0; 1 = the image trunk's launches; RB, OCC)
    m = re.match(r"k_spconv_g<(\d), (\d)>$", base)
    if m:
        return r"k_spconv_g<%s, %s, (true|false), \d, \d, %s(, (true|false))?>$" % (m.group(1), m.group(2), ar)
    return re.escape(base) + r"(<.*>)?$"


def _profile(kind, arith):
    """profiles/<PROFILE_TAG>_<kind>_<arith>.<ext> (this round's passes of this same command per arithmetic)."""
    ext = "json" if kind == "pmc_traffic" else "txt"
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", "%s_%s_%s.%s" % (tag, kind, arith, ext))
        if os.path.exists(path):
            return path
    return None


def pmc_traffic(kernel, arith):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r05_pmc_traffic_<arith>.json, produced by tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in
    separate runs; read side doubled per the gfx950 FETCH_SIZE correction).  None if absent."""
    path = _profile("pmc_traffic", arith)
    if not path:
        return None, "no PMC profile committed for %s" % arith
    ks = json.load(open(path))["kernels"]
    rx = re.compile(_family_regex(kernel, arith))
    fam = [v for k, v in ks.items() if rx.match(k[len("imf::"):] if k.startswith("imf::") else k)]
    if not fam:
        return None, f"{kernel} not in {os.path.basename(path)}"
    n = sum(v["launches"] for v in fam)
    avg = lambda f: sum(v[f] * v["launches"] for v in fam) / n
    hit = sum(v["l2_hit_rate"] * v["launches"] for v in fam) / n
    return int(avg("hbm_bytes_fetch_x2")), (f"bytes/launch = 2*FETCH_SIZE + WRITE_SIZE (raw FETCH {int(avg('fetch_bytes_raw'))} B, "
                                            f"WRITE {int(avg('write_bytes'))} B, L2 hit {hit:.4f}; {n} launches of {len(fam)} "
                                            f"symbol(s) of the family) from profiles/{os.path.basename(path)}; below the algorithmic "
                                            f"bytes because feature rows are re-gathered from L2 / Infinity Cache, not HBM")


def rocprof_avg_us(kernel, arith):
    """Launch-weighted average duration of `kernel`'s symbols in the committed `rocprofv3 --kernel-trace --stats` summary of
    this same command (profiles/r05_kernel_stats_<arith>.txt, tools/profile_round.sh) -- printed beside the live HIP-event
    timing so that roofline.frac can be recomputed from profiles/ alone.  (n, avg_us, file) or None."""
    path = _profile("kernel_stats", arith)
    if not path:
        return None
    rows = []
    for line in open(path):
        m = re.match(r"(?:void )?imf::(.*?)\(.*\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s*$", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
    rx = re.compile(_family_regex(kernel, arith))
    fam = [r for r in rows if rx.match(r[0])]
    n, tot = sum(r[1] for r in fam), sum(r[2] for r in fam)
    return (n, tot / n, os.path.basename(path)) if n else None


def pmc_counters(kernel, arith):
    """(matrix-pipe busy %, MFMA instructions per launch, file) of `kernel`'s family from the committed counter table
    (profiles/r05_pmc_counters_<arith>.txt, tools/pmc_table.py), launch-weighted; None if absent."""
    path = _profile("pmc_counters", arith)
    if not path:
        return None
    rows = []
    for line in open(path):
        m = re.match(r"(k_\S.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+\d+\s+(\d+) /", line)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(5)), int(m.group(7))))
    rx = re.compile(_family_regex(kernel, arith))
    fam = [r for r in rows if rx.match(r[0])]
    n = sum(r[1] for r in fam)
    if not n:
        return None
    return (sum(r[2] * r[1] for r in fam) / n, sum(r[3] * r[1] for r in fam) / n, os.path.basename(path))


def cpu_baseline(xyz, img, voxel, sd, seconds_budget=12.0):
    """The oracle timed on this box's host cores over a bounded sample of the same workload: C hash-map geometry
    (MinkowskiEngine's CPU coordinate-map algorithm restated) + the convolutions either as the C / OpenMP twin
    `imf_cpu_spconv_fwd` (output-stationary gather-FMA) or as torch-CPU per-offset gather-GEMM-scatter
    (ME's CPU algorithm) -- whichever is faster here is `value`, the other is reported beside it; dense parts
    (image trunk, attention) are the torch-CPU ops the reference itself would run.  Thread counts are picked by a short
    probe (torch's intra-op pool does not scale to all cores of a large host: 256 threads are 6x SLOWER than 16 on the
    EPYC GPU box) and reported as `cores`; the single-thread rate of the faster implementation is reported too
    ("1 thread and all cores")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))       # the checker: only this leg ever imports it
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


def group_trace(records, arith=None):
    """Traced launches (HIP events recorded by the library right around each convolution kernel on its launch stream) grouped
    by kernel family: total ms, algorithmic bytes and useful flops."""
    groups, cache, dcache = {}, {}, {}
    for rec in records:
        ms = rec["ev"].elapsed_ms()
        assert ms >= 0.0
        key = id(rec["rb"]) if "arena" not in rec else (rec["rb"].nbr, rec["rb"].n_slots), rec["cin"], rec["cout"]
        if key not in cache:
            cache[key] = algorithmic_bytes(rec)
        g = groups.setdefault(rec["kernel"], {"ms": 0.0, "bytes": 0, "n": 0, "flops": 0, "delivered": 0, "delivered_ms": 0.0})
        g["ms"] += ms
        g["bytes"] += cache[key][0]
        g["flops"] += 2 * cache[key][1] * rec["cin"] * rec["cout"]
        g["n"] += 1
        if arith is not None:
            dkey = key + (rec.get("kernel_tag", 0),)
            if dkey not in dcache:
                dcache[dkey] = delivered_bytes(rec, arith, cache[key][1])
            if dcache[dkey] is not None:
                g["delivered"] += dcache[dkey]
                g["delivered_ms"] += ms
    return groups


def build_roofline(groups, arith, traced_steps, ms_per_step, iso_groups=None):
    """The `roofline` object of one arithmetic.  The dominant kernel family (by time) against BOTH rooflines:
      * the matrix pipe -- `bound: "mfma"` -- useful fp32 flops (2 * pairs * Cin * Cout, no padding) per launch / launch time
        against the fp32-EQUIVALENT matrix peak of the arithmetic (dense 16-bit peak / matrix instructions per fp32
        multiply-add block; 157.3 TF for the fp32 MFMA), with the counter-measured matrix-pipe occupancy beside it;
      * HBM, notionally: SURVEY 8(d)'s algorithmic bytes per launch / launch time / 8 TB/s (`notional_hbm_frac`; measured HBM
        traffic is a third of those bytes -- rows are re-gathered from L2 -- so HBM is not what binds)."""
    A = ARITH[arith]
    dom = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[dom]
    sec = g["ms"] * 1e-3
    tf = g["flops"] / sec / 1e12
    gbs = g["bytes"] / sec / 1e9
    conv_ms = sum(v["ms"] for v in groups.values()) / traced_steps
    conv_bytes = sum(v["bytes"] for v in groups.values()) / traced_steps
    conv_flops = sum(v["flops"] for v in groups.values()) / traced_steps
    traffic, traffic_note = pmc_traffic(dom, arith)

    def counters(kernel, avg_us):
        pc = pmc_counters(kernel, arith)
        if not pc:
            return {}
        busy, mfma, src = pc
        return {"mfma_busy": round(busy / 100.0, 4), "mfma_per_launch": int(mfma),
                "issued_tflops": round(mfma * A["mfma_flops"] / (avg_us * 1e-6) / 1e12, 1),
                "counters": "profiles/" + src}

    def per_kernel_entry(k, v):
        avg_us = v["ms"] * 1e3 / v["n"]
        e = {"launches_per_step": v["n"] // traced_steps, "avg_launch_us": round(avg_us, 2),
             "useful_tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
             "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / A["peak_tf"], 4),
             "notional_hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        cp = cu_path(v)
        if cp:
            e["cu_vmem_frac"] = cp["frac"]
        e.update(counters(k, avg_us))
        rp = rocprof_avg_us(k, arith)
        if rp:
            e["rocprof_avg_launch_us"] = round(rp[1], 2)
        return e

    avg_us = g["ms"] * 1e3 / g["n"]

    def cu_path(v):
        """The CU vector-memory path: bytes delivered into the CUs (delivered_bytes) / launch time against 34.5 TB/s."""
        if not v.get("delivered_ms"):
            return None
        gbs = v["delivered"] / (v["delivered_ms"] * 1e-3) / 1e9
        return {"achieved": round(gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / L2_PEAK_GBS, 4)}

    conv_delivered = sum(v["delivered"] for v in groups.values()) / traced_steps
    conv_delivered_ms = sum(v["delivered_ms"] for v in groups.values()) / traced_steps
    r = {"bound": "mfma", "kernel": dom, "arithmetic": arith,
         "achieved": round(tf, 2), "peak": round(A["peak_tf"], 1), "unit": "TFLOP/s", "frac": round(tf / A["peak_tf"], 4),
         "peak_def": {"bf16x3": "2500 TFLOP/s dense 16-bit / 6 bf16 MFMAs per fp32 product block = 416.7",
                      "f16x2": "2500 TFLOP/s dense 16-bit / 3 f16 MFMAs per product block = 833.3",
                      "f32": "157.3 TFLOP/s fp32 MFMA"}[arith],
         "peak_note": ("fp32-equivalent matrix peak of this arithmetic: MI355X dense 16-bit matrix peak 2500 TFLOP/s / %d matrix "
                       "instructions per fp32 multiply-add block" % (6 if arith == "bf16x3" else 3)) if arith != "f32" else
                      "fp32 matrix peak (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md",
         "achieved_note": "USEFUL flops (2 * pairs * Cin * Cout of the rulebook's occupied pairs; the 64-row tiles also multiply "
                          "their empty (row, offset) slots, ~48 % of the issued matrix work: `issued_tflops`) / launch time",
         "traffic": traffic, "traffic_note": traffic_note,
         "notional_hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                          "note": "SURVEY 8(d)'s algorithmic bytes per launch / launch time / 8 TB/s: the contract's HBM roofline. "
                                  "Notional -- the counter traffic above is a fraction of these bytes (rows are re-gathered "
                                  "from L2 / Infinity Cache), so HBM is not the binding resource"},
         "cu_vmem_path": dict(cu_path(g) or {}, note="what BINDS these kernels (DESIGN 3): bytes that enter the CUs through their "
                              "vector-memory path (L2 -> L1 / LDS / registers, 64 B per clock and CU = 34.5 TB/s): every unit of rows "
                              "re-fetches the weight block of each of its active sub-stages (12 KiB bf16x3) + the gathered rows; "
                              "the L2s serve them (hit 0.91-0.93), HBM sees a sixth"),
         "step_cu_vmem_frac": round(conv_delivered / (conv_delivered_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if conv_delivered_ms else None,
         "launches_per_step": g["n"] // traced_steps,
         "avg_launch_us": round(avg_us, 2),
         "algorithmic_bytes_per_launch": g["bytes"] // g["n"],
         "algorithmic_flops_per_launch": g["flops"] // g["n"],
         "all_sparse_conv_ms_per_step": round(conv_ms, 3),
         "step_frac": round(conv_flops / (ms_per_step * 1e-3) / 1e12 / A["peak_tf"], 4),
         "step_notional_hbm_frac": round(conv_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "step_frac_note": "sum of the convolutions' useful flops (algorithmic bytes) per step / ms_per_step / peak: the whole "
                           "step against the roofline -- geometry, image branch and fusion count as time only",
         "per_kernel": {k: per_kernel_entry(k, v) for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
         "timing": "HIP events around each launch on its launch stream, in situ (other streams' kernels of the same "
                   "step overlap), %d steps after the timed regions" % traced_steps}
    r.update(counters(dom, avg_us))
    rp = rocprof_avg_us(dom, arith)
    if rp:
        r["rocprof_avg_launch_us"] = round(rp[1], 2)
        r["rocprof_frac"] = round(g["flops"] / g["n"] / (rp[1] * 1e-6) / 1e12 / A["peak_tf"], 4)
        r["rocprof_notional_hbm_frac"] = round(g["bytes"] / g["n"] / (rp[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        r["rocprof_note"] = ("%d launches of the family in profiles/%s (rocprofv3 --kernel-trace --stats of this command with "
                             "IMF_CONV_VARIANT=%d; kernel tracing serialises the streams and carries no event records)"
                             % (rp[0], rp[2], A["variant"]))
    if iso_groups and dom in iso_groups:
        gi = iso_groups[dom]
        r["isolated_avg_launch_us"] = round(gi["ms"] * 1e3 / gi["n"], 2)
        r["isolated_frac"] = round(gi["flops"] / (gi["ms"] * 1e-3) / 1e12 / A["peak_tf"], 4)
    return r


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


def build_model(dev, variant=None):
    """ResUNetBN2C with the seeded weights of imfnet_amd/seeded.py (no checkpoint is reachable)."""
    from imfnet_amd import ops
    from imfnet_amd.model import load_model
    from imfnet_amd.seeded import seeded_state_dict
    prev = ops.CONV_VARIANT
    if variant is not None:
        ops.CONV_VARIANT = variant
    try:
        sd = seeded_state_dict(seed=0, with_unused_image_layers=True)
        model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True,
                                          conv1_kernel_size=5, D=3, config=None)
        model.load_state_dict(sd, strict=True)
        model = model.eval().to(dev)
        if variant is not None:                        # plans pack their weights lazily: force it under this variant
            model.pinned_variant = variant             # (and keep them when the process default is another one)
            from imfnet_amd.model.plan import FusedPlan
            model._plan = FusedPlan(model)
            model._native_image()
    finally:
        ops.CONV_VARIANT = prev
    return model, sd


class Workload:
    """Inputs resident in HBM + the two ways of running one step on them."""

    MALL_BYTES = 320 << 20        # replicas of the points beyond the 256 MiB Infinity Cache

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
        self.replicas, self._turn = [], 0
        self.buckets, self._done, self.pipelined = [], [], False

    def exact_step(self):
        """Round 1's path: geometry stream + one count readback + native executor."""
        with torch.cuda.stream(self.stream):
            fut = self._geo(self.xyz, self.voxel, self.dev, inputs_ready=True, item_starts=self.item_starts)
            st, _ = self._sp(None, self.voxel, self.dev, geometry=fut)
            self.last_st = st
            return self.model(st, self.img).F

    def prepare_graph(self, replicate=False, pipelined=False):
        """Capacity bucket from one exact step's counts; inputs staged in the bucket's static buffers.  replicate: the
        points additionally as enough device copies to exceed the Infinity Cache; every step reads the next one.
        pipelined: TWO buckets of the same capacities, steps alternate between them and step k + 1's head (table reset,
        level-0 pyramid, image fork: ~60 us) is issued on the side stream, under step k's decoder -- what the streaming
        pipeline does with consecutive forwards (imf_fragment_io.head_on_side); same kernels, same results."""
        F = self.exact_step()
        torch.cuda.synchronize()
        cm = self.last_st.coordinate_manager
        lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
        r = self.runner = self.model.fragment_runner()
        assert r is not None, "this model configuration is not covered by the fragment graph"
        r.observe(int(self.xyz.shape[0]), [l.n for l in lv], lv[0].bbox)
        if len(self.starts) > 1:
            r.observe_batch(len(self.starts), lv[0].bbox)
        key = r.caps_for(int(self.xyz.shape[0]), len(self.starts), int(self.img.shape[2]), int(self.img.shape[3]),
                         self.voxel, self.xyz.dtype == torch.float64)
        self.stream.synchronize()
        self.stream = r.main_stream(self.dev)            # capacity-mode forwards run on the runner's own main stream
        b = self.bucket = r.bucket(key, self.dev, self.stream)
        self.n_points = r.stage(b, self.xyz, self.starts, self.img, self.stream)
        self.buckets, self._done, self.pipelined = [b], [None], bool(pipelined)
        if pipelined:
            # (a lane of its own: lanes 1..n of a key belong to the runner's FragmentStreamer, which the host-span leg uses)
            b1 = r.bucket(key, self.dev, self.stream, lane="bench")
            assert r.stage(b1, self.xyz, self.starts, self.img, self.stream) == self.n_points
            self.buckets.append(b1)
            self._done.append(None)
        if replicate:
            nbytes = self.xyz.numel() * self.xyz.element_size()
            self.replicas = [self.xyz.clone() for _ in range(max(2, -(-self.MALL_BYTES // nbytes)))]
            self._home = b.io.xyz
            self._homes = [bb.io.xyz for bb in self.buckets]
        return F

    def graph_step(self, trace_list=None):
        pipe = self.pipelined and not self.runner.use_graph and trace_list is None
        k = self._turn % len(self.buckets) if pipe else 0
        b = self.buckets[k] if self.buckets else self.bucket
        if self.replicas and not self.runner.use_graph:  # (a captured graph has the bucket's own buffer baked in)
            b.io.xyz = self.replicas[self._turn % len(self.replicas)].data_ptr()
        elif self.replicas:
            b.io.xyz = self._homes[k]
        self._turn += 1
        reuse = None
        if pipe:
            # the end of this bucket's previous forward (two steps ago): nothing else reads its buffers in the bench
            reuse = self._done[k]
            if reuse is None:
                reuse = torch.cuda.Event()
                reuse.record(self.stream)
        res = self.runner.launch(b, self.n_points, len(self.starts), self.stream, trace_list=trace_list, reuse_event=reuse)
        if pipe:
            e = torch.cuda.Event()
            e.record(self.stream)
            self._done[k] = e
        self.last_res = res
        return res


def timed(fn, steps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def median(vals):
    v = sorted(vals)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def host_span_leg(model, dev, args, barrier, sync, pts, imgs, voxel, f32_valued=False, batch=None):
    """SURVEY 8(d)'s span as a stream: host arrays in -> xyz_down + descriptors on the host, two fragments per forward
    (one `step` = the pair), through imf_pipeline_* (worker thread, transfers on the copy engines under the neighbouring
    forwards).  Timed like the headline: exactly --steps steps between barrier + synchronize, median of the repeats."""
    from imfnet_amd.extract import extract_features_stream
    if f32_valued:                                       # what a PLY holds (float32 values widened by the reader)
        pts = [p.astype(np.float32).astype(np.float64) for p in pts]
    R = 6                                                # host replicas: a step's source pages are not the previous step's
    frags = [[(p.copy(), np.ascontiguousarray(imgs[k:k + 1])) for k, p in enumerate(pts)] for _ in range(R)]

    def run(n_steps):
        gen = (frags[s % R][k] for s in range(n_steps) for k in range(len(pts)))
        m = 0
        for xd, Fh in extract_features_stream(model, gen, voxel, dev, depth=3, copy=False, batch=batch or len(pts)):
            m += Fh.shape[0]
        return m

    run(max(args.warmup, 8))                             # every pinned slot and bucket lane has been through the pipeline
    sync()
    st = model.fragment_runner().stats
    for k in [k for k in st if k.startswith("stream_")]:
        st.pop(k)
    st["stream_trace"] = []
    # A timed region of a STREAM pays the pipeline's fill and drain once (the first job's upload, the last job's download:
    # ~1.5 ms together): regions of >= 100 pair-steps, so that the edges weigh 1 % instead of the 4-6 % they were in
    # --steps-long regions (rounds 4-5 reported those; `region_steps` says what was timed).
    region = max(args.steps, 100)
    rep, m = [], 0
    for _ in range(max(3, min(args.repeats, 5))):
        barrier()
        sync()
        t0 = time.perf_counter()
        m = run(region)
        sync()
        barrier()
        rep.append(time.perf_counter() - t0)
    trace = st.pop("stream_trace")
    jobs = max(1, st.get("stream_jobs", 1))
    t = median(rep)
    streamer = model.fragment_runner().streamer(dev)
    out = {"value": round(m / t, 1), "unit": "descriptors/s", "ms_per_step": round(t / region * 1e3, 4),
           "ms_per_fragment": round(t / region / len(pts) * 1e3, 4), "region_steps": region,
           "ms_per_step_all": [round(v / region * 1e3, 4) for v in sorted(rep)],
           "span": "host float64 point arrays + host images -> (voxelise, pyramid, rulebooks, image branch, 23 convolutions, "
                   "fusion) -> xyz_down float64 and descriptors float32 as host arrays (views of the pinned block), "
                   "PCIe both ways; extract_features_stream(batch=%s, depth=3, copy=False)" % (batch or len(pts)),
           "points": "float32-valued float64 (a PLY's points: uploaded as float32)" if f32_valued else
                     "arbitrary float64 (fixture x 1.7: uploaded as float64)",
           "transfers": "copy engines (hipMemcpyAsync, ROC_CPU_WAIT_FOR_SIGNAL=0)" if streamer.sdma_copies else "copy kernels"}
    if trace:
        k = len(trace)
        out["per_job_ms"] = {
            "device": {"upload": round(sum(r[1] - r[0] for r in trace) / k, 4),
                       "forward": round(sum(r[3] - r[2] for r in trace) / k, 4),
                       "forward_period": round((trace[-1][2] - trace[0][2]) / (k - 1), 4) if k > 1 else None,
                       "download_done_after_forward": round(sum(r[4] - r[3] for r in trace) / k, 4)},
            "host": {n[len("stream_"):]: round(st[n] / jobs, 4) for n in sorted(st) if n.endswith("_ms") and n != "stream_gpu_ms"},
            "note": "HIP events / host clock per job of the LAST repeat's window; forward_period = spacing of consecutive "
                    "forwards' first kernels (the pipeline's steady state: transfers hidden when it equals `forward`)"}
    return out, trace


def sharded_pipeline_leg(model, dev, voxel, rank, world, backend, per_rank=96, distinct=24, passes=3, min_region_s=1.2,
                         barrier=None):
    """SURVEY 8(e) as a measurement: a synthetic test set (`distinct` seeded slabs of the in-tree pair -- scales and sizes
    the same on every rank, float32-valued like a PLY's points -- repeated to `per_rank` x world fragments) is LPT-sharded
    over the ranks (imfnet_amd.dist.shard_fragments); every rank streams ITS fragments through the host-array pipeline
    (xyz_down + descriptors back on the host, as generate_desc needs them) while each fragment's descriptors are ALSO
    copied device-to-device from the capacity bucket into the rank's send buffer (extract_features_stream(device_sink=...));
    then ONE variable-length gather brings all [M_i, 32] blocks to rank 0: one all_gather of the row counts + one grouped
    ncclSend / ncclRecv exchange under RCCL, every receive posted at once (dist.gather_fragment_descriptors(packed=...)) --
    nothing goes through the host on the way.  Rank 0 checks counts, order and every block bit for bit against the CRC
    its producer took from the host copy.
    Reproducibility: capacity keys, lanes and pinned slots are created by two full untimed passes over the
    SAME fragment list (the second is what a timed pass looks like); then `passes` timed passes of >= per_rank fragments
    each, the median is reported, every pass listed."""
    import zlib
    from imfnet_amd import dist as idist
    from imfnet_amd.extract import extract_features, extract_features_stream
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_clouds.npz"))
    im = np.load(os.path.join(ROOT, "tests", "golden", "fixture_images.npz"))
    rng = np.random.default_rng(7)
    protos = []
    for i in range(distinct):
        base = z[f"cloud_bin_{i % 2}"]
        scale = np.float32(rng.uniform(1.0, 1.9))
        d = rng.normal(size=3).astype(np.float32)
        proj = base @ (d / np.linalg.norm(d))
        frac = rng.uniform(0.3, 0.8)
        lo = np.quantile(proj, rng.uniform(0.0, 1.0 - frac))
        keep = np.flatnonzero(proj >= lo)[: int(frac * len(base))]
        pts = (base[np.sort(keep)] * scale).astype(np.float64)           # float32-valued, as a PLY's points
        protos.append((pts, np.transpose(im[f"image_{i % 2}"], (2, 0, 1))[None].copy()))
    n_frag = per_rank * world
    frag = lambda i: protos[i % distinct]                 # fragment i of the test set
    shards = idist.shard_fragments([len(frag(i)[0]) for i in range(n_frag)], world)
    mine = shards[rank]
    coll = idist._collective_device()
    main = None

    def one_pass(sink_buf, crcs):
        """Stream this rank's shard once; returns the rows per fragment (shard order)."""
        rows, at = [], [0]

        def sink(F_dev):                                  # D2D, on the runner's main stream: bucket -> send buffer
            if sink_buf is not None:
                sink_buf[at[0]:at[0] + F_dev.shape[0]].copy_(F_dev, non_blocking=True)
            at[0] += F_dev.shape[0]
        gen = extract_features_stream(model, (frag(i) for i in mine), voxel, dev, batch=2, copy=False, device_sink=sink)
        for i, (xd, Fh) in zip(mine, gen):
            rows.append(Fh.shape[0])
            if crcs is not None:
                crcs[i, 0], crcs[i, 1] = zlib.crc32(Fh.tobytes()), Fh.shape[0]
        return rows

    with torch.no_grad():
        if model.fragment_runner().ratios is None:       # (teach the runner: one fragment on the exact path)
            extract_features(model, frag(0)[0], voxel_size=voxel, device=dev, skip_check=True, image=frag(0)[1])
        rows = one_pass(None, None)                      # untimed: every capacity key of this shard exists afterwards ...
        model.fragment_runner().streamer(dev).fill_lanes()   # ... and every lane of each key
        send_buf = torch.empty((sum(rows), 32), dtype=torch.float32, device=dev)
        crcs = torch.zeros((n_frag, 2), dtype=torch.int64)
        assert one_pass(send_buf, crcs) == rows          # untimed: what a timed pass does, slot for slot (+ the CRCs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_pass(send_buf, None)                         # untimed probe of one pass
        torch.cuda.synchronize()
        # a timed region = `reps` passes over the shard, >= min_region_s long on the slowest rank's probe (same on all ranks)
        reps = torch.tensor([max(1, int(np.ceil(min_region_s / max(time.perf_counter() - t0, 1e-3))))], device=coll)
        if dist.is_initialized():
            dist.all_reduce(reps, op=dist.ReduceOp.MAX)
        reps = int(reps.item())
        st = model.fragment_runner().stats
        redone0 = st.get("redone", 0)
        t_pass = []
        for _ in range(passes):
            if dist.is_initialized():
                (barrier or dist.barrier)()
            t0 = time.perf_counter()
            for _ in range(reps):
                one_pass(send_buf, None)
            torch.cuda.synchronize()
            t_pass.append((time.perf_counter() - t0) / reps)
        redone = st.get("redone", 0) - redone0
        packed = (rows, send_buf if coll.type == "cuda" else send_buf.cpu())   # (gloo test hook: the collective is on the host)
        t_gather = []
        for _ in range(passes):
            if dist.is_initialized():
                (barrier or dist.barrier)()
            t1 = time.perf_counter()
            gathered = idist.gather_fragment_descriptors(None, n_frag, shards, dst=0, device=coll, packed=packed)
            if coll.type == "cuda":
                torch.cuda.synchronize()
            t_gather.append(time.perf_counter() - t1)
    tt = torch.tensor([t_pass, t_gather], dtype=torch.float64, device=coll)
    crcs = crcs.to(coll)
    if dist.is_initialized():
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)        # per pass: the slowest rank
        dist.all_reduce(crcs, op=dist.ReduceOp.MAX)      # every entry is written by exactly one rank, zero elsewhere
    if rank != 0:
        return None
    crcs = crcs.cpu()
    assert gathered is not None and sorted(gathered) == list(range(n_frag)), "gather: fragments missing / out of order"
    total_rows = 0
    for i in range(n_frag):
        blk = gathered[i].cpu().numpy()
        assert blk.shape == (int(crcs[i, 1]), 32), f"gather: fragment {i} has {blk.shape} rows, producer said {int(crcs[i, 1])}"
        assert zlib.crc32(blk.tobytes()) == int(crcs[i, 0]), f"gather: fragment {i} differs from what its rank computed"
        total_rows += blk.shape[0]
    ts, tg = median(tt[0].tolist()), median(tt[1].tolist())
    sizes = [len(p) for p, _ in protos]
    return {"fragments": n_frag, "fragments_per_rank": per_rank, "ranks": world, "backend": backend, "descriptors": total_rows,
            "fragments_per_s": round(n_frag / ts, 1), "descriptors_per_s": round(total_rows / ts, 1),
            "stream_s_max_over_ranks": round(ts, 4), "stream_s_all_passes": [round(v, 4) for v in tt[0].tolist()],
            "timed_region": "%d passes over the shard (%d fragments per rank) = %.2f s per region, %d regions, median" % (reps, reps * per_rank, ts * reps, passes),
            "gather_ms": round(tg * 1e3, 3), "gather_ms_all": [round(v * 1e3, 3) for v in tt[1].tolist()],
            "gather_bytes": total_rows * 128, "fragments_redone_on_the_exact_path": redone,
            "gather": "device-resident: D2D copy from the capacity bucket into the rank's send buffer as each fragment completes, "
                      "one all_gather of row counts + one grouped send / recv exchange (all receives posted at once)",
            "verified": "every block on rank 0: rows and CRC-32 equal to its producer's host copy", "gather_crc_ok": True,
            "note": "host-array pipeline per rank (LPT shards by point count; %d distinct fragments of %d-%d k points, repeated) "
                    "+ one variable-length gather of the [M_i,32] blocks to rank 0; median of %d timed passes after two untimed "
                    "passes over the same list" % (distinct, min(sizes) // 1000, max(sizes) // 1000, passes)}


COMPACT_LIMIT = 4096          # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout: BENCH_r05 was unparseable at 21 KB)


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d and d[k] is not None}


def compact_line(out):
    """The ONE line the driver parses, <= COMPACT_LIMIT bytes: the contract's keys + `roofline` (both rooflines: `frac` against
    the matrix peak named in `peak_def`, `notional_hbm_frac` against 8 TB/s, dominant kernel and whole step) + `cpu_baseline`
    + the host-to-host span and the multi-rank facts.  Everything else (notes, per-kernel tables, probe times, the other
    legs) goes to the full record (`full_record`: bench_full.json, also on stderr)."""
    cfg = out.get("config") or {}
    rf = out.get("roofline") or {}
    cm = cfg.get("capacity_mode") or {}
    r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "rocprof_avg_launch_us",
                   "rocprof_frac", "mfma_busy", "issued_tflops", "launches_per_step", "algorithmic_bytes_per_launch",
                   "algorithmic_flops_per_launch", "all_sparse_conv_ms_per_step", "step_frac", "step_notional_hbm_frac",
                   "isolated_avg_launch_us"))
    if rf:
        r["peak_def"] = rf.get("peak_def")
        r["notional_hbm_frac"] = (rf.get("notional_hbm") or {}).get("frac")
        r["cu_vmem_frac"] = (rf.get("cu_vmem_path") or {}).get("frac")
        r["step_cu_vmem_frac"] = rf.get("step_cu_vmem_frac")
        if rf.get("issued_tflops") and rf.get("achieved"):
            mult = {"bf16x3": 6, "f16x2": 3, "f32": 1}[rf.get("arithmetic", "bf16x3")]
            r["issued_over_useful"] = round(rf["issued_tflops"] / (rf["achieved"] * mult), 3)
        r["profiles"] = rf.get("counters")
    cpu = _pick(out.get("cpu_baseline"), ("value", "unit", "cores", "kind", "value_1_thread"))
    if cpu:
        cpu["sample"] = "one S50k fragment end to end on the host, median of bounded runs (see full record)"
    hs = _pick(out.get("host_span"), ("value", "unit", "ms_per_step", "vs_device_resident", "n_gpus"))
    if hs:
        hs["span"] = "host float64 points + images -> xyz_down + F as host arrays (SURVEY 8d), PCIe both ways"
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline")}
    line["dtype"] = (out.get("dtype") or "").split(" (")[0] + " (%s)" % cfg.get("arithmetic", "")
    line["data"] = out.get("data")
    line["config"] = {"workload": cfg.get("workload"), "arithmetic": cfg.get("arithmetic"),
                      "fragments_per_step": cfg.get("fragments_per_step"),
                      "voxels_per_step_per_gpu": cfg.get("voxels_per_step_per_gpu"),
                      "issue": cm.get("picked", cm.get("mode")), "probe_ms_per_step": cm.get("probe_ms_per_step"),
                      "equals_exact_path_bitwise": cm.get("equals_exact_path_bitwise"),
                      "value_span": "inputs resident in HBM (bench contract); host-to-host span = host_span"}
    t = out.get("timing") or {}
    line["timing"] = _pick(t, ("repeats", "ms_per_step_min", "ms_per_step_max"))
    line["roofline"] = r or None
    line["cpu_baseline"] = cpu or None
    line["host_span"] = hs or None
    e2e = cfg.get("e2e_extract_features") or {}
    legs = {"single_fragment_ms": (cfg.get("single_fragment") or {}).get("ms_per_fragment"),
            "sync_extract_features_ms": e2e.get("ms_per_fragment"),
            "sync_extract_features_device_F_ms": (e2e.get("device_descriptors") or {}).get("ms_per_fragment"),
            "batch_8_descriptors_per_s": (cfg.get("batch_8") or {}).get("descriptors_per_s"),
            "host_span_auto_batch_descriptors_per_s": (cfg.get("host_span_auto_batch") or {}).get("value")}
    line["legs"] = {k: v for k, v in legs.items() if v is not None} or None
    ar = {}
    for name, a in (out.get("arithmetics") or {}).items():
        ra = a.get("roofline") or {}
        ar[name] = {"value": a.get("value"), "ms_per_step": a.get("ms_per_step"), "kernel": ra.get("kernel"),
                    "frac": ra.get("frac"), "peak": ra.get("peak"), "notional_hbm_frac": (ra.get("notional_hbm") or {}).get("frac")}
    line["arithmetics"] = ar or None
    sp = cfg.get("sharded_pipeline")
    if sp:
        line["sharded_pipeline"] = _pick(sp, ("ranks", "backend", "fragments", "fragments_per_s", "descriptors_per_s", "gather_ms",
                                              "gather_bytes", "gather_crc_ok"))
    line["rccl"] = out.get("rccl")
    line["full_record"] = out.get("full_record")
    # never exceed the limit: drop the optional groups, least important first
    for victim in ("legs", "arithmetics", "sharded_pipeline", "timing", "host_span"):
        if len(json.dumps(line)) <= COMPACT_LIMIT:
            break
        line.pop(victim, None)
    return line


def emit(out, full_path, seal_stdout=False):
    """Full record -> `full_path` (+ gpurun_out/ when that directory exists) and stderr; the compact line is the LAST stdout line."""
    out["full_record"] = os.path.relpath(full_path, ROOT) if full_path else None
    full = json.dumps(out)
    if full_path:
        targets = [full_path]
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            targets.append(os.path.join(ROOT, "gpurun_out", os.path.basename(full_path)))
        for p in targets:
            try:
                with open(p, "w") as f:
                    f.write(full + "\n")
            except OSError as e:                         # a read-only tree must not cost the line
                print("bench: could not write %s: %s" % (p, e), file=sys.stderr)
    print("bench full record: " + full, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(out))
    assert len(line) <= COMPACT_LIMIT + 1024, len(line)
    sys.stdout.flush()
    try:                                                 # native libraries' buffered stdout (RCCL's banner) goes out BEFORE the line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                    # noqa: BLE001
        pass
    print(line, flush=True)
    # ... and nothing may follow it: whatever a native library still prints to stdout (at exit, from a destructor) is dropped
    if seal_stdout:
        try:
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        except OSError:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.7)
    ap.add_argument("--voxel", type=float, default=0.025)
    ap.add_argument("--batch", type=int, default=2, choices=(1, 2, 4, 8),
                    help="fragments per forward: 2 = the in-tree fragment PAIR (cloud_bin_0 + cloud_bin_1, one image "
                         "each) as ONE batched sparse tensor, the batched call of model/resunet.py:241-250; 4 / 8: the pair "
                         "repeated (IMF_MAX_BATCH = 8)")
    ap.add_argument("--mode", default="auto", choices=("auto", "capacity", "pipelined", "graph", "exact"),
                    help="capacity: imf_fragment_forward per step (device-side counts, no host readback), one bucket: every step "
                         "queues behind the previous one; pipelined: the same launches over TWO buckets, step k + 1's head "
                         "(table reset, level-0 pyramid, image fork) on the side stream under step k's decoder, as the streaming "
                         "pipeline issues consecutive forwards; graph: one hipGraph replay per step; auto (default): all three "
                         "are timed after the clocks have settled and the fastest runs the timed region "
                         "(config.capacity_mode.picked says which, probe_ms_per_step has all of them); exact: count readback + "
                         "native executor")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is repeated this many times; "
                         "ms_per_step / value are the MEDIAN repeat, min and max are reported beside it")
    ap.add_argument("--settle-ms", type=float, default=600.0,
                    help="untimed steps run for at least this long before the first timed region (clock / power settle)")
    ap.add_argument("--trace-steps", type=int, default=3,
                    help="steps AFTER the timed regions that carry HIP events around every convolution (live roofline)")
    ap.add_argument("--arith", default=None, choices=tuple(ARITH),
                    help="arithmetic of the convolutions for the headline (`value`): bf16x3 (default: fp32 operands split exactly "
                         "into three bf16 parts), f16x2 (the 22-bit FAST mode), f32 (fp32 MFMA).  The other two are measured "
                         "too and reported with their own roofline under `arithmetics` (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-fragment / batch / fp32-MFMA / graph legs")
    ap.add_argument("--no-host-span", action="store_true", help="skip the host-array stream leg (profiling runs)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-pipeline + gather leg")
    ap.add_argument("--sharded-per-rank", type=int, default=96, help="fragments per rank of the sharded-pipeline leg")
    ap.add_argument("--sharded-region-s", type=float, default=1.2, help="minimum length of one timed region of that leg")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the FULL record goes (every leg, note and per-kernel table); stdout carries only the compact line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from imfnet_amd import dist as idist
    from imfnet_amd import ops
    if args.arith is None:
        args.arith = VARIANT_NAME.get(ops.CONV_VARIANT, "bf16x3")
    ops.CONV_VARIANT = ARITH[args.arith]["variant"]      # process-wide: every leg below runs on the headline's arithmetic

    # test hooks (single-GPU box): IMF_DIST_BACKEND=gloo IMF_FORCE_DEVICE=0 run N ranks on one device
    backend = os.environ.get("IMF_DIST_BACKEND", "nccl")
    if "IMF_FORCE_DEVICE" in os.environ:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
        torch.cuda.set_device(int(os.environ["IMF_FORCE_DEVICE"]))
    rank, world, local = idist.init_from_env(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if rank != 0:                                        # only rank 0 owns stdout (one JSON line, last): the others' native
        try:                                             # libraries (RCCL's banner) must not write behind it
            sys.stdout.flush()
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        except OSError:
            pass
    if "IMF_FORCE_DEVICE" in os.environ:
        local = int(os.environ["IMF_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    xyz1, img1, voxel = load_workload(args.scale, args.voxel)
    model, sd = build_model(dev)
    if args.batch >= 2:
        pts, imgs = load_pair(args.scale)
        pts, imgs = pts * (args.batch // 2), np.concatenate([imgs] * (args.batch // 2), 0)
    else:
        pts, imgs = [xyz1], img1
    wl = Workload(model, dev, pts, imgs, voxel)

    def barrier():
        if dist.is_initialized():                         # (also ONE rank with a process group: IMF_DIST_FORCE_INIT=1, the RCCL test)
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()

    sync = torch.cuda.synchronize
    graph_info = None
    with torch.no_grad():
        dyn = args.mode != "exact"
        F_exact = wl.prepare_graph(replicate=True, pipelined=True).clone() if dyn else None
        step = (lambda tl=None: wl.graph_step(tl)) if dyn else (lambda tl=None: wl.exact_step())
        if dyn:
            wl.runner.use_graph = args.mode == "graph"
            wl.pipelined = args.mode == "pipelined"
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
        if args.mode == "auto":                           # eager capacity-mode launches (one bucket / two, pipelined) vs ONE
            probe = {}                                    # hipGraph replay: measure, pick
            for rnd in range(2):                          # interleaved rounds; the capture happens in round 0's warm-up
                for name, ug, pl in (("capacity", False, False), ("pipelined", False, True), ("graph", True, False)):
                    wl.runner.use_graph, wl.pipelined = ug, pl
                    for _ in range(4):
                        step()
                    sync()
                    probe.setdefault(name, []).append(timed(step, max(10, args.steps), sync) * 1e3)
            best = min(probe, key=lambda k: min(probe[k]))
            wl.runner.use_graph, wl.pipelined = best == "graph", best == "pipelined"
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
            graph_info = {"hipgraph_replay": bool(wl.runner.use_graph), "pipelined_over_two_buckets": bool(wl.pipelined),
                          "graph_nodes": wl.bucket.n_nodes,
                          "equals_exact_path_bitwise": same, "host_readbacks_per_step": 0,
                          "capacities": {"points": wl.bucket.caps.n_points, "rows": list(wl.bucket.caps.rows)},
                          "input_replicas": len(wl.replicas), **picked}
        else:
            out = step()
            sync()
            M = out.shape[0]
            F_ref = out.clone()

        # ---- the timed regions: EXACTLY --steps steps each, barrier + synchronize on both sides, nothing else inside
        # (no event records, no readbacks).  Fragments are independent units: no data-path collective (each rank
        # would write its own <frag>.npz; the gather of the sharded leg below is measured on its own).
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

        # every step recomputes the same points (from another replica): the last timed step must reproduce the warm-up step
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

        # ---- SURVEY 8(d)'s span, timed the same way on every rank (host arrays in, descriptors on the host)
        host_span = host_trace = None
        if not args.no_host_span and dyn:
            pts2, imgs2 = load_pair(args.scale)
            host_span, host_trace = host_span_leg(model, dev, args, barrier, sync, pts2, imgs2, voxel)
        sharded = None
        if not args.no_sharded and dyn:
            sharded = sharded_pipeline_leg(model, dev, voxel, rank, world, backend, per_rank=args.sharded_per_rank,
                                           min_region_s=args.sharded_region_s, barrier=barrier)

    t = torch.tensor(rep, dtype=torch.float64, device=dev)
    hs = torch.tensor([host_span["ms_per_step"] if host_span else 0.0, float(M)], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per repeat: the slowest rank
        m = torch.tensor([M], dtype=torch.int64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        total_m = int(m.item())
        hmax = hs.clone()
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        if host_span:                                     # whole job: all ranks' descriptors over the slowest rank's time
            host_span["ms_per_step"] = round(float(hmax[0]), 4)
            host_span["value"] = round(total_m * 1e3 / float(hmax[0]), 1)
            host_span["n_gpus"] = world
    else:
        total_m = M
    # what every rank measured (median of its own repeats) and the world the process group really has
    mine_ms = torch.tensor([median(sorted(rep)) / args.steps * 1e3], dtype=torch.float64, device=dev)
    per_rank = [mine_ms.clone() for _ in range(world)]
    if dist.is_initialized():
        dist.all_gather(per_rank, mine_ms)
    rccl_info = {"backend": backend if dist.is_initialized() else None, "process_group": dist.is_initialized(),
                 "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                 "per_rank_ms_per_step": [round(float(v), 4) for v in per_rank],
                 "gather_crc_ok": bool(sharded and sharded.get("gather_crc_ok")) if sharded is not None else None}
    rep = sorted(float(v) for v in t.tolist())
    elapsed = median(rep)

    final = None
    if rank == 0:
        # ---- live roofline of the dominant kernel (HIP events on the launch stream) -------------
        groups = group_trace(trace, args.arith)
        extras, arithmetics = {}, None
        with torch.no_grad():
            iso_groups = None
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
                iso_groups = group_trace(iso)
            roofline = build_roofline(groups, args.arith, traced_steps, elapsed / args.steps * 1e3, iso_groups)
            if world == 1 and not args.no_extras:
                extras = extra_legs(model, dev, args, sync)
                arithmetics = other_arithmetics(dev, args, sync, F_ref, pipelined=bool(wl.pipelined))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(xyz1, img1, voxel, sd)
        n_pts = int(wl.xyz.shape[0])
        if host_span:
            host_span["vs_device_resident"] = round(host_span["ms_per_step"] / (elapsed / args.steps * 1e3), 3)
            if host_trace:                               # a window of consecutive jobs: ms since the pipeline's creation (device clock)
                host_span["timeline_ms"] = {"columns": ["upload_begin", "upload_end", "forward_begin", "forward_end", "download_end"],
                                            "jobs": [[round(v, 3) for v in r[:5]] for r in host_trace[-6:]]}
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
            "scaling": "weak", "vs_baseline": None,
            "dtype": ARITH[args.arith]["dtype"],
            "data": ("synthetic (reference fixture fragment%s scaled x%.2f, seeded random weights)"
                     % (" PAIR cloud_bin_0 + cloud_bin_1" + (" x%d" % (args.batch // 2) if args.batch > 2 else "")
                        if args.batch >= 2 else " cloud_bin_0", args.scale)),
            "host_span": host_span,
            "config": {"workload": (f"3DMatch-shaped fragment pair: {n_pts} points -> {M} voxels @ "
                                    f"{voxel * 100:.1f} cm, one 120x160 image each, ResUNetBN2C 32-D, conv1 k5; the {args.batch} fragments are "
                                    f"ONE batched forward per step per GPU, geometry rebuilt every step"
                                    if args.batch >= 2 else
                                    f"3DMatch-shaped fragment: {n_pts} points -> {M} voxels @ "
                                    f"{voxel * 100:.1f} cm, image 120x160, ResUNetBN2C 32-D, conv1 k5; "
                                    f"one fragment per step per GPU, geometry rebuilt every step"),
                       "value_span": "inputs resident in HBM when the timed region starts, as the bench contract defines `value` "
                                     "(a different replica of the points every step: no step reads its input from the Infinity "
                                     "Cache); SURVEY 8(d)'s host-to-host span is `host_span` of this line, timed the same way",
                       "voxels_per_step_per_gpu": M, "points_per_step_per_gpu": n_pts,
                       "last_step_equals_warmup_bitwise": bit_reproducible,
                       "fragments_per_step": world * args.batch,
                       "execution": {"capacity": "one imf_fragment_forward call per step in capacity mode: device-side row counts, "
                                                 "no host readback, launches issued natively on three streams",
                                     "pipelined": "one imf_fragment_forward call per step in capacity mode (device-side row counts, no "
                                                  "host readback, launches issued natively on three streams); consecutive steps "
                                                  "alternate between TWO capacity buckets and a step's head (table reset, level-0 "
                                                  "pyramid, image fork: ~60 us) is issued on the side stream, under the previous "
                                                  "step's decoder -- the order in which the streaming pipeline issues consecutive "
                                                  "forwards (imf_fragment_io.head_on_side); every step is complete when the timed "
                                                  "region ends; the one-bucket step time is capacity_mode.probe_ms_per_step.capacity",
                                     "graph": "one hipGraph replay per step of imf_fragment_forward in capacity mode",
                                     "exact": "exact mode: row-count readback + native executor (imf_resunet_forward)"}[run_mode],
                       "capacity_mode": graph_info,
                       "image_branch": model.image_branch_mode if not dyn else wl.runner.image_branch_mode,
                       "arithmetic": args.arith, "conv_arithmetic": ARITH[args.arith]["arithmetic"],
                       "sharded_pipeline": sharded,
                       **extras},
            "roofline": roofline, "cpu_baseline": cpu,
            # the same pair step on the other two arithmetics, each timed and rooflined like the headline (N = 1)
            "arithmetics": arithmetics,
        }
        out["rccl"] = rccl_info
        final = out
    # the process group goes FIRST: whatever RCCL has to say (its version banner sits in the C library's stdout buffer until
    # the process ends -- found in round 6: it came out AFTER the JSON line) is out before the line that must be last
    if dist.is_initialized():
        dist.destroy_process_group()
    if final is not None:
        emit(final, args.full_out, seal_stdout=True)


def extra_legs(model, dev, args, sync):
    """Numbers the headline does not show (each a few hundred ms of GPU time):
      single_fragment        one S50k fragment per step (the reference harness is one fragment per call)
      batch_4 / batch_8      the pair twice / four times in ONE forward (IMF_MAX_BATCH = 8): the coarse levels fill the chip
      e2e_extract_features   SURVEY 8(d)'s span, one synchronous call at a time
      host_span_f32_valued   the stream leg on float32-valued points (what a PLY holds): uploaded as float32
    (the other arithmetics -- f16x2 fast mode, strict fp32 -- are `arithmetics` of the line: other_arithmetics below)"""
    from imfnet_amd.extract import extract_features
    xyz1, img1, voxel = load_workload(args.scale, args.voxel)
    out = {}
    wl1 = Workload(model, dev, [xyz1], img1, voxel)
    wl1.prepare_graph()
    wl1.runner.use_graph = False
    for _ in range(3):
        wl1.graph_step()
    dt = timed(wl1.graph_step, 20, sync)
    m1 = wl1.last_res.counts[0]
    out["single_fragment"] = {"descriptors_per_s": round(m1 / dt, 1), "ms_per_fragment": round(dt * 1e3, 4), "voxels": m1,
                              "execution": "capacity mode, one fragment per forward"}
    dt = timed(wl1.exact_step, 20, sync)
    out["single_fragment"]["exact_mode_ms_per_fragment"] = round(dt * 1e3, 4)
    del wl1

    pts2, imgs2 = load_pair(args.scale)
    for nb in (4, 8):
        if nb == args.batch:
            continue
        wlb = Workload(model, dev, pts2 * (nb // 2), np.concatenate([imgs2] * (nb // 2), 0), voxel)
        wlb.prepare_graph()
        wlb.runner.use_graph = False
        for _ in range(3):
            r = wlb.graph_step()
        sync()
        if r.flags:                                       # (the batch's bit grid: sized from this batch's own box, once)
            wlb.runner.observe_batch(nb, r.bbox)
            wlb.prepare_graph()
            for _ in range(3):
                r = wlb.graph_step()
            sync()
        dt = timed(wlb.graph_step, 12, sync)
        mb = wlb.last_res.counts[0]
        out["batch_%d" % nb] = {"descriptors_per_s": round(mb / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                                "ms_per_fragment": round(dt * 1e3 / nb, 4), "voxels": mb, "flags": wlb.last_res.flags,
                                "execution": "capacity mode, %d fragments (the pair x%d) per forward" % (nb, nb // 2)}
        del wlb

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
    per_dev = []
    for i in range(28):                               # the reference's own return: F stays a device tensor (util/misc.py:100-104)
        t0 = time.perf_counter()
        xd, Fd = extract_features(model, xyz_host, voxel_size=voxel, device=dev, skip_check=True, image=img1, host_descriptors=False)
        if i >= 3:
            per_dev.append(time.perf_counter() - t0)
    sync()
    assert torch.equal(Fd.cpu(), torch.as_tensor(np.asarray(Fh))) and not hasattr(Fd, "host")
    per_dev.sort()
    out["e2e_extract_features"] = {"descriptors_per_s": round(Fh.shape[0] / dt, 1), "ms_per_fragment": round(dt * 1e3, 3),
                                   "device_descriptors": {"ms_per_fragment": round(per_dev[len(per_dev) // 2] * 1e3, 3),
                                                          "ms_per_fragment_min_max": [round(per_dev[0] * 1e3, 3), round(per_dev[-1] * 1e3, 3)],
                                                          "span": "the same call with host_descriptors=False: xyz_down on the host, F "
                                                                  "left on the device (the reference's return); the call returns "
                                                                  "when xyz_down has arrived"},
                                   "ms_per_fragment_min_max": [round(per_call[0] * 1e3, 3), round(per_call[-1] * 1e3, 3)],
                                   "span": "extract_features(host float64 points [%d,3] + host image) -> xyz_down on the host, "
                                           "and F on the host (one job of the pipeline: stage, upload, forward, download), one "
                                           "fragment at a time, synchronous" % len(xyz_host),
                                   "runner": {k: v for k, v in model.fragment_runner().stats.items() if not k.startswith("stream_")}}

    hs32, _ = host_span_leg(model, dev, args, lambda: None, sync, pts2, imgs2, voxel, f32_valued=True)
    hs32.pop("per_job_ms", None)
    out["host_span_f32_valued"] = hs32
    import copy
    a2 = copy.copy(args)
    a2.steps = max(args.steps, 60)                     # (five fragments per forward: enough forwards per timed region)
    hsa, _ = host_span_leg(model, dev, a2, lambda: None, sync, pts2, imgs2, voxel, batch="auto")
    hsa.pop("per_job_ms", None)
    hsa["note"] = ("the same stream with batch='auto': fragments grouped per forward up to extract.POINT_BUDGET points (five of "
                   "these); ms_per_step is still per PAIR of fragments")
    out["host_span_auto_batch"] = hsa

    wl2 = Workload(model, dev, pts2, imgs2, voxel)
    wl2.prepare_graph()
    r = wl2.runner
    prev = r.use_graph
    r.use_graph = True
    for _ in range(3):
        wl2.graph_step()
    dt = timed(wl2.graph_step, 20, sync)
    out["graph_replay"] = {"descriptors_per_s": round(wl2.last_res.counts[0] / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                           "graph_nodes": wl2.bucket.n_nodes,
                           "note": "the pair as ONE hipGraph replay per step (same launches, bit-identical descriptors)"}
    r.use_graph = prev
    del wl2

    return out


def other_arithmetics(dev, args, sync, F_headline, pipelined=False):
    """The same pair step on the arithmetics the headline does NOT run, each measured like the headline: capacity mode
    (imf_fragment_forward, no readback), exactly --steps steps between synchronize per repeat, median of 5 repeats after a
    settle, input replicas beyond the Infinity Cache; then three traced steps -> its own `roofline` (dominant kernel, useful
    TFLOP/s against the arithmetic's matrix peak, notional HBM fraction, counters / rocprofv3 averages from
    profiles/r05_*_<arith>.*).  Capacity mode is asserted bit-identical to the arithmetic's own exact mode."""
    from imfnet_amd import ops
    pts2, imgs2 = load_pair(args.scale)
    res = {}
    prev = ops.CONV_VARIANT
    for name, A in ARITH.items():
        if name == args.arith:
            continue
        ops.CONV_VARIANT = A["variant"]
        try:
            m, _ = build_model(dev, variant=A["variant"])
            wl = Workload(m, dev, pts2 * (args.batch // 2) if args.batch >= 2 else pts2[:1],
                          np.concatenate([imgs2] * (args.batch // 2), 0) if args.batch >= 2 else imgs2[:1], args.voxel)
            F_exact = wl.prepare_graph(replicate=True, pipelined=pipelined).clone()
            assert wl.runner.variant == A["variant"]
            wl.runner.use_graph = False
            t_s = time.perf_counter()
            while (time.perf_counter() - t_s) * 1e3 < min(args.settle_ms, 300.0):
                for _ in range(10):
                    r = wl.graph_step()
                sync()
            assert r.flags == 0 and torch.equal(r.F, F_exact), "%s: capacity mode differs from its exact mode" % name
            rep = sorted(timed(wl.graph_step, args.steps, sync) for _ in range(5))
            dt = rep[len(rep) // 2]
            M = wl.last_res.counts[0]
            tr = []
            for _ in range(3):
                wl.graph_step(tr)
            sync()
            res[name] = {"value": round(M / dt, 1), "unit": "descriptors/s", "ms_per_step": round(dt * 1e3, 4),
                         "ms_per_step_all": [round(v * 1e3, 4) for v in rep], "steps": args.steps,
                         "dtype": A["dtype"], "conv_arithmetic": A["arithmetic"],
                         "equals_exact_mode_bitwise": True,
                         "max_abs_diff_vs_headline_descriptors": float((r.F - F_headline).abs().max()),
                         "roofline": build_roofline(group_trace(tr, name), name, 3, dt * 1e3)}
            del m, wl
        finally:
            ops.CONV_VARIANT = prev
    return res


if __name__ == "__main__":
    main()
