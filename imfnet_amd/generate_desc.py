#!/usr/bin/env python3
"""Batch descriptor generation over a 3DMatch-layout tree -- the reference's
scripts/generate_desc.py with the same flags, directory contract and NPZ layout:

    python -m imfnet_amd.generate_desc --source <3DMatch_test> --target <desc_dir> -m <checkpoint.pth>

    <source>/<scene>/seq-01/cloud_bin_K.ply + cloud_bin_K_0.png|jpg
 -> <target>/<scene>/seq-01/cloud_bin_K.npz  {points f64[N,3], xyz f64[M,3], feature f32[M,32]}

Differences, on purpose: the per-fragment timer synchronises the device (the
reference's does not, so it under-reports); `--voxel_size` is still ignored in favour of the
checkpoint's config (kept: generate_desc.py:146,186); list.txt is not written-then-deleted.
Launched under torchrun (one process per GPU) the fragments are sharded across ranks
longest-first; every rank writes its own NPZ files, and with --gather the descriptors are first
collected on rank 0 over RCCL and written there.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

from . import dist as idist
from .checkpoint import Config, load_checkpoint
from .dataio import image_to_nchw, process_image, read_image, read_ply_points, save_descriptors
from .extract import extract_features
from .files import ensure_dir, get_file_list, get_folder_list
from .model import load_model


def list_fragments(source_path):
    """[(scene_dir, ply_path)] in the reference's order (generate_desc.py:65-80)."""
    folders = get_folder_list(source_path)
    assert len(folders) > 0, f"Could not find 3DMatch folders under {source_path}"
    frags = []
    for scene in folders:
        if "evaluation" in scene:
            continue
        for fi in get_file_list(os.path.join(scene, "seq-01"), ".ply"):
            frags.append((scene, fi))
    return frags


def load_fragment(ply_path, config):
    xyz = read_ply_points(ply_path)
    image_file = ply_path.replace(".ply", "_0.png")
    if not os.path.exists(image_file):
        image_file = ply_path.replace(".ply", "_0.jpg")
    img = read_image(image_file)
    if img.shape[0] != config.image_H or img.shape[1] != config.image_W:
        img = process_image(image=img, aim_H=config.image_H, aim_W=config.image_W)
    return xyz, image_to_nchw(img)


def _batch_pipelined(model, runner, config, jobs, target_path, voxel_size, device, workers, depth=None, batch_points=None):
    """The batch loop without the host in the GPU's way (SURVEY 8 f-4 / 8e "bound by host-side decode"): loader threads
    decode PLY + PNG; the main thread only stages each fragment into a pinned HostSlot and hands it to the library's
    streaming pipeline (stream.py: upload kernel, capacity-mode forward, xyz_down = xyz[inds] gathered on the device,
    download kernel -- issued by the pipeline's worker thread, transfers under the neighbouring forwards); ONE thread waits
    for the jobs in order and fans the NPZ writes (straight from the slot's pinned views) out to the writer pool.
    Consecutive fragments share a forward until their points reach extract.POINT_BUDGET (the model's batched call,
    model/resunet.py:241-250: the stride-4 / 8 levels of one fragment leave half the chip idle -- 0.50 ms per S50k fragment
    in pairs, 0.40 in fives).  A fragment the runner flags (capacity / f16 range) is redone by the caller on the exact path.
    Returns (seconds per fragment, redo list)."""
    import queue
    import threading
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    from .model.graph import HostSlot
    depth = depth or max(4, 2 * workers)
    pool = getattr(runner, "cli_slots", None)           # pinned blocks are expensive to create (~7 ms each): kept with the runner
    if pool is None:
        pool = runner.cli_slots = []
    while len(pool) < depth:
        pool.append(HostSlot())
    slots = queue.Queue()
    for sl in pool[:depth]:
        slots.put(sl)
    streamer = runner.streamer(device)
    runner.main_stream(device).wait_stream(torch.cuda.current_stream(device))
    loader, writer = ThreadPoolExecutor(max_workers=workers), ThreadPoolExecutor(max_workers=workers)
    times, redo, writes, failure = {}, [], [], []
    done_q = queue.Queue()

    def load(job):
        scene, fi = job
        return read_ply_points(fi), np.ascontiguousarray(load_image(fi, config), dtype=np.float32)

    def write(job, slot, xyz, r0, n0, v, left):
        scene, fi = job
        try:
            out_dir = os.path.join(target_path, os.path.basename(scene), "seq-01")
            ensure_dir(out_dir)
            save_descriptors(os.path.join(out_dir, os.path.basename(fi).replace(".ply", ".npz")), xyz, v["sel"][r0:r0 + n0],
                             v["F"][r0:r0 + n0])
        finally:
            with left[1]:
                left[0] -= 1
                last = left[0] == 0
            if last:                                    # the slot's last fragment is on disk
                slots.put(slot)

    def waiter():
        # ONE thread waits for the jobs, in submission order (many threads blocked in hipEventSynchronize slowed the
        # launches ~5x: measured 2.7 vs 0.5 ms per enqueue); the NPZ writes fan out to the pool.  It stays ONE JOB BEHIND
        # the submissions: waiting for job k ends the deferral of its download (csrc/pipeline.hip), which belongs behind
        # job k+1's launches.  Whatever goes wrong here is recorded, every slot still comes back, and the queue is
        # drained so that the main thread never blocks on it.
        held = None
        while True:
            item = done_q.get()
            item, held = held, item
            if item is None:
                if held is None:
                    return
                continue
            group, sj = item
            try:
                if failure:
                    raise failure[0]
                res = sj.wait()
                if res.flags:
                    if len(group) > 1 and (res.flags & 4):      # the batch's bounding box outgrew the grid: size the next one by it
                        runner.observe_batch(len(group), res.bbox)
                    redo.extend(job for job, _ in group)
                    slots.put(sj.slot)
                else:
                    spans = res.items() if len(group) > 1 else [(0, res.counts[0])]
                    left = [len(group), threading.Lock()]
                    for (job, xyz), (r0, n0) in zip(group, spans):
                        # GPU time of the fragment's forward: first to last kernel (device stamps forward_begin / forward_end
                        # of imf_pipeline_wait) -- what the reference's print times (scripts/generate_desc.py:99-110), not the
                        # pipeline latency (queueing behind the previous forward, the deferred download): ADVICE r4
                        times[job[1]] = max(sj.stamps[3] - sj.stamps[2], 0.0) * 1e-3 / len(group)
                        writes.append(writer.submit(write, job, sj.slot, xyz, r0, n0, sj.views, left))
            except BaseException as e:                  # noqa: BLE001 -- re-raised by the main thread after the join
                if not failure:
                    failure.append(e)
                slots.put(sj.slot)
            if held is None:                            # the sentinel: that was the last job
                return

    wt = threading.Thread(target=waiter, daemon=True)
    wt.start()
    todo, inflight = deque(jobs), deque()

    def top_up():
        while todo and len(inflight) < depth:
            job = todo.popleft()
            inflight.append((job, loader.submit(load, job)))

    from .extract import POINT_BUDGET
    from ._lib import MAX_BATCH
    budget = POINT_BUDGET if batch_points is not None and batch_points < 0 else int(batch_points or 0)   # 0: one fragment per forward

    def submit(group):
        """One forward for `group` = [(job, xyz, image)]; fragments the runner cannot take go to the redo list."""
        while True:                                         # blocks only when `depth` jobs are on the GPU / being written
            try:
                slot = slots.get(timeout=5.0)
                break
            except queue.Empty:
                if failure or not wt.is_alive():
                    raise failure[0] if failure else RuntimeError("generate_desc: the waiter thread died")
        sj = streamer.submit([(xyz, image) for _, xyz, image in group], voxel_size, slot, more_follow=True)
        if sj is None:                                      # no capacities known (e.g. the bit grid never fitted): exact path
            redo.extend(job for job, _, _ in group)
            slots.put(slot)
            return
        done_q.put(([(job, xyz) for job, xyz, _ in group], sj))
        if slot.grown:                                      # a new size: grow every FREE slot now (HostSlot.reserve)
            idle = []
            while True:
                try:
                    idle.append(slots.get_nowait())
                except queue.Empty:
                    break
            for other in idle:
                other.reserve_like(slot, device)
                slots.put(other)

    try:
        top_up()
        group = []
        while inflight and not failure:
            job, fut = inflight.popleft()
            xyz, image = fut.result()
            top_up()
            if group and (xyz.dtype != group[0][1].dtype or image.shape != group[0][2].shape):
                submit(group)
                group = []
            group.append((job, xyz, image))
            if len(group) >= MAX_BATCH or sum(len(x) for _, x, _ in group) >= budget:
                submit(group)
                group = []
        if group and not failure:
            submit(group)
    finally:
        done_q.put(None)
        wt.join()
        for _, fut in inflight:
            fut.cancel()
        loader.shutdown()
        errs = []
        for w in writes:
            try:
                w.result()
            except BaseException as e:                          # noqa: BLE001
                errs.append(e)
        writer.shutdown()
        torch.cuda.current_stream(device).wait_stream(runner.main_stream(device))
    if failure or errs:
        raise (failure or errs)[0]
    return [times[fi] for _, fi in jobs if fi in times], redo


def _ply_count(path):
    from . import _lib
    n = _lib.lib().imf_ply_vertex_count(os.fsencode(path))
    if n < 0:
        raise ValueError(f"{path}: {_lib.lib().imf_last_error().decode()}")
    return n


def load_image(ply_path, config):
    image_file = ply_path.replace(".ply", "_0.png")
    if not os.path.exists(image_file):
        image_file = ply_path.replace(".ply", "_0.jpg")
    img = read_image(image_file)
    if img.shape[0] != config.image_H or img.shape[1] != config.image_W:
        img = process_image(image=img, aim_H=config.image_H, aim_W=config.image_W)
    return image_to_nchw(img)


def extract_features_batch(model, config, source_path, target_path, voxel_size, device, gather=False, workers=4,
                           batch_points=None):
    """scripts/generate_desc.py:44-133.  The reference decodes, computes and writes one fragment at a
    time; at ~1 ms of GPU work per fragment the PLY/PNG decode (tens of ms) and the zlib write
    (~0.1 s) would leave the GPU idle, so `workers` loader threads run ahead of the GPU and as many
    writer threads take the device-to-host copy + `savez_compressed` behind it (numpy / PIL / zlib
    release the GIL).  workers=0 is the reference's sequential order; outputs are identical, whatever the
    workers and however the fragments are sharded over ranks.  batch_points > 0 (opt-in; -1 = extract.POINT_BUDGET): the
    pipelined loop puts consecutive fragments into one forward until their points reach it (the model's batched call,
    model/resunet.py:241-250: +25 % GPU throughput at ~5 fragments per forward) -- points and xyz stay identical, descriptors
    agree to rounding (<= 2e-6: a row's partial sums are grouped by its 64-row tile's active offsets, and its tile-mates
    differ in a batch), so they then depend on the grouping."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) \
        if torch.distributed.is_initialized() else (0, 1)
    frags = list_fragments(source_path)
    shards = idist.shard_fragments([os.path.getsize(f) for _, f in frags], world)
    model.eval()
    times, results, meta = [], {}, {}
    mine = list(shards[rank])
    runner = model.fragment_runner() if hasattr(model, "fragment_runner") else None
    if runner is not None and workers > 0 and not (gather and world > 1) and len(mine) > 1:
        # first fragment on the exact path (teaches the runner the voxel-per-point ratios), the rest pipelined
        pending = [frags[i] for i in mine]
        t_all = []

        def run_exact(job):
            scene, fi = job
            xyz, image = load_fragment(fi, config)
            torch.cuda.synchronize(device)
            t0 = time.time()
            xyz_down, feature = extract_features(model, xyz=xyz, rgb=None, normal=None, voxel_size=voxel_size,
                                                 device=device, skip_check=True, image=image)
            torch.cuda.synchronize(device)
            t_all.append(time.time() - t0)
            out_dir = os.path.join(target_path, os.path.basename(scene), "seq-01")
            ensure_dir(out_dir)
            save_descriptors(os.path.join(out_dir, os.path.basename(fi).replace(".ply", ".npz")), xyz, xyz_down, feature)

        run_exact(pending.pop(0))
        # re-fetched AFTER the head fragment: an fp32 recompute or a refresh rebuilds the plans the runner points into
        runner = model.fragment_runner()
        if runner is not None and runner.ratios is not None and runner.grid_words > 0:
            t_pipe, redo = _batch_pipelined(model, runner, config, pending, target_path, voxel_size, device, workers,
                                            batch_points=batch_points)
            t_all += t_pipe
            pending = redo                                  # flagged fragments (capacity / f16 range): exact path
        for job in pending:                                 # also everything, when the runner turned out unusable
            run_exact(job)
        return t_all, len(frags)
    loader = ThreadPoolExecutor(max_workers=workers) if workers > 0 else None
    writer = ThreadPoolExecutor(max_workers=workers) if workers > 0 else None
    pending, writes, nxt = deque(), [], 0

    def submit_loads():
        nonlocal nxt
        while loader is not None and nxt < len(mine) and len(pending) < 2 * workers:
            pending.append(loader.submit(load_fragment, frags[mine[nxt]][1], config))
            nxt += 1

    def write(out_dir, out_file, xyz, xyz_down, feature, done):
        done.synchronize()                               # the descriptors of THIS fragment are complete
        ensure_dir(out_dir)
        save_descriptors(out_file, xyz, xyz_down, feature)

    submit_loads()
    for n_done, i in enumerate(mine):
        scene, fi = frags[i]
        if loader is not None:
            xyz, image = pending.popleft().result()
            submit_loads()
        else:
            xyz, image = load_fragment(fi, config)
        torch.cuda.synchronize(device)
        t0 = time.time()
        xyz_down, feature = extract_features(model, xyz=xyz, rgb=None, normal=None, voxel_size=voxel_size,
                                             device=device, skip_check=True, image=image)
        torch.cuda.synchronize(device)
        times.append(time.time() - t0)
        out_dir = os.path.join(target_path, os.path.basename(scene), "seq-01")
        out_file = os.path.join(out_dir, os.path.basename(fi).replace(".ply", ".npz"))
        if gather and world > 1:
            results[i], meta[i] = feature, (out_dir, out_file, xyz, xyz_down)
        elif writer is not None:
            done = torch.cuda.Event()
            done.record()
            writes.append(writer.submit(write, out_dir, out_file, xyz, xyz_down, feature, done))
            while len(writes) > 4 * workers:             # bound the host memory held by queued writes
                writes.pop(0).result()
        else:
            ensure_dir(out_dir)
            save_descriptors(out_file, xyz, xyz_down, feature)
    for w in writes:
        w.result()
    for pool in (loader, writer):
        if pool is not None:
            pool.shutdown()
    if gather and world > 1:
        # one exchange: descriptors to rank 0 (coordinates are re-derived there from the files' points)
        all_feats = idist.gather_fragment_descriptors(results, len(frags), shards, dst=0)
        if rank == 0:
            for i, (scene, fi) in enumerate(frags):
                out_dir = os.path.join(target_path, os.path.basename(scene), "seq-01")
                ensure_dir(out_dir)
                if i in meta:
                    _, out_file, xyz, xyz_down = meta[i]
                else:
                    from . import sparse as ME
                    xyz = read_ply_points(fi)
                    _, inds = ME.utils.sparse_quantize(np.floor(xyz / voxel_size), return_index=True)
                    xyz_down = xyz[inds]
                    out_file = os.path.join(out_dir, os.path.basename(fi).replace(".ply", ".npz"))
                save_descriptors(out_file, xyz, xyz_down, all_feats[i])
    return times, len(frags)


def main(argv=None):
    logging.basicConfig(format="%(asctime)s %(message)s", datefmt="%m/%d %H:%M:%S", level=logging.INFO,
                        stream=sys.stdout)
    p = argparse.ArgumentParser()
    p.add_argument("--source", required=True, type=str, help="the path of 3DMatch testing")
    p.add_argument("--target", required=True, type=str, help="the path of generating descriptor")
    p.add_argument("-m", "--model", default=None, type=str, help="the path of checkpoints.pth")
    p.add_argument("--voxel_size", default=0.05, type=float,
                   help="ignored, as in the reference: the checkpoint's config.voxel_size is used")
    p.add_argument("--extract_features", default=True, action="store_true")
    p.add_argument("--with_cuda", default=True, action="store_true")
    p.add_argument("--gather", action="store_true", help="multi-GPU: gather descriptors on rank 0 (RCCL)")
    p.add_argument("--workers", type=int, default=None,
                   help="loader / writer threads around the GPU (0 = the reference's sequential order; default: from "
                        "os.cpu_count() // world_size -- 4..16)")
    p.add_argument("--batch_points", type=int, default=None,
                   help="opt-in: consecutive fragments share a forward up to this many points (-1: ~1.1 M = four to five 3DMatch "
                        "fragments, +25 %% GPU throughput); descriptors then agree with the default's to 2e-6 instead of bit "
                        "for bit.  Default 0: one fragment per forward, files identical whatever the workers / ranks")
    p.add_argument("--npz_threads", type=int, default=None,
                   help="host threads of ONE descriptor file's block-parallel deflate (imf_npz_write_mt; default: from "
                        "os.cpu_count() // world_size // workers -- 1..16, or $IMFNET_NPZ_THREADS)")
    p.add_argument("--npz_level", type=int, default=None,
                   help="zlib level of the descriptor files: 0 = stored (np.savez), 1..9 deflate (np.savez_compressed is "
                        "6); default 1 or $IMFNET_NPZ_LEVEL.  np.load returns identical arrays either way")
    p.add_argument("--seeded_weights", type=int, default=None,
                   help="no checkpoint: random weights from this seed (plumbing / benchmarking)")
    args = p.parse_args(argv)

    from . import dataio
    if args.npz_level is not None:
        dataio.NPZ_LEVEL = args.npz_level
    rank, world, local = idist.init_from_env("nccl")
    # host threads: this rank's share of the box ("bound by host-side decode long before xGMI")
    share = max(1, (os.cpu_count() or 1) // max(1, world))
    if args.workers is None:
        args.workers = max(4, min(16, share // 16))
    if args.npz_threads is not None:
        dataio.NPZ_THREADS = max(1, args.npz_threads)
    elif dataio.NPZ_THREADS <= 0:
        dataio.NPZ_THREADS = max(1, min(16, share // (2 * max(1, args.workers))))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    ensure_dir(args.target)
    if args.model is not None:
        state_dict, config = load_checkpoint(args.model)
    else:
        assert args.seeded_weights is not None, "give --model or --seeded_weights"
        state_dict, config = None, Config()
    if state_dict is None:
        torch.manual_seed(args.seeded_weights)            # before construction: every rank must build the same network
    Model = load_model(config.model)
    model = Model(1, config.model_n_out, bn_momentum=0.05, normalize_feature=config.normalize_feature,
                  conv1_kernel_size=config.conv1_kernel_size, D=3, config=config)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    else:
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.1)
    model = model.eval().to(device)
    t_wall = time.time()
    with torch.no_grad():
        times, n = extract_features_batch(model, config, args.source, args.target, config.voxel_size, device,
                                          gather=args.gather, workers=args.workers, batch_points=args.batch_points)
    t_wall = time.time() - t_wall
    if times:
        r = model.fragment_runner() if hasattr(model, "fragment_runner") else None
        print(f"[rank {rank}] All Time:{np.sum(times)},AVG:{np.sum(times) / len(times)} "
              f"({len(times)} of {n} fragments); wall {t_wall:.2f} s = {len(times) / t_wall:.1f} fragments/s end to end"
              + (f"; capacity buckets {len(r.buckets)}, redone {r.stats['redone']}" if r is not None else ""))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
