"""`extract_features` -- the harness function the reference times (util/misc.py:21-104,
called from scripts/generate_desc.py:100).

Same signature and return convention:  (xyz_down float64 [M,3] on the host, F float32 [M,32] on the
device), row i of F describing row i of xyz_down.  What differs is where the work happens: the
reference voxelises on the host (np.floor + ME.utils.sparse_quantize) and uploads coordinates; here
the raw points are uploaded once and quantisation, first-occurrence unique, the pyramid and the
rulebooks are all built on the GPU (imf_voxelize & co.), bit-identical to the host result.
"""
import numpy as np
import torch

from . import ops
from . import sparse as ME
from .model import graph
from ._lib import MAX_BATCH, FLAG_RANGE, ImfError, check


def _cuda_device(device):
    """torch.device with an explicit index (`cuda` -> `cuda:<current>`): per-device state is keyed by it."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise ImfError('imfnet_amd runs on the GPU only; there is no CPU fallback')
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return device


def _as_device_points(xyz, device):
    if torch.is_tensor(xyz):
        t = xyz
    else:
        a = np.asarray(xyz)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    return t.to(device, non_blocking=True).contiguous()


def start_geometry(xyz, voxel_size, device, inputs_ready=False, item_starts=None):
    """Queue the geometry build of a fragment (upload included when xyz is a host array) and return
    at once.  Handing the result to `sparse_tensor_from_points(geometry=...)` later lets the harness
    build fragment i+1's voxel pyramid while fragment i's decoder is still running.
    A BATCH of fragments (the reference's batched SparseTensor, model/resunet.py:241-250): pass a list of
    point arrays, or one array holding the items back to back plus `item_starts` (first point of each)."""
    if isinstance(xyz, (list, tuple)):
        dev = torch.device(device)
        on_host = not all(torch.is_tensor(a) and a.is_cuda for a in xyz)
        stream = ops.geometry_stream(dev) if on_host else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream):
            parts = [_as_device_points(a, device) for a in xyz]
            if len({t.dtype for t in parts}) > 1:
                parts = [t.double() for t in parts]
            pts = torch.cat(parts, 0)
        item_starts = [0]
        for t in parts[:-1]:
            item_starts.append(item_starts[-1] + t.shape[0])
        if not on_host and not inputs_ready:
            pass                                      # the concatenation ran on the current stream: ordered
        return ops.PyramidFuture(pts, voxel_size, 4, 0, inputs_ready=on_host, item_starts=item_starts)
    on_host = not (torch.is_tensor(xyz) and xyz.is_cuda)
    if on_host:                                       # upload on the geometry stream itself: in order
        with torch.cuda.stream(ops.geometry_stream(torch.device(device))):
            pts = _as_device_points(xyz, device)
        inputs_ready = True
    else:
        pts = _as_device_points(xyz, device)
    return ops.PyramidFuture(pts, voxel_size, 4, 0, inputs_ready=inputs_ready, item_starts=item_starts)


def sparse_tensor_from_points(xyz, voxel_size, device, feats=None, before_sync=None, inputs_ready=False,
                              geometry=None):
    """Voxelise raw points on the GPU.  Returns (SparseTensor with all-ones / gathered features,
    inds int32 CUDA tensor of each voxel's first point).
    The whole geometry (voxel hash + 3 coarser levels) is one library call on a dedicated
    high-priority stream; `before_sync` (callable) runs after it is queued and before the one host
    wait, so independent work (the image branch) is put on the GPU while the host waits for the row
    counts.  `inputs_ready=True` promises that a device-resident `xyz` is already complete (no
    pending producer on the current stream), which lets the geometry of fragment i+1 overlap the
    convolutions of fragment i."""
    fut = geometry if geometry is not None else start_geometry(xyz, voxel_size, device, inputs_ready)
    if before_sync is not None:
        before_sync()
    levels = fut.result()
    pts = fut.xyz
    lv = levels[0]
    cm = ME.CoordinateManager.from_levels(levels)
    inds = lv.first_idx
    if feats is None:
        f = torch.ones((lv.n, 1), dtype=torch.float32, device=pts.device)        # util/misc.py:76-79
    else:
        f = torch.as_tensor(feats, dtype=torch.float32).to(pts.device)[inds.long()]   # :89
    st = ME.SparseTensor(f, coordinate_map_key=ME.CoordinateMapKey(1), coordinate_manager=cm)
    st._all_ones = feats is None          # lets the first conv skip the feature gather
    return st, inds


def _host_inputs(xyz, image):
    """(points as a host float32 / float64 array, image as a host float32 array), or None when either lives on the GPU."""
    if (torch.is_tensor(xyz) and xyz.is_cuda) or (torch.is_tensor(image) and image.is_cuda):
        return None
    a = xyz.detach().numpy() if torch.is_tensor(xyz) else np.asarray(xyz)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    img = image.detach().numpy() if torch.is_tensor(image) else np.asarray(image)
    return np.ascontiguousarray(a), np.ascontiguousarray(img, dtype=np.float32)


def _extract_with_runner(runner, xyz, voxel_size, device, image, host_descriptors=True):
    """extract_features through the capacity-mode graph; None = not applicable / flagged (caller runs the exact path)."""
    n = int(xyz.shape[0])
    is_f64 = (xyz.dtype == torch.float64) if torch.is_tensor(xyz) else (np.asarray(xyz).dtype != np.float32)
    img = image if torch.is_tensor(image) else torch.as_tensor(np.asarray(image), dtype=torch.float32)
    if img.dim() != 4 or img.shape[0] != 1:
        return None
    stream, outer = runner._stream_for(device, None)
    host = _host_inputs(xyz, image)
    if host is not None:
        # host arrays (the reference's call, scripts/generate_desc.py:100): one job of the streaming pipeline (stream.py) --
        # pinned staging both ways, float32 upload when the float64 values are float32 values, xyz_down = xyz[inds]
        # gathered on the device; the download is the bucket's output block (copy-engine mode: the whole capacity-sized block,
        # or only its meta + xyz_down part with host_descriptors=False; copy-kernel mode: only the rows that exist)
        slots = runner.host_slots
        slot = slots[0] if slots else graph.HostSlot()
        if not slots:
            slots.append(slot)
        job = runner.streamer(device).submit([host], voxel_size, slot, skip_descriptors=not host_descriptors)
        if job is None:
            return None
        res = job.wait()
        if res.flags:                                 # does not fit this bucket: exact path (which re-observes)
            runner.stats["redone"] += 1
            if outer is not None:
                outer.wait_stream(stream)
            return None
        m = res.counts[0]
        v = job.views
        with torch.cuda.stream(stream):
            F = res.bucket.out[:m].clone()            # the caller owns its descriptors (the bucket is reused)
        sel = v["sel"][:m].copy()
        if outer is not None:
            outer.wait_stream(stream)
            F.record_stream(outer)
        if host_descriptors:
            F.host = v["F"][:m]                       # the same descriptors, already on the host (pinned; valid until the
        return sel, F                                 # next extract_features call): saves the caller's F.cpu()
    key = runner.caps_for(n, 1, int(img.shape[2]), int(img.shape[3]), voxel_size, is_f64)
    if key is None:
        return None
    b = runner.bucket(key, device, stream)
    # a host point array next to a device image (util/misc.py:97 takes the mix through torch.as_tensor)
    src = xyz if torch.is_tensor(xyz) else torch.from_numpy(np.ascontiguousarray(np.asarray(xyz)))
    if src.dtype not in (torch.float32, torch.float64):
        src = src.double()
    runner.stage(b, src, [0], img, stream)
    res = runner.launch(b, n, 1, stream)
    if res.flags:                                     # does not fit this bucket: exact path (which re-observes)
        runner.stats["redone"] += 1
        if outer is not None:
            outer.wait_stream(stream)
        return None
    with torch.cuda.stream(stream):
        F = res.F.clone()                             # the caller owns its descriptors (the bucket is reused)
        inds = res.first_idx.long()
        if torch.is_tensor(xyz) and xyz.is_cuda:
            sel = xyz.detach()[inds].cpu().numpy().astype(np.float64)
        else:
            inds_host = inds.cpu().numpy()
            host_xyz = xyz.detach().numpy() if torch.is_tensor(xyz) else np.asarray(xyz)
            sel = host_xyz[inds_host].astype(np.float64, copy=False)
    if outer is not None:
        outer.wait_stream(stream)
        F.record_stream(outer)                        # allocated on the runner's stream, used on the caller's
    return sel, F


POINT_BUDGET = 1_100_000          # batch="auto": points per forward (four S50k fragments)


FragmentStreamerLanes = 3          # FragmentStreamer's default n_buckets (stream.py): forwards of one capacity key in flight


def extract_features_stream(model, fragments, voxel_size, device=None, depth=3, copy=True, batch=2, point_budget=None,
                            device_sink=None):
    """`extract_features` over a STREAM of host fragments (SURVEY 8d's span -- host arrays in, descriptors back on the
    host -- pipelined): yields (xyz_down float64 [M,3], F float32 [M,32] numpy) per fragment, in order.  `fragments`:
    iterable of (xyz [N,3] host array, image [1,3,H,W] host array).  Every forward is a job of the library's pipeline
    (stream.py / csrc/pipeline.hip): this thread only stages the next fragments into a pinned block while the pipeline's
    worker issues launches; the upload of job k+1 and the download of job k-1 are copy kernels on their own streams under
    job k's kernels.  Up to `depth` jobs are in flight.
    batch: consecutive fragments per forward (the model's batched call, model/resunet.py:241-250: rows grouped by
    fragment, one image each) -- the stride-4/8 levels of ONE fragment leave half the chip idle, two fill it: 0.47 vs 0.68 ms
    of GPU time per S50k fragment, four 0.41, eight 0.37 (bench.py: batch_4 / batch_8); fragments of a batch share point
    dtype and image size (others go alone).  batch="auto": consecutive fragments are grouped until their points reach
    `point_budget` (default POINT_BUDGET, ~ four S50k fragments) or IMF_MAX_BATCH fragments -- more per forward for small
    fragments, fewer for large ones; a fragment's results then wait for its whole group (latency for throughput).
    copy=True: the arrays of a yield are fresh host copies; copy=False: views of the pinned slot, valid until the NEXT
    item is requested (a 6.5 MB copy into newly faulted pages costs ~0.3 ms per fragment).  Fragments the capacity mode
    cannot take (no capacities yet, a flag) go through `extract_features`, in order.
    device_sink: optional callable(F_dev) called once per fragment, in order, BEFORE its yield, with the fragment's
    descriptors as a DEVICE tensor [M, 32] (a view of the capacity bucket's output block, or the exact path's tensor) while
    the runner's main stream is torch's current stream: a copy the sink enqueues there (e.g. into a per-rank send buffer for
    the RCCL gather, dist.gather_fragment_descriptors(packed=...)) is ordered before any later forward that reuses the
    bucket -- the descriptors go from the bucket to the collective without a host round trip."""
    from collections import deque
    device = _cuda_device(device or 'cuda:0')
    if model.training:
        model.eval()
    runner = model.fragment_runner() if hasattr(model, "fragment_runner") else None
    depth = max(1, int(depth))
    if device_sink is not None:
        # the sink reads a job's rows from its capacity BUCKET after job.wait(): with more jobs of one key in flight than the
        # streamer has lanes, the streamer completes the oldest job itself and hands its bucket to the new submit -- whose
        # forward could overwrite the rows before finish() reaches the sink.  The pinned host copy is not affected.
        lanes = runner.streamer(device).n_buckets if runner is not None else FragmentStreamerLanes
        depth = min(depth, lanes)
    n_slots = depth + 2                               # in flight + the one the consumer holds (copy=False) + one being staged
    auto = isinstance(batch, str)
    if auto and batch != "auto":
        raise ValueError("batch: an integer or 'auto'")
    budget = int(point_budget or POINT_BUDGET)
    batch = MAX_BATCH if auto else max(1, min(int(batch), MAX_BATCH))

    def state(runner):
        """(streamer, pinned slots) of this runner and device; pinned blocks are expensive to create: kept with the runner."""
        if runner is None:
            return None, []
        cache = runner.stream_state
        if cache is None or cache[0] != device or len(cache[1]) < n_slots:
            cache = runner.stream_state = (device, [graph.HostSlot() for _ in range(n_slots)])
        return runner.streamer(device), list(cache[1][:n_slots])

    streamer, free = state(runner)
    caller = torch.cuda.current_stream(device)
    if runner is not None:
        runner.main_stream(device).wait_stream(caller)
    inflight = deque()                                # (items, StreamJob or None)
    lent = []                                         # the slot whose views the consumer currently holds (copy=False)

    def exact(items):
        for xyz, image in items:
            with torch.no_grad():
                xd, F = extract_features(model, xyz, voxel_size=voxel_size, device=device, skip_check=True, image=image)
            if device_sink is not None:
                device_sink(F)
            Fh = getattr(F, "host", None)
            yield xd, (Fh.copy() if Fh is not None else F.cpu().numpy())

    def finish(entry):
        """The results of one forward (generator: one (xyz_down, F) per fragment of the entry, in order)."""
        items, job = entry
        while lent:
            free.append(lent.pop())
        res = job.wait()
        if not res.flags:
            st = runner.stats
            st["stream_gpu_ms"] = st.get("stream_gpu_ms", 0.0) + sum(job.ms)
            st["stream_n"] = st.get("stream_n", 0) + len(items)
            st["stream_jobs"] = st.get("stream_jobs", 0) + 1
            if "stream_trace" in st:
                st["stream_trace"].append(job.stamps)
            for i, k in enumerate(("queue_ms", "issue_ms", "to_download_ms", "to_done_ms", "wait_ms")):
                st["stream_" + k] = st.get("stream_" + k, 0.0) + job.host_ms[i]
            v = job.views
            spans = res.items() if len(items) > 1 else [(0, res.counts[0])]
            if device_sink is not None:               # the bucket's device rows, on the stream its forwards run on
                with torch.cuda.stream(runner.main_stream(device)):
                    for r0, m in spans:
                        device_sink(job.bucket.out[r0:r0 + m])
            if copy:
                outs = [(v["sel"][r0:r0 + m].copy(), v["F"][r0:r0 + m].copy()) for r0, m in spans]
                free.append(job.slot)
            else:
                outs = [(v["sel"][r0:r0 + m], v["F"][r0:r0 + m]) for r0, m in spans]
                lent.append(job.slot)
            yield from outs
            return
        runner.stats["redone"] += len(items)
        runner.stats["redone_flags"] = runner.stats.get("redone_flags", 0) | res.flags   # which flags asked for the redo
        if len(items) > 1 and (res.flags & 4):        # the batch's bounding box outgrew the grid: size the next one by it
            runner.observe_batch(len(items), res.bbox)
        free.append(job.slot)
        yield from exact(items)

    with torch.no_grad():
        group = []

        def flush():
            nonlocal runner, streamer, free
            if not group:
                return
            g = list(group)
            del group[:]
            while inflight and (len(inflight) >= depth or not free):
                yield from finish(inflight.popleft())
            job = None
            if streamer is not None and runner.ratios is not None:
                slot = free.pop()
                job = streamer.submit(g, voxel_size, slot, more_follow=True)
                if job is None:
                    free.append(slot)
            if job is not None:
                inflight.append((g, job))
                if slot.grown:                        # a new size: grow every FREE slot now, not one by one beside the worker
                    for other in free:
                        other.reserve_like(slot, device)
                return
            while inflight:                           # teach the runner on the exact path first, in order
                yield from finish(inflight.popleft())
            yield from exact(g)
            new = model.fragment_runner() if hasattr(model, "fragment_runner") else None
            if new is not runner:                     # (a refresh rebuilt the plans: the new runner has its own streamer / slots)
                while lent:
                    free.append(lent.pop())
                runner = new
                streamer, free = state(runner)

        for xyz, image in fragments:
            host = _host_inputs(xyz, image)
            if host is None:
                raise ImfError("extract_features_stream takes host arrays")
            if group and (host[0].dtype != group[0][0].dtype or host[1].shape != group[0][1].shape):
                yield from flush()
            group.append(host)
            if len(group) >= batch or (auto and sum(len(x) for x, _ in group) >= budget):
                yield from flush()
        yield from flush()
        while inflight:
            yield from finish(inflight.popleft())
    if runner is not None:
        caller.wait_stream(runner.main_stream(device))


def extract_features(model, xyz, rgb=None, normal=None, voxel_size=0.05, device=None,
                     skip_check=False, is_eval=True, image=None, host_descriptors=True):
    """xyz: [N,3] points (numpy float64/float32, or a tensor already on the device).
    rgb in [0,1] / normal in [-1,1] are optional per-point inputs (concatenated as rgb-0.5, normal/2);
    with neither, the input feature is a column of ones.  image: [1,3,H,W] float32.
    Returns (xyz_down float64 [M,3] on the host, F float32 [M,32] ON THE DEVICE) -- util/misc.py:100-104.  With host arrays
    in, the descriptors are ALSO brought back in the same download as xyz_down (`F.host`, a pinned view valid until the next
    call: it spares the caller's F.cpu(), scripts/generate_desc.py:122); host_descriptors=False leaves them on the device
    only -- exactly the reference's return, 6.5 MB less over PCIe per S50k fragment."""
    if is_eval and model.training:                   # (walking ~190 modules per fragment costs 0.7 ms)
        model.eval()
    if not skip_check:
        assert xyz.shape[1] == 3
        N = xyz.shape[0]
        if rgb is not None:
            assert N == len(rgb) and rgb.shape[1] == 3
            if np.any(rgb > 1):
                raise ValueError('Invalid color. Color must range from [0, 1]')
        if normal is not None:
            assert N == len(normal) and normal.shape[1] == 3
            if np.any(normal > 1):
                raise ValueError('Invalid normal. Normal must range from [-1, 1]')
    if device is None:
        device = torch.device('cuda:0')
    device = _cuda_device(device)

    feats = []
    if rgb is not None:
        feats.append(np.asarray(rgb) - 0.5)
    if normal is not None:
        feats.append(np.asarray(normal) / 2)
    feats = np.hstack(feats) if feats else None

    # Whole-fragment graph (model/graph.py): no count readback, one launch.  Used once the runner has seen a
    # fragment (it needs voxel-per-point ratios to size its capacity buckets); anything it flags is redone here.
    runner = model.fragment_runner() if (feats is None and image is not None and hasattr(model, "fragment_runner")) else None
    if runner is not None:
        got = _extract_with_runner(runner, xyz, voxel_size, device, image, host_descriptors)
        if got is not None:
            return got
    # descriptor extraction is inference: with is_eval (the reference's callers all sit under torch.no_grad())
    # no autograd graph is recorded, so the packed-plan forward runs instead of the per-layer training path
    with torch.set_grad_enabled(torch.is_grad_enabled() and not is_eval):
        out = _extract_exact(model, runner, xyz, feats, voxel_size, device, image)
        if hasattr(model, "take_flags") and model.take_flags(device) & FLAG_RANGE:
            # an activation left the f16 range of the split-f16 convolutions (it would have become inf): the
            # fragment is redone on the true-fp32 matrix instructions -- slower, never silently wrong
            import warnings
            warnings.warn("imfnet_amd: activation outside the f16 range; fragment recomputed with fp32 MFMA (variant 0)")
            out = model.forward_fp32(lambda: _extract_exact(model, None, xyz, feats, voxel_size, device, image))
    return out


def _extract_exact(model, runner, xyz, feats, voxel_size, device, image):
    """The exact path: geometry with one row-count readback, then the model's forward."""
    start = getattr(model, "start_image_branch", None)
    box = {}
    if start is not None:
        hook = lambda: box.setdefault("image", start(image, device=device))   # noqa: E731
    else:
        hook = None
    stensor, inds = sparse_tensor_from_points(xyz, voxel_size, device, feats, before_sync=hook)
    image_dev = box.get("image")
    if image_dev is None:
        image_dev = torch.as_tensor(image, dtype=torch.float32, device=device)
    F = model(stensor, image_dev).F
    if runner is not None:                            # teach the runner this fragment's voxel-per-point ratios
        cm = stensor.coordinate_manager
        lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
        if getattr(lv[0], "bbox", None) is not None:
            runner.observe(int(xyz.shape[0]), [l.n for l in lv], lv[0].bbox)

    if torch.is_tensor(xyz) and xyz.is_cuda:          # gather on the device, copy only the M selected rows
        return_coords = xyz.detach()[inds.long()].cpu().numpy().astype(np.float64)
    else:
        inds_host = inds.cpu().numpy().astype(np.int64)
        host = xyz.detach().numpy() if torch.is_tensor(xyz) else np.asarray(xyz)
        return_coords = host[inds_host].astype(np.float64, copy=False)
    return return_coords, F


def extract_features_batch(model, xyz_list, voxel_size, device, images):
    """Several fragments in ONE forward (rows grouped by fragment, one image each -- the batched call the
    reference's model accepts, model/resunet.py:241-250).  xyz_list: point arrays; images: [B,3,H,W].
    Returns [(xyz_down float64 [M_b,3], F_b device view [M_b,32])] in input order.  Small fragments share
    the per-forward fixed costs (about 0.5 ms of launch-latency-bound coarse layers) this way."""
    device = _cuda_device(device)
    if model.training:
        model.eval()

    def run():
        fut = start_geometry(list(xyz_list), voxel_size, device)
        start = getattr(model, "start_image_branch", None)
        img = start(images, device=device) if start is not None else None
        if img is None:
            img = torch.as_tensor(images, dtype=torch.float32, device=device)
        stensor, inds = sparse_tensor_from_points(None, voxel_size, device, geometry=fut)
        with torch.no_grad():
            F = model(stensor, img).F
        return fut, stensor, inds, F

    fut, stensor, inds, F = run()
    if hasattr(model, "take_flags") and model.take_flags(device) & FLAG_RANGE:   # as in extract_features: never silently wrong
        import warnings
        warnings.warn("imfnet_amd: activation outside the f16 range; batch recomputed with fp32 MFMA (variant 0)")
        fut, stensor, inds, F = model.forward_fp32(run)
    sel = fut.xyz[inds.long()].cpu().numpy().astype(np.float64)
    items = stensor.coordinate_manager.level(1).items
    return [(sel[r0:r0 + rn], F[r0:r0 + rn]) for r0, rn in items]
