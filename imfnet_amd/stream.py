"""Host arrays in -> descriptors on the host, as a stream: the Python face of the library's pipeline
(csrc/pipeline.hip, `imf_pipeline_*`).

The reference's loop body (scripts/generate_desc.py:99-123) is `extract_features(model, xyz, ..., image)` on host
arrays followed by `feature.detach().cpu().numpy()`.  Here that span is a JOB: the caller's thread stages the
fragment(s) into a pinned block (`FragmentStreamer.submit`: one C pass over the points, narrowing float64 values that
are float32 values -- what a PLY holds -- so half the bytes cross PCIe), the pipeline's worker thread issues the upload
kernel, the ~150 launches of `imf_fragment_forward` and the download kernel on three streams, and `StreamJob.wait()`
returns once xyz_down and the descriptors sit in the job's pinned output block.  Uploads and downloads of neighbouring
jobs run under the current forward's kernels; the interpreter is not on the issue path.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib
from ._lib import META_WORDS, ImfError, Job, check
from .model import graph


def _copy_engine_calls_return_at_once(device, stream):
    """`imfnet_amd.SDMA_ASYNC` only says that ROC_CPU_WAIT_FOR_SIGNAL=0 was exported before torch touched the
    runtime; if something else started HIP earlier the variable had no effect, hipMemcpyAsync behind queued kernels blocks
    its caller (~1 ms per call inside a forward) and the pipeline silently loses what section 4e of LAB_NOTES.md gained.
    So the EFFECTIVE mode is measured once per streamer: ~2 ms of GPU work is queued, then one small pinned device-to-host
    copy is issued behind it and the issue call is timed (~20 us when the dependency is handed to the GPU, the length of
    the queued work when the CPU waits).  Any failure of the probe keeps the declared mode."""
    import time
    try:
        with torch.cuda.device(device), torch.cuda.stream(stream):
            src = torch.zeros(1 << 16, dtype=torch.uint8, device=device)
            dst = torch.empty(1 << 16, dtype=torch.uint8).pin_memory()
            dst.copy_(src, non_blocking=True)              # (a process's first device-to-host copy costs ~13 ms once)
            stream.synchronize()
            torch.cuda._sleep(4_000_000)                   # ~2 ms of queued GPU time
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            dt = time.perf_counter() - t0
            stream.synchronize()
        return dt < 0.5e-3
    except Exception:                                      # noqa: BLE001 -- a probe must never take the pipeline down
        return True


class StreamJob:
    """One submitted forward: `wait()` -> FragmentResult (counts, flags, item spans); `views` are the numpy views of the
    pinned blocks (`sel` = xyz_down rows, `F` = descriptors) -- valid until the slot is handed to another submit."""

    def __init__(self, streamer, items, slot, bucket, views, ticket, narrowed):
        self.streamer, self.items, self.slot, self.bucket, self.views = streamer, items, slot, bucket, views
        self.ticket, self.narrowed = ticket, narrowed
        self.res = None
        self.ms = None               # (upload, forward incl. queueing behind the previous one, download) in ms
        self.host_ms = None
        self._lock = threading.Lock()

    def wait(self):
        return self.streamer._complete(self)


class FragmentStreamer:
    """Capacity buckets (device) + the library pipeline of one FragmentRunner on one device.  `n_buckets` forwards of one
    capacity key can be in flight (upload of k+1, forward of k, download of k-1); pinned HostSlots belong to the caller.
    The streamer's buckets are its OWN lanes 1 .. n_buckets of a key -- lane 0 is the bucket the direct capacity-mode
    launches use (FragmentRunner.launch / run), which follows the main stream's order only, whereas a pipeline job's upload
    and head run on the image / side streams (a shared bucket could be overwritten under an unsynchronised direct
    forward).  At most MAX_KEYS capacity keys stay resident: beyond that the least recently used key with nothing in flight is
    dropped -- its lanes leave this streamer, the runner and (weakly referenced) the pinned slots, and the device blocks go
    back to the allocator (a re-observe after a flagged fragment changes every key and would otherwise strand the old
    lanes: input block, output block, pyramid, arenas and image buffers per bucket, without bound over a varied data set)."""

    MAX_KEYS = 8

    def __init__(self, runner, device, n_buckets=3, sdma_copies=None, copy_blocks=0, head_on_side=True):
        self.runner, self.device, self.n_buckets = runner, device, max(1, int(n_buckets))
        # a job's table reset / level-0 pyramid / image fork on the side stream, under the previous job's last convolutions
        # (imf_fragment_io.head_on_side); the pipeline hands the upload's event in, buckets are reused only after wait()
        self.head_on_side = bool(head_on_side)
        self.L = runner.L
        main = runner.main_stream(device)
        if sdma_copies is None:                       # copy engines when hipMemcpyAsync cannot block the worker (see __init__.py)
            from . import SDMA_ASYNC
            sdma_copies = SDMA_ASYNC and _copy_engine_calls_return_at_once(device, main)
        self.sdma_copies = bool(sdma_copies)
        flags = (_lib.PIPELINE_SDMA_COPIES if sdma_copies else 0) | ((int(copy_blocks) & 0xFFF) << 8)
        with torch.cuda.device(device):
            self.handle = self.L.imf_pipeline_create(main.cuda_stream, 32, flags)
        if not self.handle:
            raise ImfError("imf_pipeline_create failed: " + self.L.imf_last_error().decode())
        self.main = main
        self._free = {}              # capacity key -> lanes not in flight
        self._made = {}              # capacity key -> SET of the live lane numbers (a dropped lane's number is free again:
                                     # ADVICE r5 -- a count would re-issue the number of a lane that still exists)
        self._used = {}              # capacity key -> submit counter at its last use (eviction is LRU over keys)
        self._tick = 0
        self._inflight = []          # StreamJobs not yet completed, in submit order
        self._lock = threading.RLock()
        self._scratch32 = None

    def close(self):
        if self.handle:
            for job in list(self._inflight):
                try:
                    self._complete(job)
                except ImfError:
                    pass
            self.L.imf_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:                             # noqa: BLE001 -- interpreter shutdown
            pass

    # -- buckets ------------------------------------------------------------------------------------------------------
    def _acquire(self, key):
        """A bucket of capacity `key` that no job is using; completes the oldest job of that key when all are busy."""
        while True:
            with self._lock:
                self._tick += 1
                self._used[key] = self._tick
                free = self._free.setdefault(key, [])
                if free:
                    b = free.pop()
                    b.in_flight = True
                    self.runner.touch(b)              # (the runner's eviction is LRU: a lane in use is not the oldest)
                    return b
                if len(self._made.get(key, ())) < self.n_buckets:
                    if key not in self._made:
                        self._evict_lru(keep=key)
                    b = self._new_lane(key)
                    b.in_flight = True
                    return b
                old = next((j for j in self._inflight if j.bucket.key == key), None)
            if old is None:
                raise ImfError("streamer: no bucket of this capacity is free and none is in flight")
            self._complete(old)                       # (outside the lock: it blocks on the GPU)

    def _new_lane(self, key):
        """A further bucket of `key` under the smallest lane number this streamer does not hold (lanes 1 ..: lane 0 belongs to
        the direct launches).  Called under the lock."""
        lanes = self._made.setdefault(key, set())
        lane = next(i for i in range(1, len(lanes) + 2) if i not in lanes)
        lanes.add(lane)
        b = self.runner.bucket(key, self.device, self.main, lane=lane)
        b.lane = lane
        return b

    def _release(self, b):
        b.in_flight = False
        lanes = self._free.get(b.key)
        if lanes is not None:                         # (None: the key was evicted while this lane was out)
            lanes.append(b)

    def _evict_lru(self, keep=None):
        """Drop least-recently-used keys with nothing in flight until fewer than MAX_KEYS remain (called under the lock)."""
        while len(self._made) >= self.MAX_KEYS:
            idle = [k for k in self._made if k != keep and len(self._free.get(k, ())) == len(self._made[k])]
            if not idle:
                return
            victim = min(idle, key=lambda k: self._used.get(k, 0))
            lanes = self._made.pop(victim)
            self._free.pop(victim, None)
            self._used.pop(victim, None)
            for lane in sorted(lanes):                 # every job of these lanes has been waited for: nothing is queued on them
                self.runner.drop_bucket((victim, lane))

    def forget(self, b):
        """The runner dropped bucket b (FragmentRunner.drop_bucket): it must not be handed out again."""
        with self._lock:
            lanes = self._free.get(b.key)
            if lanes and b in lanes:
                lanes.remove(b)
                live = self._made.get(b.key)
                if live is not None:
                    live.discard(getattr(b, "lane", None))
                    if not live and not any(j.bucket.key == b.key for j in self._inflight):
                        self._made.pop(b.key, None)
                        self._free.pop(b.key, None)
                        self._used.pop(b.key, None)

    def fill_lanes(self):
        """Create every lane still missing for the capacity keys seen so far (lanes are otherwise created on demand -- when
        all existing ones of a key are in flight -- which depends on timing; a measurement wants them all to exist)."""
        with self._lock:
            for key in list(self._made):
                while len(self._made[key]) < self.n_buckets:
                    self._free.setdefault(key, []).append(self._new_lane(key))

    # -- submit / wait ------------------------------------------------------------------------------------------------
    def submit(self, items, voxel_size, slot, more_follow=False, skip_descriptors=False):
        """Queue `items` = [(xyz [N,3] host float32/float64 array, image [1,3,H,W] host float32 array)] as ONE forward
        (several items: the model's batched call, model/resunet.py:241-250) with `slot` (graph.HostSlot) as its pinned
        staging.  more_follow: another submit is coming -- this job's download is then issued behind the next job's
        launches (the side stream's idle half) instead of in front of them; `wait()` ends the deferral.  Returns a
        StreamJob, or None when the runner has no capacities for it yet (the caller runs the exact path, which teaches the
        runner)."""
        runner = self.runner
        k = len(items)
        n_each = [int(x.shape[0]) for x, _ in items]
        n = sum(n_each)
        x0, i0 = items[0]
        H, W = int(i0.shape[2]), int(i0.shape[3])
        narrowed = False
        v = b = None
        if x0.dtype == np.float64:
            # float64 values that are float32 values (a PLY's points widened by the reader): stage and upload as float32
            key = runner.caps_for(n, k, H, W, voxel_size, False)
            if key is None:
                return None
            b = self._acquire(key)
            v = slot.bind(b)
            narrowed, at = True, 0
            for j, (xyz, _) in enumerate(items):
                src = np.ascontiguousarray(xyz)
                dst = v["xyz"][at:at + n_each[j]]
                rc = self.L.imf_host_narrow_points(src.ctypes.data, src.size, dst.ctypes.data)
                if rc != 1:
                    narrowed = False
                    break
                at += n_each[j]
            if not narrowed:
                with self._lock:
                    self._release(b)
                b = None
        if b is None:
            key = runner.caps_for(n, k, H, W, voxel_size, x0.dtype == np.float64)
            if key is None:
                return None
            b = self._acquire(key)
            v = slot.bind(b)
            at = 0
            for j, (xyz, _) in enumerate(items):
                np.copyto(v["xyz"][at:at + n_each[j]], xyz)
                at += n_each[j]
        vals, at = [n, k], 0
        for j, (_, img) in enumerate(items):
            np.copyto(v["image"][j:j + 1], img)
            vals.append(at)
            at += n_each[j]
        v["dyn"][:len(vals)] = vals
        b.dyn_values = vals
        job = Job()
        job.net, job.img = C.pointer(runner.net_desc), C.pointer(runner.img_plan.desc)
        job.caps, job.io = C.pointer(b.caps), C.pointer(b.io)
        job.host_in, job.dev_in = slot.inbuf.data_ptr(), b.inbuf.data_ptr()
        job.in_bytes = b.lay["xyz"] + n * 3 * b.xyz.element_size()
        # skip_descriptors (copy-engine mode): the download ends in front of the descriptors -- meta + xyz_down only; the caller
        # takes the descriptors from the bucket on the device (extract_features(host_descriptors=False))
        out_bytes = b.lay["F"] if (skip_descriptors and self.sdma_copies) else b.outbuf.numel()
        job.dev_out, job.host_out, job.out_bytes = b.outbuf.data_ptr(), slot.outbuf.data_ptr(), out_bytes
        job.sel, job.sel_offset = b.sel.data_ptr(), b.lay["sel"]
        job.out_offset, job.out_row_bytes = b.lay["F"], int(b.out.shape[1]) * 4
        job.defer_download = 1 if more_follow else 0
        b.io.trace = None
        b.io.head_on_side = 1 if self.head_on_side else 0
        ticket = self.L.imf_pipeline_submit(self.handle, C.byref(job))
        if ticket < 0:
            with self._lock:
                self._release(b)
            check(ticket, "imf_pipeline_submit")
        b.launches += 1
        runner.stats["eager"] += 1
        sj = StreamJob(self, items, slot, b, v, ticket, narrowed)
        with self._lock:
            self._inflight.append(sj)
        return sj

    def _complete(self, sj):
        with sj._lock:                                # one waiter per job; the streamer's lock only guards the bookkeeping
            if sj.res is not None:
                return sj.res
            ms = (C.c_float * 24)()
            rc = self.L.imf_pipeline_wait(self.handle, sj.ticket, ms)
            with self._lock:
                self._inflight.remove(sj)
                self._release(sj.bucket)
            check(rc, "imf_pipeline_wait")
            sj.ms = (float(ms[0]), float(ms[1]), float(ms[2]))
            sj.stamps = tuple(float(ms[i]) for i in range(8, 21))   # device x5, host x3 (ms since the pipeline's creation)
            sj.host_ms = tuple(float(ms[i]) for i in range(3, 8))   # queue wait, issue, to download issue, to completion, in wait()
            res = graph.FragmentResult(sj.bucket, sum(int(x.shape[0]) for x, _ in sj.items), len(sj.items), None, None,
                                       pooled=False)
            res._meta = sj.views["meta"].numpy().copy()
            sj.res = res
            return res
