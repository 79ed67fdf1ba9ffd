"""Whole-fragment execution without the host in the loop: `imf_fragment_forward` (csrc/executor.hip) in
capacity mode, captured once per capacity bucket as a hipGraph and replayed per fragment.

The exact path (extract.py -> ResUNet2.forward -> NativePlan) reads the four level counts back after the
pyramid build and spends ~0.5 ms of host time enqueueing ~150 launches per fragment.  Here every buffer,
rulebook and grid is sized for a CAPACITY; the kernels read the actual counts from the pyramid's device meta
block and take the row-count-dependent decisions (split-K partitions, fusion hidden split) on the device with the
host's rule, so the descriptors are bit-identical to the exact path.  Per fragment the host writes 16 ints (point
count, item starts), the inputs land in the bucket's static buffers, and ONE hipGraphLaunch runs
util/misc.py:82-104 + model/resunet.py:163-235 end to end.  The counts come back with the descriptors.

Capacities are chosen from the point count and the voxel-per-point ratios of fragments seen so far (with a margin,
on a geometric grid so that fragments of similar size share a bucket).  A fragment that does not fit (flag word in
meta[1]) is redone on the exact path, which also updates the ratios.
"""
import ctypes as C
import weakref
import os

import numpy as np
import torch

from .. import _lib
from .._lib import DYN_WORDS, META_WORDS, FragmentCaps, FragmentIO, ImfError, MAX_BATCH, NetTrace, check

FLAG_NAMES = {1: "coordinate out of range", 2: "a level exceeded its row capacity", 4: "bounding box exceeds the bit grid",
              8: "an item has no voxel", 16: "far fewer rows than the capacity (split cover)"}


def _grid_up(v, ratio, lo):
    """Smallest lo * ratio^k >= v on a geometric grid (k integer), rounded up to a multiple of lo's granule."""
    c = float(lo)
    while c < v:
        c *= ratio
    return int(-(-int(c + 0.5) // 64) * 64)


class FragmentResult:
    """Device-side result of one launch; `sync()` reads the counts back (one small pinned D2H + event wait).
    pooled=False: the meta words arrive inside a caller-owned pinned buffer (HostSlot) with the caller's event."""

    def __init__(self, bucket, n_points, n_items, meta_host, done, pooled=True):
        self.bucket, self.n_points, self.n_items = bucket, n_points, n_items
        self._host, self._done, self._meta, self._pooled = meta_host, done, None, pooled

    def sync(self):
        if self._meta is None:
            self._done.synchronize()
            self._meta = self._host.numpy().copy()
            if self._pooled:
                self.bucket.pool.append((self._host, self._done))
            self._host = self._done = None
        return self._meta

    def __del__(self):
        try:
            if self._host is not None and self._pooled:   # never read: hand the pair back (reused once its event completed)
                self.bucket.pool.append((self._host, self._done))
        except Exception:                             # noqa: BLE001 -- interpreter shutdown
            pass

    @property
    def flags(self):
        m = self.sync()
        # (a variant-0 bucket computes in fp32 throughout: a value beyond the f16 range is a value there, not an error)
        return (int(m[1]) | int(m[3]) | int(m[5]) | int(m[7])) & ~getattr(self.bucket, "ignore_flags", 0)

    @property
    def counts(self):
        m = self.sync()
        return [int(m[2 * l]) for l in range(4)]

    def items(self, level=0):
        m, n = self.sync(), self.counts[level]
        st = [int(v) for v in m[16 + MAX_BATCH * level:16 + MAX_BATCH * level + self.n_items]]
        return [(st[b], (st[b + 1] if b + 1 < self.n_items else n) - st[b]) for b in range(self.n_items)]

    @property
    def bbox(self):
        return [int(v) for v in self.sync()[8:16]]

    @property
    def F(self):
        """[M,32] descriptors: a VIEW into the bucket's static output (valid until the bucket is launched again)."""
        return self.bucket.out[: self.counts[0]]

    @property
    def first_idx(self):
        return self.bucket.first_idx_view()[: self.counts[0]]


def _pad256(n):
    return (int(n) + 255) // 256 * 256


class HostSlot:
    """Pinned host staging of one in-flight fragment, laid out like its capacity bucket's two device blocks:
        in : [dyn scalars | image | points]                      -> ONE host-to-device copy per fragment
        out: [meta (counts, flags) | xyz_down | descriptors]     <- ONE device-to-host copy per fragment
    (every hipMemcpyAsync costs the host ~0.1 ms on this stack, whatever its size: tools/e2e_probe.py).  Host arrays are
    copied in with a plain single-threaded memcpy (np.copyto: 0.13 ms for 6 MB) -- torch's CPU copy_ fans a copy of that
    size out over its whole thread pool, and waking 128 idle OpenMP threads was measured at 10-30 ms per call."""

    def __init__(self, timing=False):
        self.inbuf = self.outbuf = None
        self.begin, self.done = torch.cuda.Event(enable_timing=timing), torch.cuda.Event(enable_timing=timing)
        self._views = None
        self.grown = False

    def reserve(self, in_bytes, out_bytes, device):
        """Grow the pinned blocks to at least these sizes.  hipHostMalloc holds the runtime's lock for ~7 ms per 16 MB and
        every other thread's HIP call waits behind it (measured: the pipeline's worker stalled 7-70 ms mid-stream when a
        slot grew beside it) -- callers grow ALL their slots at the first job of a new size (`reserve_like`)."""
        grown = False
        if self.inbuf is None or self.inbuf.numel() < in_bytes:
            self.inbuf, grown = torch.empty(int(in_bytes), dtype=torch.uint8, pin_memory=True), True
        if self.outbuf is None or self.outbuf.numel() < out_bytes:
            self.outbuf, grown = torch.empty(int(out_bytes), dtype=torch.uint8, pin_memory=True), True
        if grown:
            self._views = None
        return grown

    def reserve_like(self, other, device):
        return self.reserve(other.inbuf.numel(), other.outbuf.numel(), device)

    def bind(self, b):
        """Views of the pinned blocks in bucket b's layout (blocks grow on demand; views cached per bucket)."""
        grown = self.reserve(b.inbuf.numel(), b.outbuf.numel(), b.inbuf.device)
        self.grown = grown
        if grown or self._views is None:
            self._views = {}
            self._np = (self.inbuf.numpy(), self.outbuf.numpy())
        v = self._views.get(id(b))
        if v is not None and v["bucket"]() is b:
            return v
        for k in [k for k, w in self._views.items() if w["bucket"]() is None]:     # views of evicted buckets
            del self._views[k]
        L, (hi, ho) = b.lay, self._np
        # (a WEAK reference: a slot must not keep an evicted bucket's device blocks alive)
        v = dict(bucket=weakref.ref(b), dyn=hi[:4 * DYN_WORDS].view(np.int32),
                 image=hi[L["img"]:L["img"] + b.image.numel() * 4].view(np.float32).reshape(tuple(b.image.shape)),
                 xyz=hi[L["xyz"]:L["xyz"] + b.xyz.numel() * b.xyz.element_size()]
                     .view(np.float64 if b.xyz.dtype == torch.float64 else np.float32).reshape(-1, 3),
                 meta=self.outbuf[:4 * META_WORDS].view(torch.int32),
                 sel=ho[L["sel"]:L["sel"] + b.sel.numel() * 8].view(np.float64).reshape(-1, 3),
                 F=ho[L["F"]:L["F"] + b.out.numel() * 4].view(np.float32).reshape(tuple(b.out.shape)))
        self._views[id(b)] = v
        return v


class _Bucket:
    def __init__(self, runner, caps_tuple, dev):
        L = self.L = runner.L
        n_points, rows, n_items, H, W, grid_words, voxel, is_f64 = caps_tuple
        self.key, self.dev = caps_tuple, dev
        self.ignore_flags = _lib.FLAG_RANGE if runner.variant in (0, 3) else 0   # fp32 / bf16x3 operands: no f16 range
        net, img = runner.net_desc, runner.img_plan
        c = self.caps = FragmentCaps()
        c.n_points, c.n_items, c.img_h, c.img_w, c.bitgrid_words = n_points, n_items, H, W, grid_words
        for i in range(4):
            c.rows[i] = rows[i]
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)      # noqa: E731
        # inputs and outputs are views of two contiguous device blocks (see HostSlot): in = dyn | image | points,
        # out = meta | xyz_down | descriptors
        isz = 8 if is_f64 else 4
        lay = self.lay = dict(img=_pad256(4 * DYN_WORDS))
        lay["xyz"] = lay["img"] + _pad256(n_items * 3 * H * W * 4)
        lay["sel"] = _pad256(4 * META_WORDS)
        lay["F"] = lay["sel"] + _pad256(rows[0] * 24)
        self.inbuf = torch.zeros(lay["xyz"] + _pad256(n_points * 3 * isz), dtype=torch.uint8, device=dev)
        self.outbuf = torch.zeros(lay["F"] + _pad256(rows[0] * net.out_channels * 4), dtype=torch.uint8, device=dev)
        self.xyz = self.inbuf[lay["xyz"]:lay["xyz"] + n_points * 3 * isz].view(torch.float64 if is_f64 else torch.float32).view(n_points, 3)
        self.image = self.inbuf[lay["img"]:lay["img"] + n_items * 3 * H * W * 4].view(torch.float32).view(n_items, 3, H, W)
        self.dyn = self.inbuf[:4 * DYN_WORDS].view(torch.int32)
        self.dyn_values = None                        # what the device copy currently holds
        self.meta = self.outbuf[:4 * META_WORDS].view(torch.int32)
        self.sel = self.outbuf[lay["sel"]:lay["sel"] + rows[0] * 24].view(torch.float64).view(rows[0], 3)   # xyz[inds]
        self.pool = []                                # (pinned meta copy, event) pairs of finished results
        self.pyr = u8(L.imf_fragment_pyramid_bytes(C.byref(c)))
        ib = img.buffers(dev, n_items, H, W, private=True)
        self.img_bufs = ib
        rows_c = (C.c_int64 * 4)(*rows)
        self.iarena = u8(L.imf_resunet_int_arena_bytes_cap(C.byref(net), rows_c, grid_words))
        self.farena = u8(L.imf_resunet_float_arena_bytes_cap(C.byref(net), rows_c))
        self.out = self.outbuf[lay["F"]:lay["F"] + rows[0] * net.out_channels * 4].view(torch.float32).view(rows[0], net.out_channels)
        self.events = [L.imf_event_create() for _ in range(13)]   # [11], [12]: diagnostic marks around the fusion (runner.diag_events)
        io = self.io = FragmentIO()
        io.xyz, io.xyz_is_f64, io.voxel_size = self.xyz.data_ptr(), int(is_f64), float(voxel)
        io.dyn, io.image, io.meta = self.dyn.data_ptr(), self.image.data_ptr(), self.meta.data_ptr()
        io.pyramid_arena, io.pyramid_arena_bytes = self.pyr.data_ptr(), self.pyr.numel()
        io.image_ws, io.image_ws_bytes = ib["ws"].data_ptr(), ib["nbytes"]
        io.kt_packed, io.v_packed, io.tokens_padded = ib["kt"].data_ptr(), ib["vp"].data_ptr(), ib["tp"]
        io.int_arena, io.int_arena_bytes = self.iarena.data_ptr(), self.iarena.numel()
        io.float_arena, io.float_arena_bytes = self.farena.data_ptr(), self.farena.numel()
        io.out = self.out.data_ptr()
        for i, e in enumerate(self.events):
            io.events[i] = e if i < 11 or runner.diag_events else None
        io.side_stream, io.image_stream = runner.raw_streams(dev)
        io.trace = None
        io.fp32_buffers = 1 if os.environ.get("IMFNET_FP32_BUFFERS") == "1" else 0   # (see model/plan.py)
        self.graph = C.c_void_p()
        self.n_nodes = 0
        self.launches = 0
        self.last_use = 0            # FragmentRunner's use counter at the last bucket() / acquire: eviction is LRU
        self.in_flight = False       # owned by a streaming job right now (FragmentStreamer): never evicted
        self._first_view = None

    def first_idx_view(self):
        if self._first_view is None:
            off = self.io.levels[0].first_idx - self.pyr.data_ptr()
            self._first_view = self.pyr[off:off + 4 * self.caps.rows[0]].view(torch.int32)
        return self._first_view

    def enqueue(self, runner, stream, trace=None, reuse_event=None):
        """All launches of one fragment on `stream` (+ the bucket's side / image streams), eagerly.
        reuse_event (a recorded torch.cuda.Event, or None): the forwards of SEVERAL buckets are issued back to back on one
        main stream and this one's head (table reset, level-0 pyramid, image fork) goes to the side stream, under the
        previous forward's decoder (imf_fragment_io.head_on_side) -- the event marks the end of everything that still
        touches THIS bucket's buffers (its own previous forward and whatever read its outputs); inputs are in place."""
        self.io.main_stream = stream.cuda_stream
        self.io.trace = trace
        # (the streaming pipeline shares these buckets and sets head_on_side per job: a plain direct launch inherits the
        # main stream's order instead -- the bucket's previous forward may still be running on it)
        self.io.head_on_side, self.io.inputs_event, self.io.reuse_event = 0, None, None
        if reuse_event is not None:
            self.io.head_on_side, self.io.reuse_event = 1, C.c_void_p(reuse_event.cuda_event)
        check(self.L.imf_fragment_forward(C.byref(runner.net_desc), C.byref(runner.img_plan.desc), C.byref(self.caps),
                                          C.byref(self.io)), "imf_fragment_forward")

    def capture(self, runner, stream):
        L = self.L
        self.enqueue(runner, stream)                 # warm-up: lazy module loads, function attributes
        stream.synchronize()
        check(L.imf_graph_begin_capture(stream.cuda_stream), "imf_graph_begin_capture")
        try:
            self.enqueue(runner, stream)
        except Exception:
            L.imf_graph_abort_capture(stream.cuda_stream)
            raise
        n = C.c_int(0)
        check(L.imf_graph_end_capture(stream.cuda_stream, C.byref(self.graph), C.byref(n)), "imf_graph_end_capture")
        self.n_nodes = n.value

    def drop_graph(self):
        """Forget the captured graph (a setting baked into it changed); the next graph launch captures again."""
        if self.graph:
            torch.cuda.synchronize()
            self.L.imf_graph_destroy(self.graph)
            self.graph = C.c_void_p()
            self.n_nodes = 0

    def __del__(self):
        try:
            if self.graph:
                self.L.imf_graph_destroy(self.graph)
            for e in self.events:
                self.L.imf_event_destroy(e)
        except Exception:                             # noqa: BLE001 -- interpreter shutdown
            pass


class FragmentRunner:
    """Capacity-mode / hipGraph front end of one model (eval mode, IMFNet's configuration)."""

    MARGIN = 1.2          # head room over the largest voxel-per-point ratio seen
    MAX_BUCKETS = 32      # (the streaming pipeline keeps up to three of one capacity key in flight: `lane`)

    def __init__(self, model):
        from .plan import FusedPlan, NativePlan
        self.L = _lib.lib()
        self.model = model
        if not model._can_fuse():
            raise ImfError("FragmentRunner needs the model in eval mode with BatchNorm blocks")
        if model._plan is None:
            model._plan = FusedPlan(model)
        if model._native_plan is None:
            model._native_plan = NativePlan(model, model._plan)
        # the descriptors below hold raw pointers into these plans' packed weights: keep them alive with the runner
        self._plans = (model._plan, model._native_plan)
        self.net_desc = model._native_plan.desc
        self.img_plan = model._native_image()
        fw = model._fusion_weights()
        variants = {c.variant for c in self.net_desc.conv if c.w_packed}
        # one arithmetic throughout: variant 6 (split-f16), 3 (bf16x3) or 0 (fp32 MFMA: the strict-fp32 path)
        self.variant = variants.pop() if len(variants) == 1 else None
        self.supported = bool(self.img_plan.supported and self.img_plan.with_kv and fw.supported and
                              model._plan.small_first and model.conv1.in_channels == 1 and self.variant in (0, 3, 6))
        # imf_fragment_forward calls imf_image_branch (csrc/image.hip) itself: a runner exists only when that plan is usable
        self.image_branch_mode = "native-hip (csrc/image.hip, inside imf_fragment_forward)" if self.supported else None
        self.ratios = None            # max rows_l / n_points seen (4 levels)
        self.grid_words = 0           # largest conv1 bit grid seen
        self.grid_words_items = {}    # the same for batches of several fragments (n_items -> words), `observe_batch`
        self.buckets = {}
        self._main = {}
        self._raw = {}
        self._streamers = {}
        # hipGraph replay is opt-in: ROCm 7.2 runs a graph's independent branches back to back (measured 1.77 vs
        # 1.37 ms per fragment pair), the eager capacity-mode call keeps the three streams concurrent
        self.use_graph = bool(os.environ.get("IMFNET_FRAGMENT_GRAPH"))
        self.diag_events = bool(os.environ.get("IMFNET_DIAG_EVENTS"))   # tools/branch_times.py: two extra marks per step
        self.stats = dict(graph=0, eager=0, redone=0, captured=0)
        self.host_slots = []          # pinned staging of the synchronous host-array path (extract.py)
        self.stream_state = None      # streams + pinned slots of extract_features_stream
        self.cli_slots = None         # pinned slots of generate_desc's pipelined loop

    # -- capacity policy ----------------------------------------------------------------------------
    def observe(self, n_points, counts, bbox):
        r = [c / float(n_points) for c in counts]
        self.ratios = r if self.ratios is None else [max(a, b) for a, b in zip(self.ratios, r)]
        box = (C.c_int32 * 8)(*bbox)
        self.grid_words = max(self.grid_words, int(self.L.imf_bitgrid_words(box, self.model.conv1.kernel_size)))

    def caps_for(self, n_points, n_items, H, W, voxel, is_f64):
        if self.ratios is None or self.grid_words == 0:
            return None
        npc = _grid_up(n_points, 1.25, 65536)
        rows, prev = [], npc
        for l in range(4):
            # from the POINT capacity, not the point count: fragments of one point-capacity class then share ONE key (a
            # test set's fragments vary continuously in size; keyed by their own counts the 433-fragment emulation went
            # through more buckets than MAX_BUCKETS holds and rebuilt them over and over: 107 -> see DESIGN 4e)
            want = self.ratios[l] * npc * self.MARGIN
            c = min(_grid_up(want, 1.125, (4096, 1024, 256, 128)[l]), prev)
            rows.append(c)
            prev = c
        # conv1's bit grid spans the batch's common bounding box once per item: for several fragments in one forward it is
        # learned from the boxes such batches produced (`observe_batch`), starting from a guess
        words = self.grid_words if n_items == 1 else self.grid_words_items.get(n_items, self.grid_words * n_items * 2)
        gw = _grid_up(words * 1.5, 2.0, 1 << 18)
        return (npc, tuple(rows), n_items, H, W, gw, float(voxel), bool(is_f64))

    def observe_batch(self, n_items, bbox):
        """The bounding box a batch of `n_items` fragments produced (FragmentResult.bbox): sizes the next such bucket's grid."""
        box = (C.c_int32 * 8)(*bbox)
        w = int(self.L.imf_bitgrid_words(box, self.model.conv1.kernel_size))
        self.grid_words_items[n_items] = max(self.grid_words_items.get(n_items, 0), w)

    def raw_streams(self, dev):
        """(side, image) hipStream_t of this device -- and, with them, the runner's MAIN stream (`main_stream`).  The three
        are created through the library, one right after the other.  Two reasons: torch's stream pool wraps around after
        32 streams and would eventually hand out the main stream again; and HIP multiplexes streams onto a few hardware
        queues (GPU_MAX_HW_QUEUES, 4 by default; a new stream goes to the least-used queue), so a main stream created at
        some other time by somebody else can land on the queue of this runner's side or image stream -- the three
        branches of a fragment then run one after the other.  Measured (round 3, tools/e2e_probe.py): the same forward
        0.76 or 2.0 ms, the host-array stream 1.05 or 2.6-3.1 ms per fragment, depending only on which torch streams the
        process had created before.  Streams born together sit on different queues."""
        s = self._raw.get(dev)
        if s is None:
            from .. import ops
            raw, views = ops.aux_streams(dev)
            self._main[dev] = views[0]
            s = self._raw[dev] = (raw[1], raw[2])
        return s

    def streamer(self, dev, **kw):
        """The device's FragmentStreamer (stream.py): pinned host blocks in, pinned host blocks out, transfers overlapped."""
        st = self._streamers.get(dev)
        if st is None:
            from ..stream import FragmentStreamer
            st = self._streamers[dev] = FragmentStreamer(self, dev, **kw)
        return st

    def main_stream(self, dev):
        """The stream every capacity-mode forward of this runner is issued on (a torch view of the library-made stream)."""
        self.raw_streams(dev)
        return self._main[dev]

    def bucket(self, key, dev, stream=None, lane=0):
        """The capacity bucket of `key`; lane > 0: further buckets of the same capacities (forwards in flight side by side
        in the streaming pipeline need their own buffers)."""
        bk = key if lane == 0 else (key, lane)
        self._use_tick = getattr(self, "_use_tick", 0) + 1
        b = self.buckets.get(bk)
        if b is None:
            if len(self.buckets) >= self.MAX_BUCKETS:           # drop the least RECENTLY used bucket no job is using
                idle = [k for k, v in self.buckets.items() if not v.in_flight]
                if idle:
                    victim = min(idle, key=lambda k: self.buckets[k].last_use)
                    torch.cuda.synchronize(dev)                 # (a direct launch on it may still be running)
                    self.drop_bucket(victim)
            with torch.cuda.stream(stream or torch.cuda.current_stream(dev)):   # static tables are built on it
                b = self.buckets[bk] = _Bucket(self, key, dev)
                # the zero fills of the new blocks are queued on THIS stream, behind whatever forwards are in flight; the
                # pipeline uploads a job's inputs on the image stream, which does not wait for this one: without the wait
                # a fill could land on top of the first upload (seen once in ~20 runs as a spurious capacity redo)
                torch.cuda.current_stream(dev).synchronize()
        b.last_use = self._use_tick
        return b

    def touch(self, b):
        """Bucket b is being used without going through bucket() (a streamer re-issuing one of its idle lanes): it is the most
        recently used one for the eviction below."""
        self._use_tick = getattr(self, "_use_tick", 0) + 1
        b.last_use = self._use_tick

    def drop_bucket(self, bk):
        """Forget bucket `bk` (= key, or (key, lane)): its device blocks are released once the last holder lets go (the
        streamers drop theirs in `FragmentStreamer._evict`; pinned slots only hold weak references)."""
        b = self.buckets.pop(bk, None)
        if b is not None:
            for st in self._streamers.values():
                st.forget(b)

    def _stream_for(self, dev, stream):
        """(the runner's main stream, the caller's stream or None): work submitted on any other stream is ordered
        behind it on the main stream; the caller's stream waits for the main stream afterwards (`outer.wait_stream`)."""
        main = self.main_stream(dev)
        cur = stream or torch.cuda.current_stream(dev)
        if cur.cuda_stream == main.cuda_stream:
            return main, None
        main.wait_stream(cur)
        return main, cur

    # -- execution ------------------------------------------------------------------------------------
    def stage(self, b, xyz, item_starts, image, stream, dyn_host=None):
        """Copy the inputs into the bucket's static buffers (skipped for tensors that already ARE those buffers)
        and write the per-fragment scalars.  `dyn_host`: a pinned int32[16] the caller owns until the copy has run
        (keeps the call asynchronous; without it the scalars go through a blocking pageable copy)."""
        n = xyz.shape[0]
        with torch.cuda.stream(stream):
            if xyz.data_ptr() != b.xyz.data_ptr():
                b.xyz[:n].copy_(torch.as_tensor(xyz), non_blocking=True)
            if image.data_ptr() != b.image.data_ptr():
                b.image.copy_(torch.as_tensor(image, dtype=torch.float32), non_blocking=True)
            vals = [n, len(item_starts)] + [int(s) for s in item_starts]
            if vals != b.dyn_values:
                if dyn_host is not None:
                    dyn_host[: len(vals)] = torch.tensor(vals, dtype=torch.int32)
                    b.dyn.copy_(dyn_host, non_blocking=True)
                else:                                 # pageable source: staged by the runtime, safe to run ahead
                    b.dyn.copy_(torch.tensor(vals + [0] * (DYN_WORDS - len(vals)), dtype=torch.int32))
                b.dyn_values = vals
        return n

    def launch(self, b, n_points, n_items, stream, trace_list=None, meta_to=None, reuse_event=None):
        """One fragment on `stream`: graph replay (captured on first use) or eager capacity-mode launches (always
        when `trace_list` is given: per-convolution HIP events are appended to it as ops.TRACE records).
        reuse_event: eager launches only -- this forward's head on the side stream (_Bucket.enqueue)."""
        from .. import ops
        from .plan import NativePlan, _RB
        fp32_buffers = 1 if os.environ.get("IMFNET_FP32_BUFFERS") == "1" else 0
        if b.io.fp32_buffers != fp32_buffers:         # the mode is part of a captured graph: capture again under the new one
            b.io.fp32_buffers = fp32_buffers
            b.drop_graph()
        with torch.cuda.stream(stream):
            if trace_list is not None or not self.use_graph:
                trace = evs = None
                if trace_list is not None:
                    trace = (NetTrace * 23)()
                    evs = [ops._Ev() for _ in range(23)]
                    for i, e in enumerate(evs):
                        trace[i].ev_begin, trace[i].ev_end, trace[i].launched = e.begin, e.end, 0
                # imf_fragment_io.gpu_idle_hint: nothing of this runner's is still running (its last forward's end event has
                # fired) -- a synchronous call; the executor then issues its side chain piecewise (LAB_NOTES 4g-11)
                last = getattr(self, "_last_done", None)
                b.io.gpu_idle_hint = 1 if (last is None or (last is not False and last.query())) else 0
                b.enqueue(self, stream, trace, reuse_event=reuse_event)
                b.io.gpu_idle_hint = 0
                self.stats["eager"] += 1
            else:
                if not b.graph:
                    b.capture(self, stream)
                    self.stats["captured"] += 1
                check(self.L.imf_graph_launch(b.graph, stream.cuda_stream), "imf_graph_launch")
                self.stats["graph"] += 1
            if meta_to is not None:                   # (pinned int32 view, event): the caller's own copy carries the counts
                host, done = meta_to
            else:
                host = done = None
                for i, (h, d) in enumerate(b.pool):   # a pair whose previous copy has landed (never wait here)
                    if d.query():
                        host, done = b.pool.pop(i)
                        break
                if host is None:
                    host, done = torch.zeros(META_WORDS, dtype=torch.int32).pin_memory(), torch.cuda.Event()
                host.copy_(b.meta, non_blocking=True)
                done.record(stream)
            self._last_done = done if meta_to is None else False   # (the caller's event: recorded by the caller, later -- no hint)
        b.launches += 1
        res = FragmentResult(b, n_points, n_items, host, done, pooled=meta_to is None)
        if trace_list is not None:
            arena = b.iarena.view(torch.int32)
            for i, e in enumerate(evs):
                t = trace[i]
                if not t.launched:
                    continue
                rb = _RB(t.n_slots, t.n_out, t.kvol, t.kvol)
                rb.nbr = t.nbr or 0
                trace_list.append(dict(kernel=ops.conv_kernel_name(self.net_desc.conv[i].variant, t.cin, t.cout,
                                                                   kernel_tag=t.kernel_tag),
                                       kvol=t.kvol, cin=t.cin, cout=t.cout, rb=rb, split=t.split, ev=e,
                                       name=NativePlan.ORDER[i], arena=arena, res=res, level=t.level,
                                       slots_extra=t.slots_extra, kernel_tag=t.kernel_tag))
        return res

    def run(self, xyz, item_starts, image, voxel, stream=None):
        """xyz [N,3] (device tensor, f64/f32), image [B,3,H,W] device tensor.  Returns a FragmentResult, or None
        when no capacities are known yet (the caller runs the exact path and calls observe())."""
        dev = xyz.device
        key = self.caps_for(xyz.shape[0], len(item_starts), image.shape[2], image.shape[3], voxel,
                            xyz.dtype == torch.float64)
        if key is None:
            return None
        stream, outer = self._stream_for(dev, stream)
        b = self.bucket(key, dev, stream)
        n = self.stage(b, xyz, item_starts, image, stream)
        res = self.launch(b, n, len(item_starts), stream)
        if outer is not None:
            outer.wait_stream(stream)
        return res
