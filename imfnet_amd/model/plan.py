"""Arena executor for the fused ResUNet forward.

The layer-at-a-time path (`sparse._ConvBase.run` -> `ops.spconv`) allocates a tensor, fills a fresh
ctypes struct and validates arguments for every launch: ~35 us of Python per convolution, which at
1-2 ms per fragment made the host the bottleneck.  Here the whole sparse part of one forward is
planned with integer arithmetic: three device allocations per fragment (rulebook words, feature
floats, split-K workspace), pre-built `imf_conv_args` structs whose static half (weights, folded
BatchNorm, epilogue flags) is filled once per model, and raw-pointer updates per fragment.  The
kernels, their order and their arithmetic are exactly those of the layer-at-a-time path
(`tests/test_gpu_parity.py::test_fused_equals_layerwise` compares the two).
"""
import ctypes as C

import torch

from .. import _lib, ops
from .._lib import ConvArgs, ImfError, MASK_WORDS, TILE_ROWS, check


import os
_POISON = bool(os.environ.get("IMF_POISON"))


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _RB:
    """Rulebook living inside the plan's int32 arena (raw device addresses)."""
    __slots__ = ("tile_rows", "nbr", "tile_mask", "n_slots", "n_out", "kvol", "max_active", "level")

    def count_pairs(self, arena, valid_slots=None, rows=None):
        """Valid (input,output) pairs -- bench.py's algorithmic-bytes accounting, outside timing.  Capacity mode:
        only the first `valid_slots` slots of every offset were written (`rows` actual output rows)."""
        if self.kvol == 1:
            return self.n_out if rows is None else rows
        start = (self.nbr - arena.data_ptr()) // 4
        tab = arena[start:start + self.kvol * self.n_slots].view(self.kvol, self.n_slots)
        if valid_slots is not None:
            tab = tab[:, :valid_slots]
        return int((tab >= 0).sum().item())

    def __init__(self, n_slots, n_out, kvol, max_active, level=0):
        self.tile_rows = self.nbr = self.tile_mask = 0
        self.n_slots, self.n_out, self.kvol, self.max_active = n_slots, n_out, kvol, max_active
        self.level = level                            # pyramid level of the OUTPUT rows (picks the kernel)

    def words(self):
        return self.n_slots + self.kvol * self.n_slots + self.n_slots // TILE_ROWS * MASK_WORDS

    def place(self, base):
        self.tile_rows = base
        self.nbr = base + 4 * self.n_slots
        self.tile_mask = self.nbr + 4 * self.kvol * self.n_slots
        return base + 4 * self.words()


class FusedPlan:
    """Built once per model (eval mode); `run` executes one fragment."""

    def __init__(self, model):
        self.model = model
        self.L = _lib.lib()
        bn = model._bn()
        self._keep = []                     # tensors the static struct fields point into
        self.convs = {}

        def conv_args(name, module, norm=None, relu=False, l2norm=False):
            a = ConvArgs()
            a.w_packed = module.packed().data_ptr()
            a.variant = ops.conv_variant_for(module.kernel_volume)
            a.kvol, a.cout = module.kernel_volume, module.out_channels
            scale = shift = None
            if norm is not None:
                scale, shift = bn[norm]
            elif module.bias is not None:
                shift = module.bias.detach().reshape(-1).contiguous()
            for t in (scale, shift):
                if t is not None:
                    self._keep.append(t)
            a.scale = None if scale is None else scale.data_ptr()
            a.shift = None if shift is None else shift.data_ptr()
            a.relu, a.l2norm = int(relu), int(l2norm)
            self.convs[name] = (a, module)

        m = model
        self.small_first = m.conv1.in_channels <= 4
        if self.small_first:
            self.first_kernel = m.conv1.kernel3().detach().contiguous()
            self.first_bn = bn["norm1"]
        else:
            conv_args("conv1", m.conv1, "norm1")
        for i in (1, 2, 3, 4):
            if i > 1:
                conv_args(f"conv{i}", getattr(m, f"conv{i}"), f"norm{i}")
            blk = getattr(m, f"block{i}")
            conv_args(f"block{i}.conv1", blk.conv1, f"block{i}.norm1", relu=True)
            conv_args(f"block{i}.conv2", blk.conv2, f"block{i}.norm2", relu=True)
        for i in (4, 3, 2):
            conv_args(f"conv{i}_tr", getattr(m, f"conv{i}_tr"), f"norm{i}_tr")
            blk = getattr(m, f"block{i}_tr")
            conv_args(f"block{i}_tr.conv1", blk.conv1, f"block{i}_tr.norm1", relu=True)
            conv_args(f"block{i}_tr.conv2", blk.conv2, f"block{i}_tr.norm2", relu=True)
        conv_args("conv1_tr", m.conv1_tr, relu=True)
        conv_args("final", m.final, l2norm=bool(m.normalize_feature))
        self.first_ksize = m.conv1.kernel_size
        self._trace_arena = None
        self._n_items = 0
        self._side = {}

    # -------------------------------------------------------------------------------------------
    def _launch(self, name, rb, in_a, c_a, out, in_b=0, c_b=0, residual=0, ws=(0, 0)):
        a, module = self.convs[name]
        e = self._ready.pop(id(rb), None)
        if e is not None:
            self._main.wait_event(e)            # join the side stream that built this rulebook
        if c_a + c_b != module.in_channels:
            raise ImfError(f"{name}: expected {module.in_channels} input channels, got {c_a + c_b}")
        a.in_a, a.in_b, a.c_a, a.c_b = in_a, (in_b or None), c_a, c_b
        a.tile_rows, a.nbr, a.tile_mask = (rb.tile_rows or None), (rb.nbr or None), (rb.tile_mask or None)
        a.n_slots, a.n_out = rb.n_slots, rb.n_out
        a.residual = residual or None
        a.out = out
        # the executors' static kernel policy (csrc/executor.hip): a function of the output level and the layer's channels only, never split-K
        split = 1
        a.split_k = split
        a.kernel_tag = self.L.imf_resunet_conv_kernel_tag(rb.level, a.kvol, c_a + c_b, a.cout, a.variant, self._n_items)
        a.workspace, a.workspace_bytes = (ws[0] or None, ws[1])
        a.tickets = None
        a.dyn_err = self._flags if (a.variant == 6 and not a.l2norm) else None
        ev = None
        if ops.TRACE is not None:
            ev = ops._Ev()
            a.ev_begin, a.ev_end = ev.begin, ev.end
        else:
            a.ev_begin = a.ev_end = None
        check(self.L.imf_spconv_fwd(C.byref(a), _stream()), f"imf_spconv_fwd[{name}]")
        if ev is not None:
            cin = c_a + c_b
            ops.TRACE.append(dict(kernel=ops.conv_kernel_name(a.variant, cin, a.cout, kernel_tag=a.kernel_tag),
                                  kvol=rb.kvol, cin=cin, cout=a.cout, rb=rb, split=split, ev=ev, name=name,
                                  arena=self._trace_arena))

    def run(self, x, fuse, after_fuse=None, n_items=0):
        """x: SparseTensor at tensor stride 1 (pyramid built); fuse(F8 [n8,C]) -> [n8,C] is the
        bottleneck fusion (torch).  Returns the [M, out] descriptor tensor."""
        m, L = self.model, self.L
        self._n_items = int(n_items)            # batch of this forward: part of the executors' static kernel policy
        cm = x.coordinate_manager
        lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
        n = [l.n for l in lv]
        dev = x.F.device
        st = _stream()
        Ch, T = m.CHANNELS, m.TR_CHANNELS
        out_ch = m.final.out_channels

        # ---- rulebooks: one int32 arena ------------------------------------------------------
        slots = [L.imf_rulebook_slots(k) for k in n]
        rb_first = _RB(slots[0], n[0], self.first_ksize ** 3, self.first_ksize ** 3)
        rb_k3 = [_RB(slots[i], n[i], 27, 27, level=i) for i in range(4)]
        rb_dn = [_RB(slots[i + 1], n[i + 1], 27, 27, level=i + 1) for i in range(3)]
        rb_up = [_RB(L.imf_rulebook_transpose_slots(n[i]), n[i], 27, 8, level=i) for i in range(3)]
        rb_id = _RB(slots[0], n[0], 1, 1)
        # occupancy-sorted twins of the stride-1 maps of levels 0-2 for the decoder's blocks (csrc/rulebook_sort.hip), exactly
        # as the native executors build and use them (imf_resunet_sorted_maps says which levels): bit-identical descriptors
        sorted_maps = L.imf_resunet_sorted_maps_n(int(self.convs["block2_tr.conv1"][0].variant), self._n_items)
        rb_k3s = [_RB(slots[i], n[i], 27, 27, level=i) if ((sorted_maps >> i) & 1 and (i > 0 or self.small_first)) else None
                  for i in range(3)]
        sort_ws_bytes = L.imf_rulebook_sorted_workspace_bytes(slots[0])
        all_rb = ([] if self.small_first else [rb_first]) + rb_k3 + rb_dn + rb_up + [r for r in rb_k3s if r is not None]
        words = sum(r.words() for r in all_rb) + 16 * 3 + 64 + sort_ws_bytes // 4
        main = torch.cuda.current_stream(dev)
        side = ops.aux_streams(dev)[1][1]        # born with the geometry / image streams: distinct hardware queues
        with torch.cuda.stream(side):           # side-stream pool: no need to wait for the main stream
            iarena = torch.empty(words, dtype=torch.int32, device=dev)
        iarena.record_stream(main)
        p = iarena.data_ptr()
        for r in all_rb:
            p = r.place(p)
        counters = [p + 64 * i for i in range(3)]
        sort_ws = (p + 64 * 3 + 255) // 256 * 256
        # rb_first and k3@1 are needed at once: main stream.  Everything else is built on a side
        # stream while conv1 / block1 (MFMA-bound, whole GPU) run; each group is joined by an event
        # right before its first use.
        def build_conv(rb, in_lv, out_lv, ksize, stream):
            check(L.imf_rulebook_conv(in_lv.table.data_ptr(), in_lv.capacity,
                                      out_lv.coords_buf.data_ptr(), out_lv.n, in_lv.ts, ksize, rb.tile_rows,
                                      rb.nbr, rb.tile_mask, stream), "imf_rulebook_conv")

        def build_up(i, stream):
            check(L.imf_rulebook_transpose(lv[i + 1].table.data_ptr(),
                                           lv[i + 1].capacity, lv[i].coords_buf.data_ptr(), n[i], 1 << i, 3,
                                           rb_up[i].tile_rows, rb_up[i].nbr, rb_up[i].tile_mask,
                                           rb_up[i].n_slots, counters[i], stream), "imf_rulebook_transpose")

        ss = side.cuda_stream
        ready = {}                              # rulebook object id -> event
        if not self.small_first:
            build_conv(rb_first, lv[0], lv[0], self.first_ksize, st)
            build_conv(rb_k3[0], lv[0], lv[0], 3, st)
        else:                                   # conv1 needs no rulebook: k3@1 is built under it
            build_conv(rb_k3[0], lv[0], lv[0], 3, ss)
            e = torch.cuda.Event()
            e.record(side)
            ready[id(rb_k3[0])] = e
        for i in range(3):
            build_conv(rb_dn[i], lv[i], lv[i + 1], 3, ss)
            build_conv(rb_k3[i + 1], lv[i + 1], lv[i + 1], 3, ss)
            e = torch.cuda.Event()
            e.record(side)
            ready[id(rb_dn[i])] = e
        for i in (2, 1, 0):
            build_up(i, ss)
            e = torch.cuda.Event()
            e.record(side)
            ready[id(rb_up[i])] = e
        for i in (2, 1, 0):                     # (small_first: the level-0 map was built on the side stream too, above)
            if rb_k3s[i] is None:
                continue
            check(L.imf_rulebook_sort_by_occupancy(rb_k3[i].nbr, 27, rb_k3[i].n_slots, n[i], None, rb_k3s[i].tile_rows,
                                                   rb_k3s[i].nbr, rb_k3s[i].tile_mask, sort_ws, sort_ws_bytes, ss),
                  "imf_rulebook_sort_by_occupancy")
            e = torch.cuda.Event()
            e.record(side)
            ready[id(rb_k3s[i])] = e
        self._ready, self._main = ready, main
        self._flags = m.flag_word(dev).data_ptr()

        # ---- schedule: (conv name, rulebook, in_a, c_a, out, in_b, c_b, residual) ---------------
        sched = []
        for i in range(4):
            c = Ch[i + 1]
            if i > 0:
                sched.append((f"conv{i + 1}", rb_dn[i - 1], f"e{i - 1}c", Ch[i], f"e{i}a", None, 0, None))
            elif not self.small_first:
                sched.append(("conv1", rb_first, "x", x.F.shape[1], "e0a", None, 0, None))
            sched.append((f"block{i + 1}.conv1", rb_k3[i], f"e{i}a", c, f"e{i}b", None, 0, None))
            sched.append((f"block{i + 1}.conv2", rb_k3[i], f"e{i}b", c, f"e{i}c", None, 0, f"e{i}a"))
        n_enc = len(sched)
        dec_ch = {2: T[4], 1: T[3], 0: T[2]}
        for i in (2, 1, 0):                                # output level of conv{i+2}_tr
            t = dec_ch[i]
            src, c_src = ("fused", Ch[4]) if i == 2 else (f"d{i + 1}c", dec_ch[i + 1])
            skip, c_skip = (None, 0) if i == 2 else (f"e{i + 1}c", Ch[i + 2])
            sched.append((f"conv{i + 2}_tr", rb_up[i], src, c_src, f"d{i}a", skip, c_skip, None))
            rbk = rb_k3s[i] if rb_k3s[i] is not None else rb_k3[i]
            sched.append((f"block{i + 2}_tr.conv1", rbk, f"d{i}a", t, f"d{i}b", None, 0, None))
            sched.append((f"block{i + 2}_tr.conv2", rbk, f"d{i}b", t, f"d{i}c", None, 0, f"d{i}a"))
        sched.append(("conv1_tr", rb_id, "d0c", T[2], "head", "e0c", Ch[1], None))
        sched.append(("final", rb_id, "head", T[1], "F", None, 0, None))

        # ---- features: one float arena (+ the largest split-K workspace any launch needs) ------
        sizes = {}
        for i in range(4):
            for sfx in "abc":                              # conv out, block mid, block out
                sizes[f"e{i}{sfx}"] = n[i] * Ch[i + 1]
        for i in (2, 1, 0):
            for sfx in "abc":
                sizes[f"d{i}{sfx}"] = n[i] * dec_ch[i]
        sizes["head"] = n[0] * T[1]
        ws_floats = 0
        for name, rb, *_ in sched:
            cout = self.convs[name][0].cout
            ws_floats = max(ws_floats, L.imf_spconv_workspace_bytes(rb.n_slots, cout, 1) // 4)
        farena = torch.empty(sum(sizes.values()) + ws_floats, dtype=torch.float32, device=dev)
        base, off, addr, foff = farena.data_ptr(), 0, {}, {}
        for name, cnt in sizes.items():
            addr[name], foff[name] = base + 4 * off, off
            off += cnt
        ws = (base + 4 * off, 4 * ws_floats) if ws_floats else (0, 0)
        F = torch.empty((n[0], out_ch), dtype=torch.float32, device=dev)
        addr["F"], addr["x"] = F.data_ptr(), x.F.data_ptr()
        self._trace_arena = iarena

        if self.small_first:
            sc, sh = self.first_bn
            ones = getattr(x, "_all_ones", False)       # util/misc.py:76-79 occupancy feature
            bbox = getattr(lv[0], "bbox", None)
            words = 0
            if ones and bbox is not None and x.F.shape[1] == 1:
                box = (C.c_int32 * 8)(*bbox)
                words = L.imf_bitgrid_words(box, self.first_ksize)
            if words:
                grid = torch.empty(words, dtype=torch.int32, device=dev)
                check(L.imf_conv_first_bitgrid(lv[0].coords_buf.data_ptr(), n[0], box, self.first_ksize,
                                               grid.data_ptr(), words, self.first_kernel.data_ptr(), Ch[1],
                                               sc.data_ptr(), sh.data_ptr(), 0, addr["e0a"], st),
                      "imf_conv_first_bitgrid")
            else:
                check(L.imf_conv_first_fused(lv[0].table.data_ptr(), lv[0].capacity,
                                             lv[0].coords_buf.data_ptr(), n[0], 1, self.first_ksize,
                                             None if ones else x.F.data_ptr(), x.F.shape[1],
                                             self.first_kernel.data_ptr(), Ch[1], sc.data_ptr(), sh.data_ptr(),
                                             0, addr["e0a"], st), "imf_conv_first_fused")

        def go(entries):
            for name, rb, a_key, c_a, o_key, b_key, c_b, r_key in entries:
                self._launch(name, rb, addr[a_key], c_a, addr[o_key], in_b=addr[b_key] if b_key else 0,
                             c_b=c_b, residual=addr[r_key] if r_key else 0, ws=ws)

        go(sched[:n_enc])                                                       # encoder
        f8 = farena[foff["e3c"]:foff["e3c"] + sizes["e3c"]].view(n[3], Ch[4])   # bottleneck fusion
        fused = fuse(f8).contiguous()
        addr["fused"] = fused.data_ptr()
        if after_fuse is not None:      # harness hook: e.g. queue the NEXT fragment's geometry / image branch
            after_fuse()                # now, so it runs under this fragment's decoder
        go(sched[n_enc:])                                                       # decoder + head
        return F


class NativePlan:
    """One C call per fragment: `imf_resunet_forward` (csrc/executor.hip) walks the schedule that
    FusedPlan.run issues from Python (same launches, order and buffers), cutting ~0.5 ms of
    interpreter time per forward to ~0.05 ms.  Covers IMFNet's configuration: BatchNorm variants, the
    fused HIP fusion kernel (batch 1, one head, depth 0).  Built once per model from FusedPlan's static
    convolution table; everything else is the FusedPlan path."""

    ORDER = (["conv1", "block1.conv1", "block1.conv2", "conv2", "block2.conv1", "block2.conv2", "conv3",
              "block3.conv1", "block3.conv2", "conv4", "block4.conv1", "block4.conv2", "conv4_tr",
              "block4_tr.conv1", "block4_tr.conv2", "conv3_tr", "block3_tr.conv1", "block3_tr.conv2", "conv2_tr",
              "block2_tr.conv1", "block2_tr.conv2", "conv1_tr", "final"])

    def __init__(self, model, fused):
        from .._lib import NetTrace, ResunetDesc, ResunetIO
        self.model, self.fused, self.L = model, fused, _lib.lib()
        m = model
        d = self.desc = ResunetDesc()
        for i in range(1, 5):
            d.channels[i], d.tr_channels[i] = m.CHANNELS[i], m.TR_CHANNELS[i]
        d.in_channels, d.out_channels = m.conv1.in_channels, m.final.out_channels
        d.first_ksize, d.small_first = m.conv1.kernel_size, int(fused.small_first)
        for i, name in enumerate(self.ORDER):
            if name not in fused.convs:
                continue
            a, module = fused.convs[name]
            c = d.conv[i]
            c.w_packed, c.kvol, c.cin, c.cout = a.w_packed, a.kvol, module.in_channels, a.cout
            c.scale, c.shift, c.relu, c.l2norm, c.variant = a.scale, a.shift, a.relu, a.l2norm, a.variant
        if fused.small_first:
            d.first_kernel = fused.first_kernel.data_ptr()
            d.first_scale, d.first_shift = (t.data_ptr() for t in fused.first_bn)
            k = fused.first_kernel
            if k.is_cuda and k.shape[1] == 1 and k.shape[0] in (27, 125) and k.shape[2] in (32, 64) and \
                    all(c.variant in (6, 3) for c in d.conv if c.w_packed):
                # conv1's weights as the f16 matrix pipe reads them, split once here instead of by every workgroup
                self.first_image = torch.empty(self.L.imf_first_kernel_image_floats(k.shape[0], k.shape[2]),
                                               dtype=torch.float32, device=k.device)
                check(self.L.imf_pack_first_kernel(k.data_ptr(), k.shape[0], k.shape[2], self.first_image.data_ptr(),
                                                   torch.cuda.current_stream(k.device).cuda_stream), "imf_pack_first_kernel")
                d.first_kernel_image = self.first_image.data_ptr()
        fw = model._fusion_weights()
        self.fw = fw                                  # keeps the packed tensors alive
        d.fusion, d.fusion_scale = fw.c, fw.scale
        self.io = ResunetIO()
        self._events = [self.L.imf_event_create() for _ in range(10)]
        for i, e in enumerate(self._events):
            self.io.events[i] = e
        self._side = {}
        self._bbox = (C.c_int32 * 8)()
        self._n = (C.c_int64 * 4)()
        self._trace = (NetTrace * 23)()
        self._trace_events = None

    def __del__(self):
        try:
            for e in self._events:
                self.L.imf_event_destroy(e)
        except Exception:                             # noqa: BLE001 -- interpreter shutdown
            pass

    def run(self, x, packed, items, image_ready, fusion_done):
        """x: stride-1 SparseTensor with the pyramid built; packed = ([K^T per item], [V per item], n_tokens,
        tokens_padded); items = [(first stride-8 row, rows)] per batch item;
        image_ready / fusion_done: torch.cuda.Event (already recorded / to be recorded).  Returns F."""
        L, io, d = self.L, self.io, self.desc
        cm = x.coordinate_manager
        lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
        dev = x.F.device
        main = torch.cuda.current_stream(dev)
        side = ops.aux_streams(dev)[1][1]        # born with the geometry / image streams: distinct hardware queues
        for i, l in enumerate(lv):
            ld = io.level[i]
            ld.coords, ld.table = l.coords_buf.data_ptr(), l.table.data_ptr()
            ld.capacity, ld.tensor_stride = l.capacity, l.ts
            self._n[i] = io.n[i] = l.n
        bbox = getattr(lv[0], "bbox", None)
        if bbox is not None:
            self._bbox[:] = list(bbox)
            io.bbox = C.addressof(self._bbox)
        else:
            io.bbox = None
        io.x, io.x_all_ones = x.F.data_ptr(), int(bool(getattr(x, "_all_ones", False)))
        io.n_items = len(items)
        for b, (r0, rn) in enumerate(items):
            io.item_row0[b], io.item_rows[b] = int(r0), int(rn)
            io.kt_packed[b], io.v_packed[b] = packed[0][b].data_ptr(), packed[1][b].data_ptr()
        io.n_tokens, io.tokens_padded = int(packed[2]), int(packed[3])
        ibytes = L.imf_resunet_int_arena_bytes(C.byref(d), self._n, io.bbox)
        fbytes = L.imf_resunet_float_arena_bytes(C.byref(d), self._n)
        with torch.cuda.stream(side):                 # side-stream pool: rulebooks are written there first
            iarena = torch.empty(ibytes, dtype=torch.uint8, device=dev)
        iarena.record_stream(main)
        farena = torch.empty(fbytes, dtype=torch.uint8, device=dev)
        F = torch.empty((lv[0].n, d.out_channels), dtype=torch.float32, device=dev)
        if _POISON:                                   # debugging aid: a read-before-write shows up as NaN
            farena.view(torch.float32).fill_(float("nan"))
            F.fill_(float("nan"))
            with torch.cuda.stream(side):
                iarena.view(torch.int32).fill_(0x7FC00000)
            main.wait_stream(side)
        io.int_arena, io.int_arena_bytes = iarena.data_ptr(), ibytes
        io.float_arena, io.float_arena_bytes = farena.data_ptr(), fbytes
        io.out = F.data_ptr()
        io.image_ready = image_ready.cuda_event if image_ready is not None else None
        fusion_done.record(main)                      # creates the handle; re-recorded natively after the fusion
        io.fusion_done = fusion_done.cuda_event
        io.side_stream, io.main_stream = side.cuda_stream, main.cuda_stream
        tracing = ops.TRACE is not None
        if tracing:
            evs = [ops._Ev() for _ in range(23)]
            for i, e in enumerate(evs):
                self._trace[i].ev_begin, self._trace[i].ev_end, self._trace[i].launched = e.begin, e.end, 0
            io.trace = self._trace
        else:
            io.trace = None
        io.flags = self.model.flag_word(dev).data_ptr()
        # IMFNET_FP32_BUFFERS=1: every feature buffer fp32 (the arithmetic of the op-by-op executor); default: the layers
        # hand split-f16 operand images on (include/imfnet_hip.h, imf_conv_args.operand_format)
        io.fp32_buffers = 1 if os.environ.get("IMFNET_FP32_BUFFERS") == "1" else 0
        check(L.imf_resunet_forward(C.byref(d), C.byref(io)), "imf_resunet_forward")
        if tracing:
            for i, e in enumerate(evs):
                t = self._trace[i]
                if not t.launched:
                    continue
                rb = _RB(t.n_slots, t.n_out, t.kvol, t.kvol)
                rb.nbr = t.nbr or 0
                ops.TRACE.append(dict(kernel=ops.conv_kernel_name(d.conv[i].variant, t.cin, t.cout, kernel_tag=t.kernel_tag),
                                      kvol=t.kvol,
                                      cin=t.cin, cout=t.cout, rb=rb, split=t.split, ev=e, name=self.ORDER[i],
                                      arena=iarena.view(torch.int32)))
        return F
