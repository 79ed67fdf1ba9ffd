"""Tensor-level wrappers over the C ABI (include/imfnet_hip.h).  torch is used only for device
memory and streams; every computation below happens in libimfnet_hip.so on the GPU."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvArgs, HeadArgs, ImfError, LevelDesc, TILE_ROWS, MASK_WORDS, check


# When set to a list, every sparse-conv launch is bracketed by HIP events recorded on the launch
# stream right around the main kernel and appended as a dict (bench.py's live roofline
# measurement).  None in normal operation.
TRACE = None


class _Ev:
    """A raw hipEvent_t pair owned by the library side of the C ABI."""

    def __init__(self):
        L = _lib.lib()
        self.begin, self.end = L.imf_event_create(), L.imf_event_create()

    def elapsed_ms(self):
        return _lib.lib().imf_event_elapsed_ms(self.begin, self.end)

    def __del__(self):
        try:
            L = _lib.lib()
            L.imf_event_destroy(self.begin)
            L.imf_event_destroy(self.end)
        except Exception:
            pass


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _req(t, dtype, name, ndim=None):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise ImfError(f"{name} must be a CUDA(HIP) tensor -- imfnet_amd has no CPU path")
    if t.dtype != dtype:
        raise ImfError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ImfError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ImfError(f"{name} must be {ndim}-D")
    return t


class Level:
    """One coordinate map: rows `coords[:n]` (int32 (b,x,y,z)) at tensor stride `ts`, and the
    voxel hash (`table`: 16-byte {key, row} slots, struct imf_slot) mapping a coordinate to its row."""
    __slots__ = ("coords_buf", "n_dev", "n", "table", "capacity", "ts", "first_idx_buf")

    def __init__(self, coords_buf, n_dev, table, capacity, ts, first_idx_buf=None):
        self.coords_buf, self.n_dev, self.n = coords_buf, n_dev, None
        self.table, self.capacity, self.ts = table, capacity, ts
        self.first_idx_buf = first_idx_buf

    @property
    def coords(self):
        return self.coords_buf[: self.n]

    @property
    def first_idx(self):
        return self.first_idx_buf[: self.n]

    @property
    def device(self):
        return self.coords_buf.device


class _Addr:
    """A raw device address with the `.data_ptr()` face of a tensor (memory owned by an arena)."""
    __slots__ = ("p",)

    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


class ArenaLevel(Level):
    """A Level whose buffers live inside one arena built by imf_pyramid_build (raw addresses; the
    coords / first_idx tensors are materialised as views only when somebody asks for them)."""
    __slots__ = ("arena", "_desc", "_coords_view", "_first_view", "bbox", "items")

    def __init__(self, arena, desc, n_dev):
        Level.__init__(self, _Addr(desc.coords), n_dev, _Addr(desc.table), desc.capacity,
                       desc.tensor_stride, _Addr(desc.first_idx) if desc.first_idx else None)
        self.arena, self._desc, self._coords_view, self._first_view = arena, desc, None, None
        self.bbox = None              # level 0: [min b,x,y,z, max b,x,y,z] (host ints)
        self.items = None             # [(first row, rows)] per batch item

    def _view(self, addr, count):
        off = addr - self.arena.data_ptr()
        return self.arena[off:off + 4 * count].view(torch.int32)

    @property
    def coords(self):
        if self._coords_view is None:
            self._coords_view = self._view(self._desc.coords, 4 * self._desc.cap_rows).view(-1, 4)
        return self._coords_view[: self.n]

    @property
    def first_idx(self):
        if self._first_view is None:
            self._first_view = self._view(self._desc.first_idx, self._desc.cap_rows)
        return self._first_view[: self.n]

    @property
    def device(self):
        return self.arena.device


_AUX_STREAMS = {}
_PINNED = {}


def aux_streams(device):
    """The package's THREE streams of a device: (raw hipStream_t handles, torch views), created through the library one
    right after the other.  HIP multiplexes streams onto four hardware queues (a new stream goes to the least-used one),
    and torch's pool hands its streams out in an order the process's history decides: branches that should overlap then
    share a queue and run one after the other (round 3: the same forward 0.75 or 2.0 ms, the fp32 exact-mode pair 2.3 or
    3.4 ms).  Streams born together sit on different queues.  Every executor takes its auxiliary streams from here:
        capacity mode (model/graph.py, stream.py):  main, side (coarse levels + rulebooks), image branch
        exact mode (extract.py, model/plan.py):      geometry, side (rulebooks), image branch
    -- the two modes never run at the same time, and the caller's own stream is the fourth."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    s = _AUX_STREAMS.get(device)
    if s is None:
        L = _lib.lib()
        with torch.cuda.device(device):
            raw = (L.imf_stream_create(), L.imf_stream_create(), L.imf_stream_create())
        if not all(raw):
            raise ImfError("could not create the package's three streams")
        s = _AUX_STREAMS[device] = (raw, tuple(torch.cuda.ExternalStream(r, device=device) for r in raw))
    return s


def geometry_stream(device):
    """Dedicated stream for the exact-mode geometry build: its row-count readback must not queue behind the previous
    fragment's convolutions on the caller's stream."""
    return aux_streams(device)[1][0]


class PyramidFuture:
    """Geometry of one fragment queued on the geometry stream; `result()` blocks on ITS event only."""

    def __init__(self, xyz, voxel_size, n_levels=4, batch_index=0, inputs_ready=False, item_starts=None):
        if xyz.dtype not in (torch.float64, torch.float32):
            raise ImfError(f"xyz must be float64/float32, got {xyz.dtype}")
        _req(xyz, xyz.dtype, "xyz", 2)
        n, dev = xyz.shape[0], xyz.device
        if n == 0 or xyz.shape[1] != 3:
            raise ImfError(f"xyz must be [N>0, 3], got {tuple(xyz.shape)}")
        L = _lib.lib()
        self.n_levels, self.dev = n_levels, dev
        self.n_items = 1 if item_starts is None else len(item_starts)   # batch of fragments: points back to back
        if self.n_items > _lib.MAX_BATCH:
            raise ImfError(f"at most {_lib.MAX_BATCH} fragments per batch, got {self.n_items}")
        n_meta = 2 * n_levels + 8 + (_lib.MAX_BATCH * n_levels if self.n_items > 1 else 0)
        main = torch.cuda.current_stream(dev)
        gs = geometry_stream(dev)
        if not inputs_ready:
            gs.wait_stream(main)
        pool = _PINNED.setdefault((dev, n_meta), [])
        self.host = pool.pop() if pool else torch.empty(n_meta, dtype=torch.int32).pin_memory()
        self.descs = (LevelDesc * n_levels)()
        with torch.cuda.stream(gs):
            nbytes = L.imf_pyramid_arena_bytes(n, n_levels)
            self.arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.meta = torch.empty(n_meta, dtype=torch.int32, device=dev)   # counts/flags + bbox (+ item starts)
            if self.n_items > 1:
                starts = (C.c_int64 * self.n_items)(*[int(v) for v in item_starts])
                check(L.imf_pyramid_build_batched(xyz.data_ptr(), int(xyz.dtype == torch.float64), n,
                                                  float(voxel_size), starts, self.n_items, n_levels,
                                                  self.arena.data_ptr(), nbytes, self.meta.data_ptr(), self.descs,
                                                  gs.cuda_stream), "imf_pyramid_build_batched")
            else:
                check(L.imf_pyramid_build(xyz.data_ptr(), int(xyz.dtype == torch.float64), n, float(voxel_size),
                                          int(batch_index), n_levels, self.arena.data_ptr(), nbytes,
                                          self.meta.data_ptr(), self.descs, gs.cuda_stream), "imf_pyramid_build")
            self.host.copy_(self.meta, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record(gs)
        xyz.record_stream(gs)
        self.xyz = xyz
        self._levels = None

    def result(self):
        """Wait for the row counts (the one host wait of the fragment) and hand the levels to the
        CURRENT stream."""
        if self._levels is not None:
            return self._levels
        n_levels = self.n_levels
        self.ev.synchronize()
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(self.ev)
        self.arena.record_stream(main)
        self.meta.record_stream(main)
        counts = self.host.tolist()
        _PINNED[(self.dev, len(counts))].append(self.host)
        self.host = None
        levels = []
        for i in range(n_levels):
            if counts[2 * i + 1] != 0:
                raise ImfError("voxelize: a coordinate fell outside [-2^17, 2^17) voxels (or was NaN)")
            lv = ArenaLevel(self.arena, self.descs[i], self.meta[2 * i:2 * i + 2])
            lv.n = int(counts[2 * i])
            levels.append(lv)
        levels[0].bbox = counts[2 * n_levels:2 * n_levels + 8]
        for i, lv in enumerate(levels):              # (first row, rows) of every batch item at this level
            if self.n_items > 1:
                st = counts[2 * n_levels + 8 + _lib.MAX_BATCH * i:][: self.n_items]
                if any(v < 0 for v in st):
                    raise ImfError("batched pyramid: an item has no voxel")
                lv.items = [(st[b], (st[b + 1] if b + 1 < self.n_items else lv.n) - st[b]) for b in range(self.n_items)]
            else:
                lv.items = [(0, lv.n)]
        self._levels = levels
        return levels


def pyramid_from_points(xyz, voxel_size, n_levels=4, batch_index=0, inputs_ready=False, before_sync=None):
    """Voxelise + coarse levels in ONE library call on the geometry stream, then one event
    synchronisation for the row counts.  xyz: CUDA [N,3] f64/f32 tensor.  `inputs_ready`: xyz is
    known to be complete (e.g. resident data), so the geometry stream need not wait for the main one.
    Returns the list of levels (n set) -- buffers are safe to use on the current stream."""
    fut = PyramidFuture(xyz, voxel_size, n_levels, batch_index, inputs_ready)
    if before_sync is not None:
        before_sync()
    return fut.result()


class Rulebook:
    """Tiled kernel map (see include/imfnet_hip.h)."""
    __slots__ = ("tile_rows", "nbr", "tile_mask", "n_slots", "n_out", "kvol", "max_active")

    def __init__(self, tile_rows, nbr, tile_mask, n_slots, n_out, kvol, max_active=None):
        self.tile_rows, self.nbr, self.tile_mask = tile_rows, nbr, tile_mask
        self.n_slots, self.n_out, self.kvol = n_slots, n_out, kvol
        self.max_active = kvol if max_active is None else max_active   # active offsets per tile


def _new_table(n, device):
    L = _lib.lib()
    cap = L.imf_hash_capacity(n)
    table = torch.empty((cap, 2), dtype=torch.int64, device=device)      # struct imf_slot {u64 key; i32 val; i32 pad}
    ws = torch.empty(L.imf_unique_workspace_bytes(n), dtype=torch.uint8, device=device)
    return cap, table, ws


def new_meta(n_levels, device):
    """Zeroed [n_levels, 2] int32 block of (row count, error flag) words: ONE D2H copy reads every
    level's count back."""
    return torch.zeros((n_levels, 2), dtype=torch.int32, device=device)


def voxelize(xyz, voxel_size, batch_index=0, meta=None):
    """util/misc.py:82-87 on the GPU.  xyz: CUDA [N,3] float64 or float32.  Asynchronous: the
    returned Level has n=None until `sync_levels` reads the counts back."""
    if xyz.dtype not in (torch.float64, torch.float32):
        raise ImfError(f"xyz must be float64/float32, got {xyz.dtype}")
    _req(xyz, xyz.dtype, "xyz", 2)
    n, dev = xyz.shape[0], xyz.device
    if n == 0 or xyz.shape[1] != 3:
        raise ImfError(f"xyz must be [N>0, 3], got {tuple(xyz.shape)}")
    cap, table, ws = _new_table(n, dev)
    coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    first = torch.empty(n, dtype=torch.int32, device=dev)
    if meta is None:
        meta = torch.zeros(2, dtype=torch.int32, device=dev)       # [m, err]
    check(_lib.lib().imf_voxelize(xyz.data_ptr(), int(xyz.dtype == torch.float64), n, float(voxel_size),
                                  int(batch_index), coords.data_ptr(), first.data_ptr(),
                                  meta[0:1].data_ptr(), table.data_ptr(), cap,
                                  ws.data_ptr(), meta[1:2].data_ptr(), _stream()), "imf_voxelize")
    lv = Level(coords, meta, table, cap, 1, first)
    return lv


def downsample(level, out_stride, n_in_max=None, meta=None):
    """coordinate_manager.stride(): level at tensor stride `out_stride` (asynchronous)."""
    n_max = int(n_in_max if n_in_max is not None else level.n)
    dev = level.device
    cap, table, ws = _new_table(n_max, dev)
    coords = torch.empty((n_max, 4), dtype=torch.int32, device=dev)
    m = meta if meta is not None else torch.zeros(2, dtype=torch.int32, device=dev)   # [m, unused]
    check(_lib.lib().imf_downsample(level.coords_buf.data_ptr(), level.n_dev.data_ptr(), n_max,
                                    int(out_stride), coords.data_ptr(), m.data_ptr(), table.data_ptr(),
                                    cap, ws.data_ptr(), _stream()), "imf_downsample")
    return Level(coords, m, table, cap, out_stride)


def level_from_coords(coords):
    """Coordinate map of caller-supplied int32 (b,x,y,z) rows (ME.SparseTensor(coordinates=...),
    util/misc.py:95).  Duplicates collapse to their first occurrence."""
    _req(coords, torch.int32, "coordinates", 2)
    n = coords.shape[0]
    n_dev = torch.tensor([n, 0], dtype=torch.int32, device=coords.device)
    src = Level(coords, n_dev, None, 0, 1)
    src.n = n
    lv = downsample(src, 1)
    lv.ts = 1
    return lv


def sync_levels(levels, meta_block=None):
    """One host synchronisation: read the row counts of `levels` back.  `meta_block`: the shared
    [n,2] count block whose row i belongs to levels[i] (one contiguous D2H copy)."""
    if meta_block is not None:
        counts = meta_block[: len(levels)].cpu().reshape(-1).tolist()
    else:
        counts = torch.cat([lv.n_dev.reshape(-1)[:2] for lv in levels]).cpu().tolist()   # [m, err] per level
    for i, lv in enumerate(levels):
        if counts[2 * i + 1] != 0:
            raise ImfError("voxelize: a coordinate fell outside [-2^17, 2^17) voxels (or was NaN)")
        lv.n = int(counts[2 * i])


def rulebook_conv(in_level, out_level, ksize):
    """Kernel map of ME.MinkowskiConvolution(kernel_size=ksize) from in_level to out_level."""
    L = _lib.lib()
    n_out, dev = out_level.n, out_level.device
    kvol = ksize ** 3
    n_slots = L.imf_rulebook_slots(n_out)
    tile_rows = torch.empty(n_slots, dtype=torch.int32, device=dev)
    nbr = torch.empty(kvol * n_slots, dtype=torch.int32, device=dev)
    mask = torch.empty(n_slots // TILE_ROWS * MASK_WORDS, dtype=torch.int32, device=dev)
    check(L.imf_rulebook_conv(in_level.table.data_ptr(), in_level.capacity,
                              out_level.coords_buf.data_ptr(), n_out, in_level.ts, ksize,
                              tile_rows.data_ptr(), nbr.data_ptr(), mask.data_ptr(), _stream()),
          "imf_rulebook_conv")
    return Rulebook(tile_rows, nbr, mask, n_slots, n_out, kvol)


def rulebook_sorted(rb):
    """imf_rulebook_sort_by_occupancy: the occupancy-sorted twin of a stride-1 map in identity slot order (same rows, same
    inputs, tiles of rows with similar neighbour masks: fewer active (tile, offset) pairs for imf_spconv_fwd to walk)."""
    L = _lib.lib()
    dev = rb.nbr.device
    tile_rows = torch.empty(rb.n_slots, dtype=torch.int32, device=dev)
    nbr = torch.empty(rb.kvol * rb.n_slots, dtype=torch.int32, device=dev)
    mask = torch.empty(rb.n_slots // TILE_ROWS * MASK_WORDS, dtype=torch.int32, device=dev)
    ws = torch.empty(L.imf_rulebook_sorted_workspace_bytes(rb.n_slots), dtype=torch.uint8, device=dev)
    check(L.imf_rulebook_sort_by_occupancy(rb.nbr.data_ptr(), rb.kvol, rb.n_slots, rb.n_out, None, tile_rows.data_ptr(),
                                           nbr.data_ptr(), mask.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "imf_rulebook_sort_by_occupancy")
    return Rulebook(tile_rows, nbr, mask, rb.n_slots, rb.n_out, rb.kvol, rb.max_active)


def rulebook_transpose(coarse_level, fine_level, ksize=3):
    """Kernel map of ME.MinkowskiConvolutionTranspose(kernel_size=3, stride=2): coarse -> fine."""
    L = _lib.lib()
    n_fine, dev = fine_level.n, fine_level.device
    kvol = ksize ** 3
    n_slots = L.imf_rulebook_transpose_slots(n_fine)
    tile_rows = torch.empty(n_slots, dtype=torch.int32, device=dev)
    nbr = torch.empty(kvol * n_slots, dtype=torch.int32, device=dev)
    mask = torch.empty(n_slots // TILE_ROWS * MASK_WORDS, dtype=torch.int32, device=dev)
    counters = torch.empty(16, dtype=torch.int32, device=dev)
    check(L.imf_rulebook_transpose(coarse_level.table.data_ptr(),
                                   coarse_level.capacity, fine_level.coords_buf.data_ptr(), n_fine,
                                   fine_level.ts, ksize, tile_rows.data_ptr(), nbr.data_ptr(),
                                   mask.data_ptr(), n_slots, counters.data_ptr(), _stream()),
          "imf_rulebook_transpose")
    return Rulebook(tile_rows, nbr, mask, n_slots, n_fine, kvol, max_active=8)


def rulebook_identity(n_out, device):
    """kvol == 1 (pointwise) 'rulebook': slot == row, one active offset; nothing to build."""
    return Rulebook(None, None, None, _lib.lib().imf_rulebook_slots(n_out), n_out, 1)


# Sparse-conv arithmetic used by the model layers (include/imfnet_hip.h, imf_conv_args.variant; env IMF_CONV_VARIANT):
#   3 (default since round 5) = "bf16x3": every fp32 operand as three bf16 parts (exact), six bf16 MFMAs per 32 channels,
#       fp32 accumulation -- fp32 operand precision and range on the 16-bit matrix pipe, fp32 buffers, no range guard;
#   6 = split-f16 MFMA: two f16 parts per operand (22 bits), three MFMAs, operand images between layers, ~27 % faster,
#       activations must stay below 65504 (IMF_FLAG_RANGE -> the fragment is redone on variant 0): the FAST mode;
#   0 = fp32 MFMA (v_mfma_f32_16x16x4_f32): the reference's own arithmetic, ~1.6x slower than 3.
CONV_VARIANT = int(os.environ.get("IMF_CONV_VARIANT", "3"))
MAX_PIPELINED_KVOL = 27


def conv_variant_for(kvol):
    """Variants 6 and 3 need kvol <= 27 (their 16-bit weight images are only read by the LDS-DMA kernels)."""
    return CONV_VARIANT if (CONV_VARIANT not in (6, 3) or kvol <= MAX_PIPELINED_KVOL) else 0


def pack_weights(kernel, out=None, split16=False, variant=None):
    """ME kernel tensor [kvol,cin,cout] (or [cin,cout]) -> MFMA fragment-major image: fp32 B
    fragments (variants 0 and 1), with split16 / variant=6 the hi/lo f16 fragments of variant 6 (same size + trailer), with
    variant=3 the three bf16 parts of variant 3 (1.5 x the size)."""
    if variant is not None:
        split16 = variant == 6
    k = kernel.detach()
    if k.dim() == 2:
        k = k.unsqueeze(0)
    k = _req(k.contiguous().float(), torch.float32, "kernel", 3)
    kvol, cin, cout = k.shape
    L = _lib.lib()
    b3 = variant == 3
    need = (L.imf_packed_weight_floats_bf16x3 if b3 else L.imf_packed_weight_floats_split16 if split16
            else L.imf_packed_weight_floats)(kvol, cin, cout)
    packed = out if out is not None else torch.empty(need, dtype=torch.float32, device=k.device)
    if packed.numel() < need:
        raise ImfError(f"packed weight buffer has {packed.numel()} floats, needs {need}")
    fn = L.imf_pack_weights_bf16x3 if b3 else L.imf_pack_weights_split16 if split16 else L.imf_pack_weights
    check(fn(k.data_ptr(), kvol, cin, cout, packed.data_ptr(), _stream()), "imf_pack_weights")
    return packed


# variant 6 runs k_spconv_g (operands staged by LDS-DMA, csrc/spconv_g.hip) unless IMF_H3_GLDS=0 or the call asks
# for the register-staged k_spconv_h3 (`staging="regs"`, kernel_tag bit 1); the two agree bit for bit
H3_DMA = int(os.environ.get("IMF_H3_GLDS", "1")) != 0


def conv_kernel_name(variant, cin, cout, staging=None, kernel_tag=0):
    """Label of the kernel family a launch runs on (bench.py groups by it).  Variant 0 (fp32 MFMA) runs on the LDS-DMA
    kernels too since round 5 (AR = kArF32, csrc/spconv_g.hip / spconv_w.hip): same family names with an `/f32` suffix;
    kernel_tag bit 1 / staging="regs" selects round 1's register-staged k_spconv_mfma."""
    if kernel_tag & 16:
        return "k_pointwise_head_b3" if variant == 3 else "k_pointwise_head"
    suffix = {0: "/f32", 3: "/b3"}.get(variant, "")
    regs0 = variant == 0 and (kernel_tag & 2 or staging == "regs")
    if variant in (0, 3, 6) and not regs0 and (kernel_tag & 12 or staging in ("wave8", "wave4", "wave4h", "wave8u", "wave4u", "wave4o", "wave4h4", "wave8h4")):
        return f"k_spconv_w<{8 if (kernel_tag & 4 or staging == 'wave8') else 4}>" + suffix
    if variant == 3:
        return f"k_spconv_g<{4 if cout % 64 == 0 else 2}, 0>" + suffix
    if variant == 6:
        dma = H3_DMA if staging is None else staging == "dma"
        return f"k_spconv_{'g' if dma else 'h3'}<{4 if cout % 64 == 0 else 2}, 0>"
    if variant == 0 and not regs0:
        return f"k_spconv_g<{4 if cout % 64 == 0 else 2}, 0>" + suffix
    return f"k_spconv_mfma<{4 if cout % 64 == 0 else 2},{4 if cin % 64 == 0 else 2}>"


FMT_A_SPLIT, FMT_RES_SPLIT, FMT_OUT_SPLIT = 1, 2, 4      # imf_conv_args.operand_format (include/imfnet_hip.h)


def to_operand_image(x):
    """fp32 [n, c] (c % 32 == 0) -> the split-f16 operand image of the same shape and dtype (bytes reinterpreted): per row
    and 32-channel chunk [4 hi pieces | 4 lo pieces], piece j = the 8 halves of channels {4j..4j+3, 16+4j..16+4j+3},
    hi = f16(x), lo = f16(x - hi) (include/imfnet_hip.h, imf_conv_args.operand_format)."""
    n, c = x.shape
    v = x.view(n, c // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, c // 32, 32)      # [.., j, half, t]
    hi = v.to(torch.float16)
    lo = (v - hi.to(torch.float32)).to(torch.float16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).view(n, c)


def from_operand_image(img):
    """The values an operand image carries: hi + lo in fp32 (22 significant bits of the original)."""
    n, c = img.shape
    h = img.contiguous().view(torch.float16).view(n, c // 32, 2, 32).to(torch.float32)
    v = (h[:, :, 0] + h[:, :, 1]).view(n, c // 32, 4, 2, 4).permute(0, 1, 3, 2, 4)
    return v.reshape(n, c).contiguous()


def spconv(in_a, w_packed, cout, rb, in_b=None, scale=None, shift=None, residual=None,
           relu=False, l2norm=False, out=None, split_k=0, variant=0, fused_reduce=False, flags=None, staging=None,
           operand_format=0):
    """out[o] = epilogue(sum_k in[nbr[k][o]] @ W[k]) -- imf_spconv_fwd.  `operand_format` (variant 6, unsplit): FMT_A_SPLIT
    | FMT_RES_SPLIT | FMT_OUT_SPLIT -- which of in_a / in_b, residual, out are split-f16 operand images
    (`to_operand_image`) instead of fp32 rows.  `flags`: optional int32[1] device word that
    receives IMF_FLAG_RANGE (32) when an output is NaN or >= 65504 in magnitude.  `staging` (variant 6): None = the
    library default (LDS-DMA kernel k_spconv_g), "wave8" / "wave4" = the wave-split kernel for coarse levels
    (csrc/spconv_w.hip; kvol > 1, cout % 64 == 0, no split-K); "regs" = the register-staged k_spconv_h3, which exists in
    diagnostic builds of the library only (IMF_LIB=.../libimfnet_hip_h3.so; the product answers IMF_EUNSUPPORTED)."""
    _req(in_a, torch.float32, "in_a", 2)
    if in_b is not None:
        _req(in_b, torch.float32, "in_b", 2)
    if out is None:
        out = torch.empty((rb.n_out, cout), dtype=torch.float32, device=in_a.device)
    a = ConvArgs()
    a.in_a, a.in_b = in_a.data_ptr(), _ptr(in_b)
    a.c_a, a.c_b = in_a.shape[1], (0 if in_b is None else in_b.shape[1])
    a.w_packed, a.kvol, a.cout = w_packed.data_ptr(), rb.kvol, cout
    a.tile_rows, a.nbr, a.tile_mask = _ptr(rb.tile_rows), _ptr(rb.nbr), _ptr(rb.tile_mask)
    a.n_slots, a.n_out = rb.n_slots, rb.n_out
    a.scale, a.shift, a.residual = _ptr(scale), _ptr(shift), _ptr(residual)
    a.relu, a.l2norm = int(bool(relu)), int(bool(l2norm))
    a.out = out.data_ptr()
    L = _lib.lib()
    split = 1 if variant == 1 else (int(split_k) if split_k else L.imf_spconv_auto_split(rb.n_slots, cout, rb.max_active))
    if rb.kvol == 1 or staging in ("wave8", "wave4", "wave4h", "wave8u", "wave4u", "wave4o", "wave4h4", "wave8h4"):
        split = 1
    a.split_k, a.variant = split, int(variant)
    a.operand_format = int(operand_format)
    a.dyn_err = None if flags is None else flags.data_ptr()
    if staging not in (None, "dma", "regs", "wave8", "wave4", "wave4h", "wave8u", "wave4u", "wave4o", "wave4h4", "wave8h4"):
        raise ImfError(f"spconv: staging={staging!r}")
    a.kernel_tag = {"regs": 2, "wave8": 4, "wave4": 8, "wave4h": 8 | 64, "wave8u": 4 | 128, "wave4u": 8 | 128, "wave4o": 8 | 256, "wave4h4": 8 | 64 | 256, "wave8h4": 4 | 64 | 256}.get(staging, 0)
    ws = None
    nbytes = L.imf_spconv_workspace_bytes(rb.n_slots, cout, split)   # split-K partials / balanced-tail partials
    if nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=in_a.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    if split > 1:
        if fused_reduce and variant in (0, 6):
            tk = torch.zeros(rb.n_slots // TILE_ROWS * max(1, cout // 32), dtype=torch.int32, device=in_a.device)
            a.tickets = tk.data_ptr()
    big = max(in_a.numel(), 0 if in_b is None else in_b.numel()) * 4 >= 2 ** 31
    if variant in (6, 3) and big:
        raise ImfError("variants 6 / 3 address their inputs through a 2 GiB buffer window: use variant 0 for larger matrices")
    if variant == 0 and big:
        a.kernel_tag = 2                                  # the LDS-DMA kernels share that window: the register-staged kernel
    need = (L.imf_packed_weight_floats_split16 if variant == 6 else L.imf_packed_weight_floats_bf16x3 if variant == 3
            else L.imf_packed_weight_floats)(rb.kvol, a.c_a + a.c_b, cout)
    if w_packed.numel() != need:
        raise ImfError(f"packed weight has {w_packed.numel()} floats, expected {need} "
                       f"({rb.kvol}x{a.c_a + a.c_b}x{cout}, variant {variant})")
    ev = None
    if TRACE is not None:
        ev = _Ev()
        a.ev_begin, a.ev_end = ev.begin, ev.end
    check(L.imf_spconv_fwd(C.byref(a), _stream()), "imf_spconv_fwd")
    if ev is not None:
        cin = a.c_a + a.c_b
        TRACE.append(dict(kernel=conv_kernel_name(variant, cin, cout, "regs" if (staging == "regs" or fused_reduce) else staging),
                          kvol=rb.kvol, cin=cin, cout=cout, rb=rb, split=split, ev=ev))
    return out


def pointwise_head(in_a, in_b, w1_packed, w2_packed, scale1=None, shift1=None, relu1=True, scale2=None, shift2=None,
                   l2norm=True, out=None, flags=None, n_dev=None, a_split=False, variant=6):
    """imf_pointwise_head: conv1_tr + norm1_tr + ReLU + final + L2 normalisation (model/resunet.py:219-233) in one
    launch.  a_split: in_a / in_b are split-f16 operand images (`to_operand_image`).  in_a [n, c_a], in_b [n, c_b] or None; weight images from pack_weights_split16 (kvol 1); hidden width 64,
    output [n, 32].  Bit-identical to two `spconv(..., variant=6)` calls.  variant=3: bf16x3 weight images
    (`pack_weights(..., variant=3)`), fp32 rows, at most 96 input channels, no range flag -- bit-identical to two
    `spconv(..., variant=3)` calls."""
    _req(in_a, torch.float32, "in_a", 2)
    if in_b is not None:
        _req(in_b, torch.float32, "in_b", 2)
        if in_b.shape[0] != in_a.shape[0]:
            raise ImfError("pointwise_head: in_a / in_b row mismatch")
    n = in_a.shape[0]
    if out is None:
        out = torch.empty((n, 32), dtype=torch.float32, device=in_a.device)
    a = HeadArgs()
    a.in_a, a.in_b = in_a.data_ptr(), _ptr(in_b)
    a.c_a, a.c_b = in_a.shape[1], (0 if in_b is None else in_b.shape[1])
    L = _lib.lib()
    if variant not in (3, 6):
        raise ImfError(f"pointwise_head: variant={variant} (6 = split-f16 images, 3 = bf16x3 images)")
    floats = L.imf_packed_weight_floats_bf16x3 if variant == 3 else L.imf_packed_weight_floats_split16
    if w1_packed.numel() != floats(1, a.c_a + a.c_b, 64) or w2_packed.numel() != floats(1, 64, 32):
        raise ImfError("pointwise_head: packed weight images do not match [c_a + c_b, 64] / [64, 32] of this variant")
    a.w1_packed, a.scale1, a.shift1, a.relu1, a.c_mid = w1_packed.data_ptr(), _ptr(scale1), _ptr(shift1), int(bool(relu1)), 64
    a.w2_packed, a.scale2, a.shift2, a.l2norm, a.c_out = w2_packed.data_ptr(), _ptr(scale2), _ptr(shift2), int(bool(l2norm)), 32
    a.n, a.n_dev, a.out = n, _ptr(n_dev), out.data_ptr()
    a.a_split = int(bool(a_split))
    a.variant = int(variant)
    a.flags = None if flags is None else flags.data_ptr()
    check(L.imf_pointwise_head(C.byref(a), _stream()), "imf_pointwise_head")
    return out


def spconv_small_cin(feat, kernel, rb, scale=None, shift=None, relu=False):
    """First-layer conv for cin <= 4 (unpacked ME kernel [kvol,cin,cout])."""
    _req(feat, torch.float32, "feat", 2)
    k = _req(kernel.detach().contiguous(), torch.float32, "kernel", 3)
    kvol, cin, cout = k.shape
    if feat.shape[1] != cin or kvol != rb.kvol:
        raise ImfError("spconv_small_cin: feature / kernel / rulebook mismatch")
    out = torch.empty((rb.n_out, cout), dtype=torch.float32, device=feat.device)
    check(_lib.lib().imf_spconv_small_cin(feat.data_ptr(), cin, k.data_ptr(), kvol, cout,
                                          rb.nbr.data_ptr(), rb.n_slots, rb.n_out, _ptr(scale),
                                          _ptr(shift), int(bool(relu)), out.data_ptr(), _stream()),
          "imf_spconv_small_cin")
    return out


def conv_first_fused(level, feat, kernel, ksize, scale=None, shift=None, relu=False):
    """First-layer conv fused with its kernel map (imf_conv_first_fused).  feat None = all ones."""
    k = _req(kernel.detach().contiguous(), torch.float32, "kernel", 3)
    kvol, cin, cout = k.shape
    if kvol != ksize ** 3 or (feat is not None and feat.shape[1] != cin):
        raise ImfError("conv_first_fused: feature / kernel mismatch")
    if feat is not None:
        _req(feat, torch.float32, "feat", 2)
    out = torch.empty((level.n, cout), dtype=torch.float32, device=k.device)
    check(_lib.lib().imf_conv_first_fused(level.table.data_ptr(), level.capacity,
                                          level.coords_buf.data_ptr(), level.n, level.ts, ksize, _ptr(feat),
                                          cin, k.data_ptr(), cout, _ptr(scale), _ptr(shift), int(bool(relu)),
                                          out.data_ptr(), _stream()), "imf_conv_first_fused")
    return out


def conv_first_bitgrid(level, kernel, ksize, scale=None, shift=None, relu=False):
    """First-layer conv of the all-ones feature on an occupancy bit grid (imf_conv_first_bitgrid).
    Returns None when the level has no bounding box or the box is too large (use conv_first_fused)."""
    bbox = getattr(level, "bbox", None)
    if bbox is None:
        return None
    k = _req(kernel.detach().contiguous(), torch.float32, "kernel", 3)
    kvol, cin, cout = k.shape
    if cin != 1 or kvol != ksize ** 3:
        raise ImfError("conv_first_bitgrid: needs a [ksize^3, 1, cout] kernel")
    L = _lib.lib()
    box = (C.c_int32 * 8)(*bbox)
    words = L.imf_bitgrid_words(box, ksize)
    if words == 0:
        return None
    grid = torch.empty(words, dtype=torch.int32, device=k.device)
    out = torch.empty((level.n, cout), dtype=torch.float32, device=k.device)
    check(L.imf_conv_first_bitgrid(level.coords_buf.data_ptr(), level.n, box, ksize, grid.data_ptr(), words,
                                   k.data_ptr(), cout, _ptr(scale), _ptr(shift), int(bool(relu)),
                                   out.data_ptr(), _stream()), "imf_conv_first_bitgrid")
    return out


class FusionKernelWeights:
    """Packed weights of the bottleneck fusion block for imf_fusion_attention (built once per model)."""

    def __init__(self, attention_fusion, variant=None):
        """variant: the arithmetic of the owning model's convolutions (None: the process default) -- picks the 16-bit image of
        the two feed-forward matrices (bf16x3 for 3, split-f16 otherwise)."""
        from ._lib import FusionWeights
        variant = CONV_VARIANT if variant is None else int(variant)
        blk0, blk1 = attention_fusion.cross_attend_blocks
        att, ff = blk0.fn, blk1.fn.net
        self.dim, self.inner, self.hidden = att.to_q.in_features, att.to_q.out_features, ff[2].in_features
        self.scale = float(att.scale)
        self.supported = (att.heads == 1 and len(attention_fusion.layers) == 0 and self.dim == 256 and
                          self.inner == 128 and self.hidden == 1024 and att.to_q.weight.is_cuda)
        if not self.supported:
            return
        f = lambda t: t.detach().float().contiguous()                     # noqa: E731
        # feed-forward on the split-f16 convolution kernels (include/imfnet_hip.h, imf_fusion_weights): W1^T with its
        # columns interleaved per 64-column slab as [32 values | 32 gates] (GEGLU epilogue), b1 likewise
        H = self.hidden
        j = torch.arange(H // 32).view(-1, 1, 1) * 32
        c = torch.arange(32).view(1, 1, -1)
        perm = (j + c + torch.tensor([0, H]).view(1, 2, 1)).reshape(-1).to(ff[0].weight.device)   # packed col -> torch row
        w1t = f(ff[0].weight)[perm].t().contiguous()                       # [256, 2048], packed column order
        self.t = dict(ln1_g=f(blk0.norm.weight), ln1_b=f(blk0.norm.bias), wq_p=pack_weights(f(att.to_q.weight).t()),
                      wo_p=pack_weights(f(att.to_out.weight).t()), bo=f(att.to_out.bias), ln2_g=f(blk1.norm.weight),
                      # (w1_p / w2_p: the image of the 16-bit variant the process runs -- split-f16 for 6, three bf16 parts for 3)
                      ln2_b=f(blk1.norm.bias), w1_p=pack_weights(w1t, variant=3 if variant == 3 else 6),
                      b1=f(ff[0].bias)[perm].contiguous(),
                      w2_p=pack_weights(f(ff[2].weight).t(), variant=3 if variant == 3 else 6), b2=f(ff[2].bias),
                      # the same two matrices as fp32 images: the feed-forward of the variant-0 (fp32 MFMA) recompute
                      w1_f32=pack_weights(w1t), w2_f32=pack_weights(f(ff[2].weight).t()))
        self.c = FusionWeights(**{k: v.data_ptr() for k, v in self.t.items()})


def fusion_attention_batched(x, items, kt_packed, v_packed, n_tokens, tokens_padded, fw, out=None, flags=None, variant=None):
    """Rows [row0, row0+rows) of x for every (row0, rows) in `items` attend to image b's packed K^T / V
    (lists, one per item): imf_fusion_attention_batched_v.  flags: device int32[1] that receives IMF_FLAG_RANGE when a
    value feeding an f16 operand leaves the f16 range (None: not observed); variant: arithmetic of the feed-forward (6
    split-f16, 0 fp32 MFMA; default: the process-wide CONV_VARIANT, so that the fp32 recompute is fp32 throughout)."""
    _req(x, torch.float32, "x", 2)
    if out is None:
        out = torch.empty_like(x)
    if variant is None:
        variant = CONV_VARIANT if CONV_VARIANT in (6, 3) else 0
    B = len(items)
    L = _lib.lib()
    r0 = (C.c_int64 * B)(*[int(a) for a, _ in items])
    rn = (C.c_int64 * B)(*[int(b) for _, b in items])
    kp = (C.c_void_p * B)(*[t.data_ptr() for t in kt_packed])
    vp = (C.c_void_p * B)(*[t.data_ptr() for t in v_packed])
    nbytes = L.imf_fusion_workspace_bytes(x.shape[0])
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    check(L.imf_fusion_attention_batched_v(x.data_ptr(), B, r0, rn, kp, vp, int(n_tokens), int(tokens_padded),
                                           C.byref(fw.c), C.c_float(fw.scale), out.data_ptr(), ws.data_ptr(), nbytes,
                                           _ptr(flags), int(variant), _stream()), "imf_fusion_attention_batched")
    return out


def fusion_attention(x, kt_packed, v_packed, n_tokens, tokens_padded, fw, out=None, flags=None, variant=None):
    """x [n,256] -> [n,256]: the fusion block for one image (attention kernel + the two feed-forward GEMMs on the conv
    kernels); flags / variant as in `fusion_attention_batched`."""
    return fusion_attention_batched(x, [(0, x.shape[0])], [kt_packed], [v_packed], n_tokens, tokens_padded, fw, out=out,
                                    flags=flags, variant=variant)
