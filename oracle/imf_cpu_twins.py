"""TEST INFRASTRUCTURE ONLY -- numpy face of oracle/imf_cpu_twins.c: the `imf_cpu_*` twins of the C ABI, same
signatures as include/imfnet_hip.h on host pointers.  Every wrapper returns arrays in the GPU library's own layouts, so the
GPU tests compare device results with these directly.  Never imported by imfnet_amd/."""
import ctypes as C
import os

import numpy as np

import imf_oracle_cbind as OC

TILE = 64


def _lib():
    L = OC.lib()
    if not getattr(L, "_twins_bound", False):
        P, I, L64, D = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.imf_cpu_voxelize.restype, L.imf_cpu_voxelize.argtypes = I, [P, I, L64, D, I, P, P, P, P, L64, P, P, P]
        L.imf_cpu_downsample.restype, L.imf_cpu_downsample.argtypes = I, [P, P, L64, I, P, P, P, L64, P, P]
        L.imf_cpu_rulebook_conv.restype, L.imf_cpu_rulebook_conv.argtypes = I, [P, L64, P, L64, I, I, P, P, P, P]
        L.imf_cpu_rulebook_transpose.restype = I
        L.imf_cpu_rulebook_transpose.argtypes = [P, L64, P, L64, I, I, P, P, P, L64, P, P]
        L.imf_cpu_spconv_fwd_abi.restype, L.imf_cpu_spconv_fwd_abi.argtypes = I, [P, P]
        L.imf_cpu_rulebook_sorted_workspace_bytes.restype, L.imf_cpu_rulebook_sorted_workspace_bytes.argtypes = C.c_size_t, [L64]
        L.imf_cpu_rulebook_sort_by_occupancy.restype = I
        L.imf_cpu_rulebook_sort_by_occupancy.argtypes = [P, I, L64, L64, P, P, P, P, P, C.c_size_t, P]
        L._twins_bound = True
    return L


def _capacity(n):
    c = 1024
    while c < 2 * n:
        c *= 2
    return c


class Level:
    """coords int32 [m, 4] + the host hash table (imf_slot[capacity]: u64 key, i32 row, i32 pad) the next call probes."""

    def __init__(self, coords, table, first_idx=None):
        self.coords, self.table, self.first_idx, self.n = coords, table, first_idx, len(coords)
        self.capacity = len(table)


_SLOT = np.dtype([("key", "<u8"), ("val", "<i4"), ("pad", "<i4")])


def voxelize(xyz, voxel_size, batch_index=0):
    xyz = np.ascontiguousarray(xyz)
    assert xyz.dtype in (np.float64, np.float32)
    n = len(xyz)
    coords, first = np.empty((n, 4), np.int32), np.empty(n, np.int32)
    m, err = np.zeros(2, np.int32), np.zeros(1, np.int32)
    table = np.empty(_capacity(n), _SLOT)
    rc = _lib().imf_cpu_voxelize(xyz.ctypes.data, int(xyz.dtype == np.float64), n, float(voxel_size), batch_index, coords.ctypes.data,
                                 first.ctypes.data, m.ctypes.data, table.ctypes.data, len(table), None, err.ctypes.data, None)
    assert rc == 0
    return Level(coords[:m[0]].copy(), table, first[:m[0]].copy()), int(err[0])


def downsample(level, out_stride):
    out, m = np.empty_like(level.coords), np.zeros(2, np.int32)
    table = np.empty(_capacity(level.n), _SLOT)
    rc = _lib().imf_cpu_downsample(level.coords.ctypes.data, None, level.n, out_stride, out.ctypes.data, m.ctypes.data,
                                   table.ctypes.data, len(table), None, None)
    assert rc == 0
    return Level(out[:m[0]].copy(), table)


def rulebook_conv(in_level, out_level, ts_in, ksize):
    """(tile_rows [n_slots], nbr [kvol, n_slots], tile_mask [tiles, 4] uint32) as imf_rulebook_conv lays them out."""
    n_slots, kvol = (out_level.n + TILE - 1) // TILE * TILE, ksize ** 3
    rows, nbr = np.empty(n_slots, np.int32), np.empty((kvol, n_slots), np.int32)
    mask = np.empty((n_slots // TILE, 4), np.uint32)
    rc = _lib().imf_cpu_rulebook_conv(in_level.table.ctypes.data, in_level.capacity, out_level.coords.ctypes.data, out_level.n,
                                      ts_in, ksize, rows.ctypes.data, nbr.ctypes.data, mask.ctypes.data, None)
    assert rc == 0
    return rows, nbr, mask


def rulebook_sort_by_occupancy(nbr, n_out):
    """imf_rulebook_sort_by_occupancy on a host map in identity slot order (nbr [kvol, n_slots]): the permuted (tile_rows, nbr,
    tile_mask)."""
    kvol, n_slots = nbr.shape
    nbr = np.ascontiguousarray(nbr, np.int32)
    rows, out = np.empty(n_slots, np.int32), np.empty((kvol, n_slots), np.int32)
    mask = np.empty((n_slots // TILE, 4), np.uint32)
    L = _lib()
    ws = np.empty(L.imf_cpu_rulebook_sorted_workspace_bytes(n_slots), np.uint8)
    rc = L.imf_cpu_rulebook_sort_by_occupancy(nbr.ctypes.data, kvol, n_slots, n_out, None, rows.ctypes.data, out.ctypes.data,
                                              mask.ctypes.data, ws.ctypes.data, ws.size, None)
    assert rc == 0
    return rows, out, mask


def rulebook_transpose(coarse_level, fine_level, ts_fine, ksize=3):
    n_slots, kvol = ((fine_level.n + TILE - 1) // TILE + 8) * TILE, ksize ** 3
    rows, nbr = np.empty(n_slots, np.int32), np.empty((kvol, n_slots), np.int32)
    mask, counters = np.empty((n_slots // TILE, 4), np.uint32), np.zeros(16, np.int32)
    rc = _lib().imf_cpu_rulebook_transpose(coarse_level.table.ctypes.data, coarse_level.capacity, fine_level.coords.ctypes.data,
                                           fine_level.n, ts_fine, ksize, rows.ctypes.data, nbr.ctypes.data, mask.ctypes.data,
                                           n_slots, counters.ctypes.data, None)
    assert rc == 0
    return rows, nbr, mask


def spconv_fwd(conv_args):
    """conv_args: a filled imfnet_amd._lib.ConvArgs (the ctypes mirror of imf_conv_args) whose pointers are HOST pointers."""
    rc = _lib().imf_cpu_spconv_fwd_abi(C.byref(conv_args), None)
    assert rc == 0, rc
