"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/imf_oracle_c.c, presenting the same
functions as oracle/imf_oracle.py's numpy geometry (used as its cross-check and as the geometry
half of bench.py's cpu_baseline)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libimf_oracle_c.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            import subprocess
            subprocess.run(["make", "-C", _HERE], check=True)
        L = C.CDLL(_SO)
        P, I, L64, D = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.orc_voxelize.restype, L.orc_voxelize.argtypes = L64, [P, L64, D, I, P, P]
        L.orc_downsample.restype, L.orc_downsample.argtypes = L64, [P, L64, I, P, P]
        L.orc_rulebook.restype, L.orc_rulebook.argtypes = I, [P, L64, P, L64, I, I, I, P]
        L.imf_cpu_spconv_fwd.restype, L.imf_cpu_spconv_fwd.argtypes = None, [P, I, P, I, I, P, L64, P]
        L.orc_set_threads.restype, L.orc_set_threads.argtypes = None, [I]
        _lib = L
    return _lib


def voxelize(xyz, voxel_size, batch_index=0):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    n = len(xyz)
    coords = np.empty((n, 4), np.int32)
    inds = np.empty(n, np.int64)
    m = lib().orc_voxelize(xyz.ctypes.data, n, float(voxel_size), batch_index, coords.ctypes.data,
                           inds.ctypes.data)
    if m < 0:
        raise RuntimeError(f"orc_voxelize rc={m}")
    return coords[:m].copy(), inds[:m].copy()


def downsample(coords, out_stride):
    c = np.ascontiguousarray(coords, dtype=np.int32)
    out = np.empty_like(c)
    parent = np.empty(len(c), np.int32)
    m = lib().orc_downsample(c.ctypes.data, len(c), out_stride, out.ctypes.data, parent.ctypes.data)
    return out[:m].copy(), parent


def _rb(in_c, out_c, ts, ksize, sign):
    a = np.ascontiguousarray(in_c, dtype=np.int32)
    b = np.ascontiguousarray(out_c, dtype=np.int32)
    nbr = np.empty((len(b), ksize ** 3), np.int32)
    rc = lib().orc_rulebook(a.ctypes.data, len(a), b.ctypes.data, len(b), ts, ksize, sign, nbr.ctypes.data)
    assert rc == 0
    return nbr


def rulebook(in_coords, out_coords, ts_in, ksize):
    return _rb(in_coords, out_coords, ts_in, ksize, +1)


def rulebook_transpose(coarse_coords, fine_coords, ts_fine, ksize):
    return _rb(coarse_coords, fine_coords, ts_fine, ksize, -1)


class Geometry:
    """Same products as imf_oracle.Geometry, built by the C restatement."""

    def __init__(self, coords, conv1_kernel_size=5):
        self.levels = [np.asarray(coords, np.int32)]
        self.parents = []
        for lv in range(3):
            c, p = downsample(self.levels[-1], 2 << lv)
            self.levels.append(c)
            self.parents.append(p)
        L = self.levels
        self.k_first = rulebook(L[0], L[0], 1, conv1_kernel_size)
        self.k3 = [rulebook(L[i], L[i], 1 << i, 3) for i in range(4)]
        self.down = [rulebook(L[i], L[i + 1], 1 << i, 3) for i in range(3)]
        self.up = [rulebook_transpose(L[i + 1], L[i], 1 << i, 3) for i in range(3)]


def spconv(feat, kernel, nbr):
    """The C / OpenMP twin of the sparse convolution (imf_cpu_spconv_fwd): same contract as imf_oracle.spconv."""
    import torch
    f = np.ascontiguousarray(torch.as_tensor(feat, dtype=torch.float32).numpy())
    w = np.ascontiguousarray(torch.as_tensor(kernel, dtype=torch.float32).numpy())
    if w.ndim == 2:
        w = w[None]
    kvol, cin, cout = w.shape
    assert cout <= 512 and f.shape[1] == cin
    if nbr is None:
        n_out, nb = f.shape[0], None
    else:
        nb = np.ascontiguousarray(np.asarray(nbr), dtype=np.int32)
        n_out = nb.shape[0]
        assert nb.shape[1] == kvol
    out = np.empty((n_out, cout), np.float32)
    lib().imf_cpu_spconv_fwd(f.ctypes.data, cin, w.ctypes.data, kvol, cout, None if nb is None else nb.ctypes.data, n_out,
                             out.ctypes.data)
    return torch.from_numpy(out)


def set_threads(n):
    """OpenMP threads of the C restatement (geometry + convolution twin)."""
    lib().orc_set_threads(int(n))
