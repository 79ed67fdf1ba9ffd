"""Host-side codecs of the batch path: PLY points in, image in, NPZ out.

The reference uses Open3D (`o3d.io.read_point_cloud`, scripts/generate_desc.py:83), matplotlib
(`image.imread`, :92), OpenCV (`cv2.resize`, util/uio.py:33-40) and `np.savez_compressed`
(:118-123).  None of the first three is needed: the fixture PLYs are plain binary-little-endian
vertex lists, PIL decodes the images, and INTER_LINEAR resize is the half-pixel bilinear formula.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def _native():
    from . import _lib
    return _lib.lib()


def read_ply_points(path, out=None):
    """Vertex x,y,z of a PLY file as float64 [N,3] (what np.array(pcd.points) is in the reference:
    Open3D widens float32 vertices to float64).  Native reader (imf_ply_read_points, csrc/codecs.hip); `out`:
    optional float64 [cap,3] buffer (e.g. pinned host memory) to read into -- a view of its first N rows is returned."""
    import ctypes as C
    L = _native()
    p = os.fsencode(path)
    n = L.imf_ply_vertex_count(p)
    if n < 0:
        raise ValueError(f"{path}: {L.imf_last_error().decode()}")
    buf = out if out is not None and out.shape[0] >= n else np.empty((n, 3), dtype=np.float64)
    got = L.imf_ply_read_points(p, buf.ctypes.data_as(C.c_void_p), buf.shape[0])
    if got < 0:
        raise ValueError(f"{path}: {L.imf_last_error().decode()}")
    return buf[:got]


def read_ply_points_numpy(path):
    """The same in numpy (kept as the independent check of the native reader)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(a in names for a in "xyz"):
            raise ValueError(f"{path}: vertex element has no x/y/z")
        if fmt == "ascii":
            a = np.loadtxt(f, max_rows=n, ndmin=2)
            return np.stack([a[:, names.index(c)] for c in "xyz"], 1).astype(np.float64)
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, end + t) for nm, t in props])
        a = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return np.stack([a["x"], a["y"], a["z"]], 1).astype(np.float64)


def read_image(path):
    """matplotlib.image.imread semantics (generate_desc.py:92): PNG -> float32 in [0,1] (HWC, alpha
    dropped is NOT done by the reference; fixtures are RGB); any other format -> uint8 0..255, which the
    reference then feeds to the network un-normalised (SURVEY App. D.5 -- kept for parity)."""
    if os.path.splitext(path)[1].lower() == ".png":
        import ctypes as C
        L = _native()
        p = os.fsencode(path)
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        if L.imf_png_info(p, C.byref(h), C.byref(w), C.byref(c)) == 0:
            out = np.empty((h.value, w.value, c.value), dtype=np.float32)
            rc = L.imf_png_read_f32(p, out.ctypes.data_as(C.c_void_p), out.size, C.byref(h), C.byref(w), C.byref(c))
            if rc == 0:
                return out[:, :, 0] if c.value == 1 else out          # matplotlib returns [H,W] for grey images
    elif os.path.splitext(path)[1].lower() in (".jpg", ".jpeg"):
        # the native decoder (csrc/jpeg.hip): baseline YCbCr files, bit-identical to PIL; anything else -> PIL below
        import ctypes as C
        L = _native()
        p = os.fsencode(path)
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        if L.imf_jpeg_info(p, C.byref(h), C.byref(w), C.byref(c)) == 0:
            out = np.empty((h.value, w.value, 3), dtype=np.uint8)
            if L.imf_jpeg_read_u8(p, out.ctypes.data_as(C.c_void_p), out.size, C.byref(h), C.byref(w), C.byref(c)) == 0:
                return out
    return read_image_pil(path)


def read_image_pil(path):
    """Generic decoder (every format PIL knows): the fallback of read_image and the check of the native PNG path."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "P":
            im = im.convert("RGB")
        arr = np.asarray(im)
    if os.path.splitext(path)[1].lower() == ".png":
        return np.divide(arr, 255 if arr.dtype == np.uint8 else 65535, dtype=np.float32)
    return arr


def process_image(image, aim_H=480, aim_W=640, mode="resize", clip_mode="center"):
    """util/uio.py:18-99.  `resize` (the only branch on the path, generate_desc.py:94-95): cv2.resize(image, (aim_W, aim_H),
    INTER_LINEAR) == bilinear with half-pixel centres and no anti-aliasing; returns float32 HWC.  Already-sized images are
    returned unchanged.  `clip` and `padding`: _process_clip / _process_padding below (host-side numpy, off the path)."""
    img = np.asarray(image)
    H, W, _ = img.shape
    if H == aim_H and W == aim_W:
        return img
    if mode == "clip":
        return _process_clip(img, aim_H, aim_W, clip_mode)
    if mode == "padding":
        return _process_padding(img, aim_H, aim_W)
    if mode != "resize":
        return img                                    # (the reference falls through its if / elif chain: unchanged)
    if img.dtype == np.uint8:
        # cv2.resize on 8-bit input (the .jpg branch of generate_desc.py:88-95) interpolates in fixed point and returns
        # uint8: 11-bit coefficients, result rounded to the nearest integer (OpenCV resize.cpp, INTER_RESIZE_COEF_BITS = 11;
        # [RECALLED], within 1 LSB of every OpenCV code path).  The reference then casts to float32 0..255.
        return _resize_linear_u8(img, aim_H, aim_W)
    import ctypes as C
    src = np.ascontiguousarray(img, dtype=np.float32)
    out = np.empty((aim_H, aim_W, src.shape[2]), dtype=np.float32)
    rc = _native().imf_resize_bilinear_f32(src.ctypes.data_as(C.c_void_p), H, W, src.shape[2],
                                           out.ctypes.data_as(C.c_void_p), aim_H, aim_W, 0)
    if rc != 0:
        raise ValueError(_native().imf_last_error().decode())
    return out


# ---- the modes that are NOT on the descriptor-generation path (util/uio.py:41-99), restated for completeness ----------
def _reflect101(i, n):
    """cv2.BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba) for indices one or two beyond the edge."""
    if n == 1:
        return np.zeros_like(i)
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def _pyr_down(img):
    """cv2.pyrDown: 5x5 Gaussian ([1 4 6 4 1] / 16 per axis, BORDER_REFLECT_101), then every second row / column ->
    ((H + 1) // 2, (W + 1) // 2).  8-bit images: integer sums, (s + 128) >> 8.  [RECALLED from the OpenCV documentation and
    pyramids.cpp; OpenCV is absent here, so this branch is UNPINNED -- it is not on the generate_desc path.]"""
    H, W, _ = img.shape
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    src = img.astype(np.int64) if img.dtype == np.uint8 else img.astype(np.float64)
    oy, ox = np.arange((H + 1) // 2) * 2, np.arange((W + 1) // 2) * 2
    rows = sum(k[t] * src[:, _reflect101(ox + t - 2, W)] for t in range(5))
    out = sum(k[t] * rows[_reflect101(oy + t - 2, H)] for t in range(5))
    if img.dtype == np.uint8:
        return np.clip((out + 128) >> 8, 0, 255).astype(np.uint8)
    return (out / 256.0).astype(img.dtype)


def _pyr_up(img):
    """cv2.pyrUp: zero-interleave to (2H, 2W), the same Gaussian times 4.  Per axis: even outputs (s[i-1] + 6 s[i] + s[i+1]) / 8,
    odd outputs (s[i] + s[i+1]) / 2, borders reflected (101) on the left and replicated on the right; 8-bit: (s + 32) >> 6.
    [RECALLED, unpinned: see _pyr_down.]"""
    H, W, _ = img.shape
    src = img.astype(np.int64) if img.dtype == np.uint8 else img.astype(np.float64)

    def up(a, n, axis):
        i = np.arange(n)
        prev, nxt = _reflect101(i - 1, n), np.minimum(i + 1, n - 1)
        a_p, a_n = np.take(a, prev, axis), np.take(a, nxt, axis)
        even, odd = a_p + 6 * a + a_n, 4 * (a + a_n)
        shape = list(a.shape)
        shape[axis] = 2 * n
        out = np.empty(shape, dtype=a.dtype)
        sl_e, sl_o = [slice(None)] * a.ndim, [slice(None)] * a.ndim
        sl_e[axis], sl_o[axis] = slice(0, None, 2), slice(1, None, 2)
        out[tuple(sl_e)], out[tuple(sl_o)] = even, odd
        return out

    out = up(up(src, W, 1), H, 0)
    if img.dtype == np.uint8:
        return np.clip((out + 32) >> 6, 0, 255).astype(np.uint8)
    return (out / 64.0).astype(img.dtype)


def _process_clip(img, aim_H, aim_W, clip_mode):
    """util/uio.py:41-62: double the image until it covers the target, halve it once when it is more than twice as large in
    both directions, then cut an aim_H x aim_W window (centre / top-left / uniformly random offset from numpy's global RNG)."""
    H, W, _ = img.shape
    while H < aim_H or W < aim_W:
        img = _pyr_up(img)
        H, W, _ = img.shape
    if H > aim_H * 2 and W > aim_W * 2:
        img = _pyr_down(img)
        H, W, _ = img.shape
    if clip_mode == "center":
        top, left = int((H - aim_H) / 2), int((W - aim_W) / 2)
        return img[top:top + aim_H, left:left + aim_W]
    if clip_mode == "normal":
        return img[0:aim_H, 0:aim_W]
    if clip_mode == "random":
        top = int(np.random.random() * (H - aim_H))          # (two draws, rows first: the reference's order)
        left = int(np.random.random() * (W - aim_W))
        return img[top:top + aim_H, left:left + aim_W]
    return img                                               # unknown clip_mode: the reference returns the pyramid image


def _process_padding(img, aim_H, aim_W):
    """util/uio.py:64-97 AS WRITTEN, quirks included: its four cases compare the wrong way round, so
      * an image larger than the target in both directions asks numpy for a block of negative height and raises ValueError;
      * larger in one direction only: that direction is cut, the other one gets a zero block of (target - size) rows or
        columns -- ValueError again unless the image is smaller there;
      * not larger in either direction: returned as it is (the slice is a no-op), float64 is NOT forced.
    Zero blocks are float64, so a padded result is float64 (np.concatenate promotes)."""
    H, W, C = img.shape
    chw = np.transpose(img, (2, 0, 1))
    if aim_H < H and aim_W < W:
        chw = np.concatenate([chw, np.zeros((C, aim_H - H, W))], axis=1)     # raises: negative dimension
        chw = np.concatenate([chw, np.zeros((C, aim_H, aim_W - W))], axis=2)
    elif aim_H < H:
        chw = chw[:, 0:aim_H, :]
        chw = np.concatenate([chw, np.zeros((C, aim_H, aim_W - W))], axis=2)
    elif aim_W < W:
        chw = chw[:, :, 0:aim_W]
        chw = np.concatenate([chw, np.zeros((C, aim_H - H, W))], axis=1)
    else:
        chw = chw[:, 0:aim_H, 0:aim_W]
    return np.transpose(chw, (1, 2, 0))


def process_image_torch(image, aim_H, aim_W):
    """The same resize through torch (the independent check of the native one)."""
    t = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).permute(2, 0, 1)[None]
    out = F.interpolate(t, size=(aim_H, aim_W), mode="bilinear", align_corners=False)
    return out[0].permute(1, 2, 0).contiguous().numpy()


def _linear_taps(n_in, n_out, bits=11):
    """Source index pairs and fixed-point weights of cv2's INTER_LINEAR along one axis (half-pixel centres, edge clamp)."""
    scale = n_in / n_out
    f = (np.arange(n_out) + 0.5) * scale - 0.5
    i0 = np.floor(f).astype(np.int64)
    frac = f - i0
    frac = np.where(i0 < 0, 0.0, frac)
    i0 = np.clip(i0, 0, n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    frac = np.where(i0 >= n_in - 1, 0.0, frac)
    w1 = np.rint(frac * (1 << bits)).astype(np.int64)
    return i0, i1, (1 << bits) - w1, w1


def _resize_linear_u8(img, aim_H, aim_W):
    H, W, _ = img.shape
    y0, y1, b0, b1 = _linear_taps(H, aim_H)
    x0, x1, a0, a1 = _linear_taps(W, aim_W)
    src = img.astype(np.int64)
    rows0 = src[y0][:, x0] * a0[None, :, None] + src[y0][:, x1] * a1[None, :, None]     # horizontal pass, scale 2^11
    rows1 = src[y1][:, x0] * a0[None, :, None] + src[y1][:, x1] * a1[None, :, None]
    out = (rows0 * b0[:, None, None] + rows1 * b1[:, None, None] + (1 << 21)) >> 22        # vertical pass, round
    return np.clip(out, 0, 255).astype(np.uint8)


def image_to_nchw(image):
    """generate_desc.py:96-97: HWC -> [1,C,H,W]."""
    return np.expand_dims(np.transpose(image, (2, 0, 1)), 0)


NPZ_LEVEL = int(os.environ.get("IMFNET_NPZ_LEVEL", "1"))
NPZ_THREADS = int(os.environ.get("IMFNET_NPZ_THREADS", "0"))       # 0: chosen by the caller (generate_desc sizes it)


def save_npz(path, level=None, threads=None, **arrays):
    """np.savez_compressed(path, **arrays) through the native ZIP writer (imf_npz_write_mt): the same members and arrays,
    deflated at `level` (0 = stored like np.savez; 2..9 = zlib at that level, numpy uses 6; default 1 = the library's own
    deflate producers, csrc/fast_deflate.h: value-granular matches for float64 point arrays, byte Huffman for float32
    descriptors -- ~10x numpy's speed per file at a file no larger than zlib level 1's), the deflate cut into independent
    256 KiB blocks over `threads` host threads (default NPZ_THREADS, else 1)."""
    import ctypes as C
    if not str(path).endswith(".npz"):
        path = str(path) + ".npz"
    names = list(arrays)
    arrs = [np.require(arrays[k], requirements='C') for k in names]      # (ascontiguousarray would make a 0-d array 1-d)
    for a in arrs:
        if a.dtype.byteorder == ">" or a.dtype.hasobject or a.dtype.fields is not None or a.dtype.kind not in "fiub":
            return np.savez_compressed(path, **arrays)        # strings / exotic dtypes: numpy's writer
    n = len(arrs)
    c_names = (C.c_char_p * n)(*[k.encode() for k in names])
    c_dtype = (C.c_char_p * n)(*[a.dtype.str.replace("=", "<").replace("|", "|").encode() for a in arrs])
    c_ndim = (C.c_int32 * n)(*[a.ndim for a in arrs])
    dims = [d for a in arrs for d in a.shape]
    c_shape = (C.c_int64 * max(1, len(dims)))(*dims)
    c_data = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    L = _native()
    rc = L.imf_npz_write_mt(os.fsencode(str(path)), n, c_names, c_dtype, c_ndim, c_shape, c_data,
                            NPZ_LEVEL if level is None else int(level), max(1, int(threads if threads is not None else NPZ_THREADS)))
    if rc != 0:
        raise OSError(L.imf_last_error().decode())


def save_descriptors(path, points, xyz_down, feature):
    """generate_desc.py:118-123 -- keys and dtypes consumed by scripts/evaluation_3dmatch.py:129-132."""
    if torch.is_tensor(feature):
        feature = feature.detach().cpu().numpy()
    save_npz(path, points=np.asarray(points), xyz=np.asarray(xyz_down), feature=feature)
