"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/imfnet_hip.h
declares; the host-side mirror of the reference interface (model registry, state_dict schema,
codecs, file ordering, checkpoint reader); the oracle's C restatement vs its numpy twin."""
import json
import os
import re
import struct

import numpy as np
import pytest
import torch

import imf_oracle as O
from conftest import GOLDEN, ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "imfnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(imf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from imfnet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(declared)
    # pure host-side entry points work without a GPU
    assert L.imf_version() >= 100
    assert L.imf_hash_capacity(1000) == 2048 and L.imf_hash_capacity(1) == 1024
    assert L.imf_rulebook_slots(65) == 128
    assert L.imf_rulebook_transpose_slots(65) == (2 + 8) * 64
    assert L.imf_packed_weight_floats(27, 64, 64) == 27 * 64 * 64
    assert L.imf_packed_weight_floats_split16(27, 64, 64) == 27 * 64 * 64 + 64
    assert L.imf_image_tokens(120, 160) == 300 and L.imf_image_tokens(64, 96) == 96
    assert L.imf_image_workspace_bytes(1, 120, 160) > 0 and L.imf_image_workspace_bytes(0, 120, 160) == 0
    assert L.imf_spconv_auto_split(51264, 64, 27) == 1          # 801 tiles: no split
    assert L.imf_spconv_auto_split(1088, 256, 27) == 8          # 17 tiles x 4 slabs: split 8
    assert L.imf_spconv_auto_split(1088, 64, 1) == 1
    assert L.imf_spconv_workspace_bytes(1088, 256, 8) == 8 * 1088 * 256 * 4
    assert L.imf_spconv_workspace_bytes(1088, 256, 1) == 0


def test_kernel_policy_is_static_per_layer_and_batch():
    """imf_resunet_conv_kernel_tag (include/imfnet_hip.h): the executors' kernel choice is a function of (level, kvol, cin,
    cout, variant, n_items) ONLY -- never of a row count -- and encodes the measured rules: bit 2 / 3 = 8 / 4 wavefronts, bit 6
    = half tiles, bit 7 = 48-row units; pointwise layers and cout % 64 != 0 go to k_spconv_g (0)."""
    from imfnet_amd import _lib
    tag = _lib.lib().imf_resunet_conv_kernel_tag
    for v in (0, 3, 6):
        assert tag(0, 1, 96, 64, v, 2) == 0 and tag(2, 27, 64, 32, v, 2) == 0           # pointwise / 32 output channels
        assert tag(1, 27, 64, 64, v, 2) == (8 | 256 if v == 3 else 8) and tag(2, 27, 128, 128, v, 2) == 4   # a pair: level 1 on 4 (bf16x3: the build for three per SIMD), level 2 on 8 wavefronts
        assert tag(3, 27, 256, 256, v, 2) == (4 | 128) == tag(3, 27, 128, 256, v, 8)    # 48-row units on level 3 from two fragments on
        assert tag(2, 27, 128, 128, v, 3) == 8 == tag(2, 27, 128, 128, v, 8)            # level 2 on 4 wavefronts from three on
        assert tag(0, 27, 32, 32, v, 2) == 0 and tag(0, 27, 128, 64, v, 2) == ((8 | 64 | 256) if v == 3 else 0)   # stride 1: k_spconv_g (bf16x3's conv2_tr: half tiles, round 6)
    assert tag(0, 27, 64, 64, 3, 2) == (8 | 128) and tag(0, 27, 64, 64, 6, 2) == 0      # bf16x3: block1_tr on 48-row units of 4 wavefronts (round 6)
    assert tag(1, 27, 256, 64, 3, 2) == (8 | 64 | 256) == tag(2, 27, 256, 128, 3, 2)    # bf16x3: the up-convolutions on half tiles (bit 8: the four-wavefronts-per-SIMD build)
    for level in (1, 2, 3):                                                             # one fragment per forward: half tiles
        assert tag(level, 27, 128, 128, 3, 1) == ((8 | 64 | 256) if level == 1 else (4 | 64 | 256))     # (levels 2, 3: half tiles of 8 wavefronts)
        assert tag(level, 27, 128, 128, 6, 1) == (8 if level == 1 else 4) == tag(level, 27, 128, 128, 0, 1)
    assert tag(2, 27, 64, 64, 1, 2) == 0                                                 # other variants: no wave-split kernel


def test_conv_args_struct_matches_header_layout():
    """ctypes mirror of struct imf_conv_args: field order / count tracks the header."""
    from imfnet_amd._lib import ConvArgs
    text = open(os.path.join(ROOT, "include", "imfnet_hip.h")).read()
    body = text[text.index("typedef struct imf_conv_args {"):text.index("} imf_conv_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip(" *") for n in re.sub(r"^(const\s+)?\w+\s", "", decl).split(",")]
    assert names == [f[0] for f in ConvArgs._fields_]


def _struct_fields(name):
    text = open(os.path.join(ROOT, "include", "imfnet_hip.h")).read()
    body = text[text.index("typedef struct %s {" % name):text.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip(" *") for n in re.sub(r"^(const\s+)?\w+\s", "", decl).split(",")]
    return names


def test_pipeline_job_and_fusion_weight_structs_match_the_header():
    """ctypes mirrors of struct imf_job (the streaming pipeline) and imf_fusion_weights (fp32 feed-forward images added in
    round 4): field order / count tracks the header."""
    from imfnet_amd._lib import FusionWeights, Job
    assert _struct_fields("imf_job") == [f[0] for f in Job._fields_]
    assert _struct_fields("imf_fusion_weights") == [f[0] for f in FusionWeights._fields_]


def test_host_narrow_points_is_exact_or_refuses():
    """imf_host_narrow_points (host code, no GPU): float64 values that are float32 values narrow bit-exactly; one value
    that is not makes the call refuse (return 0); NaNs travel; odd lengths (the tail loop) too."""
    from imfnet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 100003):
        a = rng.normal(size=n).astype(np.float32).astype(np.float64)
        d = np.full(n, 7.0, np.float32)
        assert L.imf_host_narrow_points(a.ctypes.data, n, d.ctypes.data) == 1
        assert (d.astype(np.float64) == a).all()
        if n:
            b = a.copy()
            b[n // 2] += 1e-12 if b[n // 2] != 0 else 1e-300
            assert L.imf_host_narrow_points(b.ctypes.data, n, d.ctypes.data) == 0
            c = a.copy()
            c[n - 1] = np.nan
            assert L.imf_host_narrow_points(c.ctypes.data, n, d.ctypes.data) == 1 and np.isnan(d[n - 1])
    big = np.array([1e300, -1e300, 3.5], np.float64)              # beyond float32: inf after narrowing, not equal
    assert L.imf_host_narrow_points(big.ctypes.data, 3, np.empty(3, np.float32).ctypes.data) == 0


def test_package_import_prepares_the_runtime_for_async_copies():
    """Importing imfnet_amd before the HIP runtime starts sets ROC_CPU_WAIT_FOR_SIGNAL=0 (hipMemcpyAsync behind queued
    kernels must not block the pipeline's worker); a value the user set is left alone and decides SDMA_ASYNC."""
    import subprocess
    import sys
    code = "import os, imfnet_amd; print(os.environ.get('ROC_CPU_WAIT_FOR_SIGNAL'), imfnet_amd.SDMA_ASYNC)"
    env = {k: v for k, v in os.environ.items() if k != "ROC_CPU_WAIT_FOR_SIGNAL"}
    env["PYTHONPATH"] = ROOT
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).split() == ["0", "True"]
    env["ROC_CPU_WAIT_FOR_SIGNAL"] = "1"
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).split() == ["1", "False"]
    # the opt-out: IMFNET_LEAVE_ENV=1 -- the import does not touch the environment; copy engines only if the host set the mode
    del env["ROC_CPU_WAIT_FOR_SIGNAL"]
    env["IMFNET_LEAVE_ENV"] = "1"
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).split() == ["None", "False"]
    env["ROC_CPU_WAIT_FOR_SIGNAL"] = "0"
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).split() == ["0", "True"]


def test_errors_are_loud_without_gpu():
    from imfnet_amd import ImfError, ops
    with pytest.raises(ImfError):
        ops.voxelize(torch.zeros(8, 3, dtype=torch.float64), 0.025)       # CPU tensor: no fallback
    import imfnet_amd.sparse as ME
    with pytest.raises(ImfError):
        ME.SparseTensor(torch.ones(4, 1), coordinates=torch.zeros(4, 4, dtype=torch.int32), device="cpu")


def test_model_registry_and_schema():
    from imfnet_amd.model import load_model, MODELS
    names = {m.__name__ for m in MODELS}
    assert {"ResUNet2", "ResUNetBN2", "ResUNetBN2B", "ResUNetBN2C", "ResUNetBN2D", "ResUNetBN2E",
            "ResUNetIN2", "ResUNetIN2B", "ResUNetIN2C", "ResUNetIN2D", "ResUNetIN2E"} <= names
    assert load_model("NoSuchNet") is None
    Model = load_model("ResUNetBN2C")
    m = Model(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_schema.json")))
    sd = m.state_dict()
    assert len(sd) == 361 and set(sd) == set(ref)
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    # strict load of reference-schema weights, and the older 'perceiver_io' prefix
    seeded = O.seeded_state_dict(0, with_unused_image_layers=True)
    res = m.load_state_dict(seeded, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert Model.CHANNELS == [None, 32, 64, 128, 256] and Model.TR_CHANNELS == [None, 64, 64, 64, 128]
    with pytest.raises(ValueError):
        load_model("ResUNet2")(1, 32, conv1_kernel_size=5, D=3)           # NORM_TYPE None, as upstream


def test_dense_submodules_match_reference_goldens(golden, images, seeded_sd):
    """AttentionFusion / ImageEncoder (torch modules, run here on CPU) vs outputs of the reference's
    own modules."""
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(seeded_sd, strict=True)
    m.eval()
    with torch.no_grad():
        out = m.attention_fusion(torch.as_tensor(golden["af_ctx"])[None],
                                 queries_encoder=torch.as_tensor(golden["af_in"])[None])[0]
        img = m.img_encoder(torch.as_tensor(images[0]))
    assert np.abs(out.numpy() - golden["af_out"]).max() < 2e-5
    assert np.abs(img.numpy() - golden["img_out"]).max() < 1e-5


def test_bn_folding_is_eval_batchnorm():
    import imfnet_amd.sparse as ME
    bn = ME.MinkowskiBatchNorm(8).eval()
    with torch.no_grad():
        bn.bn.weight.uniform_(0.5, 1.5); bn.bn.bias.uniform_(-1, 1)
        bn.bn.running_mean.normal_(); bn.bn.running_var.uniform_(0.5, 2)
    x = torch.randn(50, 8)
    sc, sh = bn.folded()
    assert torch.allclose(x * sc + sh, bn.bn(x), atol=1e-6)


def test_ply_reader_and_image_codecs(tmp_path, clouds):
    from imfnet_amd.dataio import image_to_nchw, process_image, read_image, read_ply_points, save_descriptors
    pts = clouds[0][:1000]
    p = tmp_path / "cloud_bin_0.ply"
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000\nproperty float x\n"
                b"property float y\nproperty float z\nend_header\n")
        f.write(pts.astype("<f4").tobytes())
    got = read_ply_points(str(p))
    assert got.dtype == np.float64 and (got == pts.astype(np.float64)).all()
    p2 = tmp_path / "d.ply"                                   # double xyz + uchar rgb (3D_head_map layout)
    with open(p2, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment x\nelement vertex 3\nproperty double x\n"
                b"property double y\nproperty double z\nproperty uchar red\nproperty uchar green\n"
                b"property uchar blue\nend_header\n")
        for i in range(3):
            f.write(struct.pack("<dddBBB", i + 0.5, -i, 2.0 * i, 1, 2, 3))
    assert read_ply_points(str(p2)).tolist() == [[0.5, 0, 0], [1.5, -1, 2], [2.5, -2, 4]]
    p3 = tmp_path / "a.ply"
    p3.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\n"
                  "property float z\nend_header\n1 2 3\n4 5 6\n")
    assert read_ply_points(str(p3)).tolist() == [[1, 2, 3], [4, 5, 6]]

    from PIL import Image
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    Image.fromarray(u8).save(tmp_path / "i.png")
    img = read_image(str(tmp_path / "i.png"))
    assert img.dtype == np.float32 and np.array_equal(img, np.divide(u8, 255, dtype=np.float32))
    small = process_image(img, aim_H=120, aim_W=160)
    ref = 0.25 * (img[1::4, 1::4] + img[1::4, 2::4] + img[2::4, 1::4] + img[2::4, 2::4])
    assert small.shape == (120, 160, 3) and np.abs(small - ref).max() < 1e-6
    assert process_image(small, aim_H=120, aim_W=160) is small or np.array_equal(process_image(small, 120, 160), small)
    assert image_to_nchw(small).shape == (1, 3, 120, 160)
    assert process_image(img, 120, 160, mode="clip").shape == (120, 160, 3)
    out = tmp_path / "o.npz"
    save_descriptors(str(out), got, got[:10], torch.ones(10, 32))
    z = np.load(out)
    assert sorted(z.files) == ["feature", "points", "xyz"] and z["feature"].dtype == np.float32
    assert z["points"].dtype == np.float64 and z["xyz"].shape == (10, 3)


def test_file_ordering_is_natural(tmp_path):
    from imfnet_amd.files import get_file_list, get_folder_list, sorted_alphanum
    assert sorted_alphanum(["cloud_bin_10.ply", "cloud_bin_2.ply", "cloud_bin_1.ply"]) == \
        ["cloud_bin_1.ply", "cloud_bin_2.ply", "cloud_bin_10.ply"]
    for n in ("cloud_bin_10.ply", "cloud_bin_9.ply", "cloud_bin_9_0.png"):
        (tmp_path / n).write_text("")
    (tmp_path / "sub2").mkdir(); (tmp_path / "sub10").mkdir()
    assert [os.path.basename(f) for f in get_file_list(str(tmp_path), ".ply")] == ["cloud_bin_9.ply", "cloud_bin_10.ply"]
    assert [os.path.basename(f) for f in get_folder_list(str(tmp_path))] == ["sub2", "sub10"]


def test_checkpoint_reader(tmp_path, seeded_sd):
    from imfnet_amd.checkpoint import Config, load_checkpoint
    sd = {k.replace("attention_fusion", "perceiver_io"): v for k, v in seeded_sd.items()}
    cfg = Config(voxel_size=0.025)
    torch.save({"state_dict": sd, "config": dict(cfg), "epoch": 1}, tmp_path / "c.pth")
    got, c = load_checkpoint(str(tmp_path / "c.pth"))
    assert set(got) == set(seeded_sd) and c["voxel_size"] == 0.025
    assert Config().model == "ResUNetBN2C" and Config().image_W == 160


def test_me_utils_host_helpers():
    import imfnet_amd.sparse as ME
    bc = ME.utils.batched_coordinates([np.zeros((2, 3), np.int32), np.ones((3, 3), np.int32)])
    assert bc.dtype == torch.int32 and bc[:, 0].tolist() == [0, 0, 1, 1, 1]
    a = np.array([[1, 2, 3], [-1, 0, 5]])
    assert (ME.utils.fnv_hash_vec(a) == O.fnv_hash_vec(a)).all()


def test_c_restatement_equals_numpy_oracle(clouds):
    import imf_oracle_cbind as OC
    xyz = clouds[1].astype(np.float64)
    c, i = OC.voxelize(xyz, 0.05)
    c2, i2 = O.voxelize(xyz, 0.05)
    assert (c == c2).all() and (i == i2).all()
    g, g2 = OC.Geometry(c), O.Geometry(c2)
    for a, b in zip(g.levels + [g.k_first] + g.k3 + g.down + g.up,
                    g2.levels + [g2.k_first] + g2.k3 + g2.down + g2.up):
        assert (a == b).all()


def test_resize_uint8_rounds_like_cv2():
    """process_image on 8-bit input (the .jpg branch, generate_desc.py:88-95): fixed-point bilinear, uint8 out --
    within 0.5 of the float interpolation, exact on a 2x down-sample of even blocks, identity when already sized."""
    from imfnet_amd.dataio import process_image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    out = process_image(img, aim_H=12, aim_W=16)
    assert out.dtype == np.uint8 and out.shape == (12, 16, 3)
    ref = process_image(img.astype(np.float32), aim_H=12, aim_W=16)
    assert np.abs(out.astype(np.float32) - ref).max() <= 0.5 + 1e-3
    half = process_image(img, aim_H=24, aim_W=32).astype(np.int64)
    blk = img.astype(np.int64).reshape(24, 2, 32, 2, 3).sum((1, 3))
    assert (half == (blk + 2) // 4).all()                                # centre average of 2x2 blocks, rounded
    assert process_image(img, aim_H=48, aim_W=64) is img
    up = process_image(img, aim_H=96, aim_W=128)
    assert up.dtype == np.uint8 and (up[0, 0] == img[0, 0]).all() and (up[-1, -1] == img[-1, -1]).all()


def test_conv_refuses_autograd():
    """No backward is implemented: a convolution whose kernel requires grad must fail loudly under autograd."""
    import torch
    from imfnet_amd import sparse as ME
    from imfnet_amd._lib import ImfError
    conv = ME.MinkowskiConvolution(32, 32, kernel_size=3, stride=1, dilation=1, bias=False, dimension=3)
    with pytest.raises(ImfError, match="inference-only"):
        conv.run(None)


def _write_ply(path, pts, fmt="binary_little_endian", extra=False, dtype="float"):
    import struct as st
    n = len(pts)
    head = f"ply\nformat {fmt} 1.0\ncomment test\nelement vertex {n}\n"
    if extra:
        head += "property uchar red\n"
    head += f"property {dtype} x\nproperty {dtype} y\n"
    if extra:
        head += "property float nx\n"
    head += f"property {dtype} z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
    with open(path, "wb") as f:
        f.write(head.encode())
        end = "<" if fmt != "binary_big_endian" else ">"
        code = "f" if dtype == "float" else "d"
        for p in pts:
            if fmt == "ascii":
                vals = ([7] if extra else []) + [repr(float(p[0])), repr(float(p[1]))] + ([0.5] if extra else []) + [repr(float(p[2]))]
                f.write((" ".join(str(v) for v in vals) + "\n").encode())
            else:
                if extra:
                    f.write(st.pack("B", 7))
                f.write(st.pack(end + code * 2, p[0], p[1]))
                if extra:
                    f.write(st.pack(end + "f", 0.5))
                f.write(st.pack(end + code, p[2]))
        f.write(b"\x03\x00\x00\x00\x00\x01\x00\x00\x00\x02\x00\x00\x00" if fmt != "ascii" else b"3 0 1 2\n")


@pytest.mark.parametrize("fmt", ["binary_little_endian", "binary_big_endian", "ascii"])
@pytest.mark.parametrize("dtype", ["float", "double"])
def test_native_ply_reader(tmp_path, clouds, fmt, dtype):
    """imf_ply_read_points vs the numpy reader on every layout Open3D reads: both byte orders, ascii, float / double
    coordinates, interleaved extra properties, a face element behind the vertices, empty and missing files."""
    from imfnet_amd import dataio
    pts = clouds[0][:997].astype(np.float32 if dtype == "float" else np.float64)
    for extra in (False, True):
        path = str(tmp_path / f"c_{extra}.ply")
        _write_ply(path, pts, fmt, extra, dtype)
        got = dataio.read_ply_points(path)
        assert got.dtype == np.float64 and got.shape == (997, 3)
        assert (got == pts.astype(np.float64)).all() and (got == dataio.read_ply_points_numpy(path)).all()
    _write_ply(str(tmp_path / "empty.ply"), pts[:0], fmt, False, dtype)
    assert dataio.read_ply_points(str(tmp_path / "empty.ply")).shape == (0, 3)
    buf = np.zeros((2000, 3))
    view = dataio.read_ply_points(str(tmp_path / "c_True.ply"), out=buf)
    assert view.base is buf and (view == pts.astype(np.float64)).all()
    with pytest.raises(ValueError):
        dataio.read_ply_points(str(tmp_path / "missing.ply"))


def test_native_jpeg_decoder_equals_pil(tmp_path):
    """The `.jpg` branch (scripts/generate_desc.py:88-92: matplotlib.image.imread = PIL = libjpeg's defaults): the native
    baseline decoder (csrc/jpeg.hip: integer islow inverse DCT, fancy chroma upsampling, fixed-point YCbCr -> RGB) returns PIL's
    bytes exactly -- 4:4:4 / 4:2:2 / 4:2:0, qualities 5-100, odd and degenerate sizes (components one or two samples wide are
    replicated, not filtered), restart intervals, optimised tables, saturating noise; what it does not cover (progressive)
    answers IMF_EUNSUPPORTED and read_image falls back to PIL; truncated / corrupt files are errors, not crashes."""
    import ctypes as C
    from PIL import Image
    from imfnet_amd import _lib, dataio
    L = _lib.lib()
    rng = np.random.default_rng(1)
    path = str(tmp_path / "t.jpg")

    def picture(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 100 * np.sin(x / 17.0 + y / 23.0), 127 + 90 * np.cos(x / 9.0) * np.sin(y / 31.0), (x * 3 + y * 5) % 256], -1)
        return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)

    def native(p):
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        rc = L.imf_jpeg_info(os.fsencode(p), C.byref(h), C.byref(w), C.byref(c))
        if rc:
            return rc, None
        out = np.empty((h.value, w.value, 3), np.uint8)
        rc = L.imf_jpeg_read_u8(os.fsencode(p), out.ctypes.data_as(C.c_void_p), out.size, C.byref(h), C.byref(w), C.byref(c))
        return rc, out

    n = 0
    for h, w in ((480, 640), (37, 53), (1, 1), (33, 2), (33, 3), (33, 5), (2, 33), (5, 6), (64, 48)):
        img = picture(h, w)
        for q in (5, 30, 75, 95, 100):
            for ss in (0, 1, 2):
                for kw in ({}, {"restart_marker_blocks": 2}, {"optimize": True}):
                    try:
                        Image.fromarray(img).save(path, quality=q, subsampling=ss, **kw)
                    except OSError:                      # (PIL's encoder buffer on a few tiny / quality-100 cases)
                        continue
                    rc, got = native(path)
                    assert rc == 0, (h, w, q, ss, kw, L.imf_last_error())
                    assert np.array_equal(got, np.asarray(Image.open(path))), (h, w, q, ss, kw)
                    n += 1
    assert n > 300
    noise = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    for q in (10, 50, 100):
        for ss in (0, 1, 2):
            Image.fromarray(noise).save(path, quality=q, subsampling=ss)
            rc, got = native(path)
            assert rc == 0 and np.array_equal(got, np.asarray(Image.open(path)))
    assert np.array_equal(dataio.read_image(path), np.asarray(Image.open(path)))           # the harness entry: uint8 HWC
    # not covered natively -> IMF_EUNSUPPORTED, read_image decodes with PIL
    Image.fromarray(picture(64, 64)).save(path, quality=80, progressive=True)
    assert native(path)[0] == -3 and b"progressive" in L.imf_last_error()
    assert np.array_equal(dataio.read_image(path), np.asarray(Image.open(path)))
    Image.fromarray(picture(64, 64)[:, :, 0]).save(path, quality=80)                          # grey: one component
    assert native(path)[0] == -3
    # truncated and corrupted streams: an error code (or a decoded picture), never a crash
    Image.fromarray(picture(120, 160)).save(path, quality=75)
    data = open(path, "rb").read()
    for cut in (1, 3, 20, 200, len(data) // 2):
        open(path, "wb").write(data[:cut])
        assert native(path)[0] != 0 or cut > 200
    for k in range(40):
        b = bytearray(data)
        for pos in rng.integers(2, len(b), 8):
            b[pos] = int(rng.integers(0, 256))
        open(path, "wb").write(bytes(b))
        native(path)


def test_native_png_reader_and_resize(tmp_path):
    """imf_png_read_f32 == matplotlib's imread semantics as PIL decodes them (8/16-bit, RGB / RGBA / grey / palette, every
    scan-line filter via real image content), interlaced files fall back; imf_resize_bilinear_f32 == the torch bilinear."""
    from PIL import Image
    from imfnet_amd import dataio
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:96, 0:128]
    smooth = np.stack([(yy * 2 + xx) % 256, (xx * 3) % 256, (yy * xx) % 256], -1).astype(np.uint8)
    noisy = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    cases = {"rgb_smooth": Image.fromarray(smooth), "rgb_noise": Image.fromarray(noisy),
             "rgba": Image.fromarray(np.concatenate([smooth, noisy[:, :, :1]], -1), "RGBA"),
             "grey": Image.fromarray(smooth[:, :, 0], "L"), "palette": Image.fromarray(smooth).convert("P"),
             "grey16": Image.fromarray((smooth[:, :, 0].astype(np.uint16) * 257), "I;16")}
    for name, im in cases.items():
        path = str(tmp_path / f"{name}.png")
        im.save(path, optimize=(name == "rgb_smooth"))
        got, ref = dataio.read_image(path), dataio.read_image_pil(path)
        assert got.dtype == np.float32 and got.shape == ref.shape, name
        assert (got == ref).all(), name
    img = dataio.read_image(str(tmp_path / "rgb_noise.png"))
    small = dataio.process_image(img, aim_H=24, aim_W=32)
    assert small.shape == (24, 32, 3) and np.abs(small - dataio.process_image_torch(img, 24, 32)).max() < 2e-6
    odd = dataio.process_image(img, aim_H=50, aim_W=70)
    assert np.abs(odd - dataio.process_image_torch(img, 50, 70)).max() < 2e-6
    up = dataio.process_image(img[:10, :12], aim_H=33, aim_W=40)
    assert np.abs(up - dataio.process_image_torch(img[:10, :12], 33, 40)).max() < 2e-6


@pytest.mark.parametrize("level", [0, 1, 6])
def test_native_npz_writer(tmp_path, level):
    """imf_npz_write: np.load returns exactly the arrays np.savez_compressed would have stored (keys, dtypes, shapes,
    values), for the descriptor-file layout and edge shapes (empty, 1-D, scalar-like, non-contiguous input)."""
    from imfnet_amd import dataio
    rng = np.random.default_rng(1)
    arrays = dict(points=rng.normal(size=(5001, 3)), xyz=rng.normal(size=(777, 3)),
                  feature=rng.normal(size=(777, 32)).astype(np.float32), idx=np.arange(13, dtype=np.int32),
                  empty=np.zeros((0, 32), np.float32), strided=rng.normal(size=(50, 8))[:, ::2])
    path = tmp_path / f"d{level}.npz"
    dataio.save_npz(str(path), level=level, **arrays)
    z = np.load(path)
    assert sorted(z.files) == sorted(arrays)
    for k, a in arrays.items():
        assert z[k].dtype == a.dtype and z[k].shape == a.shape and (z[k] == a).all(), k
    ref = tmp_path / "ref.npz"
    np.savez_compressed(ref, **arrays)
    if level == 0:
        assert os.path.getsize(path) > sum(a.nbytes for a in arrays.values())
    else:
        assert os.path.getsize(path) < 1.25 * os.path.getsize(ref)
    import zipfile
    assert zipfile.ZipFile(path).testzip() is None                        # CRCs and sizes are consistent
    dataio.save_descriptors(str(tmp_path / "frag.npz"), arrays["points"], arrays["xyz"], arrays["feature"])
    z = np.load(tmp_path / "frag.npz")
    assert sorted(z.files) == ["feature", "points", "xyz"] and z["points"].dtype == np.float64


def test_native_codecs_reject_corrupt_files(tmp_path, clouds):
    """Untrusted input: size fields taken from a file are checked against the file before anything is
    allocated, short chunks are refused, no C++ exception crosses the C ABI, unsupported PNG flavours (tRNS, 16-bit
    colour) are handed to the generic decoder, string arrays go to numpy's NPZ writer, a failing close is reported."""
    import ctypes as C
    import struct
    import zlib
    from PIL import Image
    from imfnet_amd import _lib, dataio
    L = _lib.lib()
    # PLY whose header promises 2^40 vertices over a 36-byte body
    p = tmp_path / "huge.ply"
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 1099511627776\nproperty float x\nproperty float y\n"
                  b"property float z\nend_header\n" + bytes(36))
    buf = np.empty((4, 3))
    assert L.imf_ply_read_points(os.fsencode(str(p)), buf.ctypes.data_as(C.c_void_p), 1 << 41) < 0
    assert b"declares" in L.imf_last_error()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    sig = b"\x89PNG\r\n\x1a\n"
    out = np.empty(64, np.float32)
    h, w, c = C.c_int(), C.c_int(), C.c_int()

    def read(path):
        return L.imf_png_read_f32(os.fsencode(str(path)), out.ctypes.data_as(C.c_void_p), out.size, C.byref(h), C.byref(w),
                                  C.byref(c))
    # IHDR chunk shorter than 13 bytes
    p = tmp_path / "short_ihdr.png"
    p.write_bytes(sig + chunk(b"IHDR", bytes(8)) + chunk(b"IDAT", zlib.compress(bytes(40))) + chunk(b"IEND", b""))
    assert read(p) == -1 and b"IHDR" in L.imf_last_error()
    # header promising 2^30 x 2^30 pixels over a few bytes of image data: refused before any allocation
    p = tmp_path / "bomb.png"
    p.write_bytes(sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 1 << 30, 1 << 30, 8, 2, 0, 0, 0))
                  + chunk(b"IDAT", zlib.compress(bytes(40))) + chunk(b"IEND", b""))
    assert read(p) == -1
    # tRNS and 16-bit RGB: IMF_EUNSUPPORTED, read_image falls back to the generic decoder
    rgb = np.random.default_rng(0).integers(0, 256, (4, 4, 3), dtype=np.uint8)
    pal = Image.fromarray(rgb).convert("P")
    pal.save(tmp_path / "trns.png", transparency=0)
    assert read(tmp_path / "trns.png") == -3
    raw = b"".join(b"\x00" + bytes(4 * 6) for _ in range(4))
    (tmp_path / "rgb16.png").write_bytes(sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 16, 2, 0, 0, 0))
                                         + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    assert read(tmp_path / "rgb16.png") == -3
    assert dataio.read_image(str(tmp_path / "trns.png")).shape[:2] == (4, 4)
    # a unicode array is written by numpy, not as a corrupt member
    dataio.save_npz(str(tmp_path / "s.npz"), names=np.array(["a", "bcd"]), x=np.arange(3.0))
    z = np.load(tmp_path / "s.npz")
    assert list(z["names"]) == ["a", "bcd"] and (z["x"] == np.arange(3.0)).all()
    names, dt = (C.c_char_p * 1)(b"names"), (C.c_char_p * 1)(b"<U3")
    nd, sh, data = (C.c_int32 * 1)(1), (C.c_int64 * 1)(2), (C.c_void_p * 1)(np.array(["a", "bcd"]).ctypes.data)
    assert L.imf_npz_write(os.fsencode(str(tmp_path / "u.npz")), 1, names, dt, nd, sh, data, 0) == -1
    # a device that cannot take the data: the error surfaces instead of a truncated archive reported as success
    if os.path.exists("/dev/full"):
        dt = (C.c_char_p * 1)(b"<f8")
        big = np.zeros(1 << 16)
        sh, data = (C.c_int64 * 1)(big.size), (C.c_void_p * 1)(big.ctypes.data)
        assert L.imf_npz_write(b"/dev/full", 1, names, dt, nd, sh, data, 0) == -1


def test_npz_block_parallel_deflate_is_thread_count_independent(tmp_path):
    """imf_npz_write_mt: np.load returns the arrays bit for bit at every level; the FILE's bytes depend on the arrays and
    the level only, never on the number of deflate threads; a descriptor-like float32 member (incompressible for LZ77) is
    Huffman-coded, a point-like member keeps LZ77 -- both read back; empty, scalar and sub-block members too."""
    import zipfile
    from imfnet_amd.dataio import save_npz
    rng = np.random.default_rng(0)
    pts = np.cumsum(rng.normal(size=(120000, 3)).astype(np.float32), 0).astype(np.float64)      # 2.9 MB, compressible
    F = rng.normal(size=(30000, 32)).astype(np.float32)                                         # 3.8 MB, not
    arrays = dict(points=pts, xyz=pts[::7].copy(), feature=F, empty=np.zeros((0, 3)), one=np.float32(2.5).reshape(()),
                  small=np.arange(1000, dtype=np.int32), edge=np.arange(256 << 10, dtype=np.uint8))
    ref = {}
    for level in (0, 1, 6):
        for threads in (1, 3, 8):
            path = str(tmp_path / f"l{level}_t{threads}.npz")
            save_npz(path, level=level, threads=threads, **arrays)
            z = np.load(path)
            assert set(z.files) == set(arrays)
            for k, a in arrays.items():
                assert z[k].dtype == a.dtype and z[k].shape == a.shape and (z[k] == a).all(), (level, threads, k)
            assert zipfile.ZipFile(path).testzip() is None
            data = open(path, "rb").read()
            assert ref.setdefault(level, data) == data, f"level {level}: bytes differ with {threads} threads"
    info = {i.filename: i for i in zipfile.ZipFile(str(tmp_path / "l1_t8.npz")).infolist()}
    assert info["points.npy"].compress_size < 0.8 * info["points.npy"].file_size
    assert info["feature.npy"].compress_size < info["feature.npy"].file_size


def test_process_image_clip_and_padding_modes():
    """util/uio.py:41-99, the branches generate_desc never takes.  No OpenCV here: the pyramid filters are checked through
    what cv2.pyrUp / pyrDown guarantee by definition (sizes, constants stay constants, 8-bit stays 8-bit, the Gaussian's
    weights on an impulse), the window arithmetic and the padding branch's behaviour AS WRITTEN in the reference."""
    from imfnet_amd.dataio import _pyr_down, _pyr_up, process_image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    flat = np.full((11, 13, 3), 77, np.uint8)
    assert _pyr_up(flat).shape == (22, 26, 3) and (_pyr_up(flat) == 77).all()
    assert _pyr_down(flat).shape == (6, 7, 3) and (_pyr_down(flat) == 77).all()
    imp = np.zeros((9, 9, 1), np.float32)
    imp[4, 4] = 256.0
    assert np.array_equal(_pyr_down(imp)[1:4, 1:4, 0], np.outer([1, 6, 1], [1, 6, 1]).astype(np.float32))   # taps at distance 2, 0, 2
    up = _pyr_up(imp)[6:11, 6:11, 0]
    assert np.array_equal(up, np.outer([1, 4, 6, 4, 1], [1, 4, 6, 4, 1]).astype(np.float32) / 64 * 256)
    # clip: 50x70 -> doubled twice (200x280) to cover 120x160, centre window
    big = _pyr_up(_pyr_up(img))
    out = process_image(img, 120, 160, mode="clip")
    assert out.dtype == np.uint8 and np.array_equal(out, big[40:160, 60:220])
    assert np.array_equal(process_image(img, 120, 160, mode="clip", clip_mode="normal"), big[:120, :160])
    np.random.seed(4)
    top, left = int(np.random.random() * 80), int(np.random.random() * 120)
    np.random.seed(4)
    assert np.array_equal(process_image(img, 120, 160, mode="clip", clip_mode="random"), big[top:top + 120, left:left + 160])
    # more than twice the target in both directions: halved once
    assert np.array_equal(process_image(img, 20, 30, mode="clip", clip_mode="normal"), _pyr_down(img)[:20, :30])
    # padding, as the reference behaves
    assert process_image(img, 60, 80, mode="padding").shape == (50, 70, 3)            # smaller image: returned as it is
    cut = process_image(img, 40, 80, mode="padding")                                  # taller, narrower: cut + zero columns
    assert cut.shape == (40, 80, 3) and cut.dtype == np.float64
    assert np.array_equal(cut[:, :70], img[:40]) and (cut[:, 70:] == 0).all()
    with pytest.raises(ValueError):
        process_image(img, 40, 60, mode="padding")                                    # larger both ways: negative block
    assert process_image(img, 40, 60, mode="something else") is not None              # unknown mode: falls through unchanged


def test_instance_norm_module_matches_the_oracle():
    """sparse.MinkowskiInstanceNorm against oracle.instance_norm on a ragged three-item batch (the module only touches
    .F, .C and ._like of its input: a stand-in carrier, since SparseTensor itself lives on the GPU)."""
    import imfnet_amd.sparse as ME

    class Rows:
        def __init__(self, F, C):
            self.F, self.C = F, C

        def _like(self, F):
            return Rows(F, self.C)
    g = torch.Generator().manual_seed(2)
    n = [37, 5, 120]
    item = np.concatenate([np.full(k, i, np.int32) for i, k in enumerate(n)])
    coords = torch.as_tensor(np.stack([item, np.arange(len(item)), np.zeros(len(item)), np.zeros(len(item))], 1).astype(np.int32))
    f = torch.randn(len(item), 16, generator=g) * 3 + 1
    norm = ME.MinkowskiInstanceNorm(16)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(1, 16, generator=g) + 0.5)
        norm.bias.copy_(torch.rand(1, 16, generator=g) - 0.5)
        out = norm(Rows(f, coords)).F
    ref = O.instance_norm(f, item, norm.weight.detach(), norm.bias.detach())
    assert float((out - ref).abs().max()) < 1e-5
    one = norm(Rows(f[:37], coords[:37])).F.detach()
    assert float((one - ref[:37]).abs().max()) < 1e-5


def test_fast_deflate_producers_round_trip_through_zlib(tmp_path, clouds):
    """csrc/fast_deflate.h (level 1 of imf_npz_write_mt): every member is inflated by zlib (zipfile / np.load) to exactly the
    input -- point-like float64 (value-granular matches), descriptor-like float32 (byte Huffman), and the corner cases of a
    hand-written encoder: incompressible data (stored-block fallback), one repeated value (long matches at distance 8), a
    single distinct byte (one-symbol code), frequencies that force the 15-bit length limit, lengths around the 256 KiB
    segment and 65 535-byte stored-block boundaries, empty and one-element members.  On the in-tree fragment the file is
    smaller than zlib level 1's (IMFNET_NPZ_ZLIB=1 in a subprocess) and np.load returns identical arrays from both."""
    import subprocess
    import sys
    import zipfile
    from imfnet_amd.dataio import save_npz
    rng = np.random.default_rng(3)
    pts = (clouds[0] * np.float32(1.3)).astype(np.float64)
    F = rng.normal(size=(len(pts) // 6, 32)).astype(np.float32)
    F /= np.linalg.norm(F, axis=1, keepdims=True)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    skew = np.concatenate([np.full(min(f, 200000), i, np.uint8) for i, f in enumerate(fib[:34])])     # deep Huffman tree
    rng.shuffle(skew)
    arrays = dict(points=pts, xyz=pts[::5].copy(), feature=F,
                  random64=rng.integers(0, 2 ** 63, size=70001, dtype=np.int64),                      # nothing repeats: stored
                  const64=np.full(100003, 3.25), ramp64=np.arange(40000, dtype=np.float64),
                  period64=np.tile(rng.normal(size=37), 3000),                                        # matches at distance 37 values
                  const8=np.full((256 << 10) + 5, 7, np.uint8), skew8=skew,
                  noise8=rng.integers(0, 256, size=(256 << 10) * 2 + 65535 + 3, dtype=np.uint8),
                  f32zero=np.zeros(65536 * 4 + 1, np.float32), empty64=np.zeros((0, 3)), one64=np.array([1.5]),
                  block64=np.arange((256 << 10) // 8, dtype=np.int64) % 97, i4=np.arange(5000, dtype=np.int32) % 11)
    path = str(tmp_path / "fast.npz")
    for threads in (1, 5):
        save_npz(path, level=1, threads=threads, **arrays)
        z = np.load(path)
        assert set(z.files) == set(arrays)
        for k, a in arrays.items():
            assert z[k].dtype == a.dtype and z[k].shape == a.shape and (z[k] == a).all(), k
        assert zipfile.ZipFile(path).testzip() is None
    info = {i.filename[:-4]: i for i in zipfile.ZipFile(path).infolist()}
    assert info["points"].compress_size < 0.16 * info["points"].file_size            # the in-tree fragment: 0.135
    assert info["const64"].compress_size < 0.01 * info["const64"].file_size
    assert info["period64"].compress_size < 0.05 * info["period64"].file_size
    # stored fallback: 5 bytes per 65 535-byte stored block + the segment's sync marker, never more
    assert info["random64"].compress_size <= info["random64"].file_size * 1.0002 + 64
    assert info["noise8"].compress_size <= info["noise8"].file_size * 1.0002 + 64
    assert info["feature"].compress_size < 0.95 * info["feature"].file_size
    # against zlib level 1 on the descriptor-file members (the library reads IMFNET_NPZ_ZLIB once per process)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from imfnet_amd.dataio import save_npz; z = np.load(%r);"
            "save_npz(%r, level=1, threads=2, points=z['points'], xyz=z['xyz'], feature=z['feature'])"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, str(tmp_path / "zlib.npz")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, IMFNET_NPZ_ZLIB="1"))
    save_npz(str(tmp_path / "mine.npz"), level=1, threads=2, points=pts, xyz=arrays["xyz"], feature=F)
    a, b = np.load(tmp_path / "mine.npz"), np.load(tmp_path / "zlib.npz")
    assert all((a[k] == b[k]).all() for k in ("points", "xyz", "feature"))
    assert os.path.getsize(tmp_path / "mine.npz") <= os.path.getsize(tmp_path / "zlib.npz")


def test_fast_deflate_fuzz(tmp_path):
    """Property test of csrc/fast_deflate.h through the NPZ writer: whatever the bytes -- random lengths, alphabets from one
    symbol to all 256, heavy skew, runs, periodic values, values whose repeats sit just inside / outside the 32 KiB window --
    numpy (zlib inflate + CRC check) reads back exactly what was written."""
    from hypothesis import given, settings, strategies as st
    from imfnet_amd.dataio import save_npz
    path = str(tmp_path / "f.npz")

    @settings(max_examples=60, deadline=None)
    @given(st.integers(0, 2 ** 32 - 1), st.integers(0, 70000), st.integers(1, 256), st.integers(1, 5000), st.booleans())
    def run(seed, n, alphabet, period, skew):
        rng = np.random.default_rng(seed)
        p = rng.dirichlet(np.full(alphabet, 0.05 if skew else 5.0))
        u8 = rng.choice(alphabet, size=n, p=p).astype(np.uint8)
        base = rng.normal(size=period)
        f8 = np.tile(base, n // period + 1)[:n].copy()
        f8[rng.integers(0, max(n, 1), size=n // 50)] = rng.normal(size=n // 50) if n else 0.0
        far = np.arange(max(n, 1), dtype=np.float64) % 4097              # repeats at distance 4097 values: outside the window
        near = np.arange(max(n, 1), dtype=np.float64) % 4096             # at 4096: the last distance deflate can express
        f4 = rng.normal(size=n).astype(np.float32)
        arrays = dict(u8=u8, f8=f8, far=far, near=near, f4=f4, i8=rng.integers(-3, 3, size=n).astype(np.int64))
        save_npz(path, level=1, threads=1 + seed % 3, **arrays)
        z = np.load(path)
        for k, a in arrays.items():
            assert z[k].dtype == a.dtype and z[k].shape == a.shape and (z[k] == a).all(), (k, seed, n)

    run()
