"""ctypes binding of libimfnet_hip.so (include/imfnet_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMF_LIB") or os.path.join(_HERE, "libimfnet_hip.so")   # IMF_LIB: diagnostic builds

TILE_ROWS = 64
MASK_WORDS = 4
MAX_KVOL = 125
MAX_BATCH = 8
FLAG_RANGE = 32


class ImfError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    """struct imf_conv_args (include/imfnet_hip.h)."""
    _fields_ = [
        ("in_a", C.c_void_p), ("in_b", C.c_void_p),
        ("c_a", C.c_int32), ("c_b", C.c_int32),
        ("w_packed", C.c_void_p),
        ("kvol", C.c_int32), ("cout", C.c_int32),
        ("tile_rows", C.c_void_p), ("nbr", C.c_void_p),
        ("tile_mask", C.c_void_p),
        ("n_slots", C.c_int64), ("n_out", C.c_int64),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
        ("relu", C.c_int32), ("l2norm", C.c_int32),
        ("out", C.c_void_p),
        ("split_k", C.c_int32), ("variant", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("tickets", C.c_void_p),
        ("ev_begin", C.c_void_p), ("ev_end", C.c_void_p),
        ("n_out_dev", C.c_void_p), ("dyn_split_kvol", C.c_int32), ("slots_extra", C.c_int32),
        ("kernel_tag", C.c_int32), ("dyn_err", C.c_void_p), ("geglu", C.c_int32), ("operand_format", C.c_int32),
    ]


class HeadArgs(C.Structure):
    """struct imf_head_args (include/imfnet_hip.h)."""
    _fields_ = [
        ("in_a", C.c_void_p), ("in_b", C.c_void_p), ("c_a", C.c_int32), ("c_b", C.c_int32),
        ("w1_packed", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p),
        ("relu1", C.c_int32), ("c_mid", C.c_int32),
        ("w2_packed", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p),
        ("l2norm", C.c_int32), ("c_out", C.c_int32),
        ("n", C.c_int64), ("n_dev", C.c_void_p), ("out", C.c_void_p), ("flags", C.c_void_p),
        ("ev_begin", C.c_void_p), ("ev_end", C.c_void_p), ("a_split", C.c_int32), ("variant", C.c_int32),
    ]


class LevelDesc(C.Structure):
    """struct imf_level (include/imfnet_hip.h)."""
    _fields_ = [("coords", C.c_void_p), ("table", C.c_void_p),
                ("capacity", C.c_int64), ("first_idx", C.c_void_p), ("cap_rows", C.c_int64),
                ("tensor_stride", C.c_int32)]


class FusionWeights(C.Structure):
    """struct imf_fusion_weights (include/imfnet_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "wq_p", "wo_p", "bo", "ln2_g", "ln2_b", "w1_p", "b1",
                                          "w2_p", "b2", "w1_f32", "w2_f32")]


class NetConv(C.Structure):
    """struct imf_net_conv."""
    _fields_ = [("w_packed", C.c_void_p), ("kvol", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int32), ("l2norm", C.c_int32),
                ("variant", C.c_int32)]


class ResunetDesc(C.Structure):
    """struct imf_resunet_desc."""
    _fields_ = [("channels", C.c_int32 * 5), ("tr_channels", C.c_int32 * 5), ("in_channels", C.c_int32),
                ("out_channels", C.c_int32), ("first_ksize", C.c_int32), ("small_first", C.c_int32),
                ("conv", NetConv * 23), ("first_kernel", C.c_void_p), ("first_scale", C.c_void_p),
                ("first_shift", C.c_void_p), ("fusion", FusionWeights), ("fusion_scale", C.c_float),
                ("first_kernel_image", C.c_void_p)]


class ImageDesc(C.Structure):
    """struct imf_image_desc."""
    _fields_ = [("stem_w", C.c_void_p), ("stem_scale", C.c_void_p), ("stem_shift", C.c_void_p),
                ("conv", NetConv * 15), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("kv_w", C.c_void_p),
                ("variant", C.c_int32)]


class NetTrace(C.Structure):
    """struct imf_net_trace."""
    _fields_ = [("ev_begin", C.c_void_p), ("ev_end", C.c_void_p), ("nbr", C.c_void_p), ("kvol", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("split", C.c_int32), ("n_slots", C.c_int64),
                ("n_out", C.c_int64), ("launched", C.c_int32), ("level", C.c_int32), ("slots_extra", C.c_int32),
                ("kernel_tag", C.c_int32)]


class ResunetIO(C.Structure):
    """struct imf_resunet_io."""
    _fields_ = [("level", LevelDesc * 4), ("n", C.c_int64 * 4), ("bbox", C.c_void_p), ("x", C.c_void_p),
                ("x_all_ones", C.c_int32), ("n_items", C.c_int32), ("item_row0", C.c_int64 * MAX_BATCH),
                ("item_rows", C.c_int64 * MAX_BATCH), ("kt_packed", C.c_void_p * MAX_BATCH),
                ("v_packed", C.c_void_p * MAX_BATCH),
                ("n_tokens", C.c_int32), ("tokens_padded", C.c_int32), ("image_ready", C.c_void_p),
                ("fusion_done", C.c_void_p), ("int_arena", C.c_void_p), ("int_arena_bytes", C.c_size_t),
                ("float_arena", C.c_void_p), ("float_arena_bytes", C.c_size_t), ("out", C.c_void_p),
                ("events", C.c_void_p * 16), ("side_stream", C.c_void_p), ("main_stream", C.c_void_p),
                ("trace", C.POINTER(NetTrace)), ("dyn", C.c_int32), ("meta", C.c_void_p),
                ("bitgrid_words", C.c_size_t), ("pyramid", C.c_void_p), ("flags", C.c_void_p),
                ("fp32_buffers", C.c_int32)]


DYN_WORDS = 16
META_WORDS = 64


class FragmentCaps(C.Structure):
    """struct imf_fragment_caps."""
    _fields_ = [("n_points", C.c_int64), ("rows", C.c_int64 * 4), ("n_items", C.c_int32), ("img_h", C.c_int32),
                ("img_w", C.c_int32), ("bitgrid_words", C.c_size_t)]


class FragmentIO(C.Structure):
    """struct imf_fragment_io."""
    _fields_ = [("xyz", C.c_void_p), ("xyz_is_f64", C.c_int32), ("voxel_size", C.c_double), ("dyn", C.c_void_p),
                ("image", C.c_void_p), ("meta", C.c_void_p), ("pyramid_arena", C.c_void_p),
                ("pyramid_arena_bytes", C.c_size_t), ("image_ws", C.c_void_p), ("image_ws_bytes", C.c_size_t),
                ("kt_packed", C.c_void_p), ("v_packed", C.c_void_p), ("tokens_padded", C.c_int32),
                ("int_arena", C.c_void_p), ("int_arena_bytes", C.c_size_t), ("float_arena", C.c_void_p),
                ("float_arena_bytes", C.c_size_t), ("out", C.c_void_p), ("events", C.c_void_p * 16),
                ("main_stream", C.c_void_p), ("side_stream", C.c_void_p), ("image_stream", C.c_void_p),
                ("trace", C.POINTER(NetTrace)), ("levels", LevelDesc * 4), ("serialize", C.c_int32),
                ("fp32_buffers", C.c_int32), ("head_on_side", C.c_int32), ("gpu_idle_hint", C.c_int32),
                ("inputs_event", C.c_void_p), ("reuse_event", C.c_void_p)]


class Job(C.Structure):
    """struct imf_job (one forward of the streaming pipeline)."""
    _fields_ = [("net", C.POINTER(ResunetDesc)), ("img", C.POINTER(ImageDesc)), ("caps", C.POINTER(FragmentCaps)),
                ("io", C.POINTER(FragmentIO)), ("host_in", C.c_void_p), ("dev_in", C.c_void_p), ("in_bytes", C.c_size_t),
                ("dev_out", C.c_void_p), ("host_out", C.c_void_p), ("out_bytes", C.c_size_t), ("sel", C.c_void_p),
                ("sel_offset", C.c_size_t), ("out_offset", C.c_size_t), ("out_row_bytes", C.c_int32),
                ("defer_download", C.c_int32)]


PIPELINE_SDMA_COPIES = 1

_P, _I, _L, _D, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/imfnet_hip.h declares
SIGNATURES = {
    "imf_version": (_I, []),
    "imf_last_error": (C.c_char_p, []),
    "imf_spconv_wgrad_workspace_bytes": (_Z, [_L, _I, _I, _I]),
    "imf_spconv_wgrad": (_I, [_P, _I, _P, _I, _P, _P, _L, _L, _I, _P, _P, _Z, _P]),
    "imf_ply_vertex_count": (_L, [C.c_char_p]),
    "imf_ply_read_points": (_L, [C.c_char_p, _P, _L]),
    "imf_png_info": (_I, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imf_png_read_f32": (_I, [C.c_char_p, _P, _L, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imf_jpeg_info": (_I, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imf_jpeg_read_u8": (_I, [C.c_char_p, _P, _L, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "imf_resize_bilinear_f32": (_I, [_P, _I, _I, _I, _P, _I, _I, _I]),
    "imf_npz_write": (_I, [C.c_char_p, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                           C.POINTER(C.c_int64), C.POINTER(C.c_void_p), _I]),
    "imf_npz_write_mt": (_I, [C.c_char_p, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int64), C.POINTER(C.c_void_p), _I, _I]),
    "imf_stream_create": (_P, []),
    "imf_stream_destroy": (None, [_P]),
    "imf_event_create": (_P, []),
    "imf_event_destroy": (None, [_P]),
    "imf_event_record": (_I, [_P, _P]),
    "imf_event_elapsed_ms": (C.c_float, [_P, _P]),
    "imf_nn_workspace_bytes": (_Z, [_L, _L]),
    "imf_nn_search": (_I, [_P, _L, _P, _L, _I, _P, _P, _P, _Z, _P]),
    "imf_mutual_inliers": (_I, [_P, _L, _P, _L, _P, _P, _P, _D, _P, _P, _P]),
    "imf_keypoint_workspace_bytes": (_Z, [_L, _L]),
    "imf_select_keypoints": (_I, [_P, _L, _P, _L, _D, _P, _P, _P, _Z, _P]),
    "imf_resunet_int_arena_bytes": (_Z, [C.POINTER(ResunetDesc), C.POINTER(C.c_int64), _P]),
    "imf_resunet_float_arena_bytes": (_Z, [C.POINTER(ResunetDesc), C.POINTER(C.c_int64)]),
    "imf_resunet_conv_kernel_tag": (_I, [_I, _I, _I, _I, _I, _I]),
    "imf_resunet_forward": (_I, [C.POINTER(ResunetDesc), C.POINTER(ResunetIO)]),
    "imf_resunet_int_arena_bytes_cap": (_Z, [C.POINTER(ResunetDesc), C.POINTER(C.c_int64), _Z]),
    "imf_resunet_float_arena_bytes_cap": (_Z, [C.POINTER(ResunetDesc), C.POINTER(C.c_int64)]),
    "imf_fragment_pyramid_bytes": (_Z, [C.POINTER(FragmentCaps)]),
    "imf_fragment_forward": (_I, [C.POINTER(ResunetDesc), C.POINTER(ImageDesc), C.POINTER(FragmentCaps),
                                  C.POINTER(FragmentIO)]),
    "imf_pipeline_create": (_P, [_P, _I, _I]),
    "imf_pipeline_destroy": (None, [_P]),
    "imf_pipeline_submit": (_I, [_P, C.POINTER(Job)]),
    "imf_pipeline_wait": (_I, [_P, _I, C.POINTER(C.c_float)]),
    "imf_host_narrow_points": (_I, [_P, _L, _P]),
    "imf_graph_begin_capture": (_I, [_P]),
    "imf_graph_end_capture": (_I, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "imf_graph_abort_capture": (_I, [_P]),
    "imf_graph_launch": (_I, [_P, _P]),
    "imf_graph_destroy": (None, [_P]),
    "imf_pyramid_arena_bytes_caps": (_Z, [_L, _I, C.POINTER(C.c_int64)]),
    "imf_pyramid_build_dyn": (_I, [_P, _I, _P, _L, C.POINTER(C.c_int64), _D, _I, _P, _Z, _P, C.POINTER(LevelDesc), _P]),
    "imf_rulebook_conv_dyn": (_I, [_P, _L, _P, _L, _P, _I, _I, _P, _P, _P, _P]),
    "imf_rulebook_transpose_dyn": (_I, [_P, _L, _P, _L, _P, _I, _I, _P, _P, _P, _L, _P, _P]),
    "imf_conv_first_bitgrid_dyn": (_I, [_P, _L, _P, _P, _P, _I, _P, _Z, _P, _I, _P, _P, _I, _P, _P]),
    "imf_fusion_workspace_bytes_cap": (_Z, [_L]),
    "imf_fusion_attention_dyn": (_I, [_P, _L, _P, _P, _I, _P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I,
                                      C.POINTER(FusionWeights), C.c_float, _P, _P, _Z, _P]),
    "imf_ransac_workspace_bytes": (_Z, [_I]),
    "imf_ransac_registration": (_I, [_P, _L, _P, _L, _P, _I, _D, _D, _I, C.c_uint64, _P, _P, _P, _P, _Z, _P]),
    "imf_gather_points": (_I, [_P, _I, _P, _P, _L, _P, _P]),
    "imf_hash_capacity": (_L, [_L]),
    "imf_unique_workspace_bytes": (_Z, [_L]),
    "imf_voxelize": (_I, [_P, _I, _L, _D, _I, _P, _P, _P, _P, _L, _P, _P, _P]),
    "imf_downsample": (_I, [_P, _P, _L, _I, _P, _P, _P, _L, _P, _P]),
    "imf_pyramid_arena_bytes": (_Z, [_L, _I]),
    "imf_pyramid_build": (_I, [_P, _I, _L, _D, _I, _I, _P, _Z, _P, C.POINTER(LevelDesc), _P]),
    "imf_pyramid_build_batched": (_I, [_P, _I, _L, _D, C.POINTER(C.c_int64), _I, _I, _P, _Z, _P,
                                       C.POINTER(LevelDesc), _P]),
    "imf_rulebook_slots": (_L, [_L]),
    "imf_rulebook_conv": (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _P, _P]),
    "imf_rulebook_sorted_workspace_bytes": (C.c_size_t, [_L]),
    "imf_rulebook_sort_by_occupancy": (_I, [_P, _I, _L, _L, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "imf_resunet_sorted_maps": (_I, [_I]),
    "imf_resunet_sorted_maps_n": (_I, [_I, _I]),
    "imf_rulebook_transpose_slots": (_L, [_L]),
    "imf_rulebook_transpose": (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _P, _L, _P, _P]),
    "imf_packed_weight_floats": (_L, [_I, _I, _I]),
    "imf_packed_weight_floats_split16": (_L, [_I, _I, _I]),
    "imf_pack_weights": (_I, [_P, _I, _I, _I, _P, _P]),
    "imf_first_kernel_image_floats": (_L, [_I, _I]),
    "imf_pack_first_kernel": (_I, [_P, _I, _I, _P, _P]),
    "imf_pack_weights_split16": (_I, [_P, _I, _I, _I, _P, _P]),
    "imf_packed_weight_floats_bf16x3": (_L, [_I, _I, _I]),
    "imf_pack_weights_bf16x3": (_I, [_P, _I, _I, _I, _P, _P]),
    "imf_spconv_auto_split": (_I, [_L, _I, _I]),
    "imf_spconv_max_split": (_I, [_I, _I]),
    "imf_spconv_occupancy": (_I, [_I, _I, _I]),
    "imf_spconv_workspace_bytes": (_Z, [_L, _I, _I]),
    "imf_spconv_fwd": (_I, [C.POINTER(ConvArgs), _P]),
    "imf_pointwise_head": (_I, [C.POINTER(HeadArgs), _P]),
    "imf_spconv_small_cin": (_I, [_P, _I, _P, _I, _I, _P, _L, _L, _P, _P, _I, _P, _P]),
    "imf_fusion_workspace_bytes": (_Z, [_L]),
    "imf_fusion_attention_batched": (_I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), _I, _I, C.POINTER(FusionWeights), C.c_float, _P, _P,
                                          _Z, _P]),
    "imf_fusion_attention": (_I, [_P, _L, _P, _P, _I, _I, C.POINTER(FusionWeights), C.c_float, _P, _P, _Z, _P]),
    "imf_fusion_attention_batched_v": (_I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), _I, _I, C.POINTER(FusionWeights), C.c_float, _P, _P,
                                            _Z, _P, _I, _P]),
    "imf_image_workspace_bytes": (_Z, [_I, _I, _I]),
    "imf_image_tokens": (_I, [_I, _I]),
    "imf_image_tables_build": (_I, [_I, _I, _I, _P, _Z, _P]),
    "imf_image_branch": (_I, [C.POINTER(ImageDesc), _P, _I, _I, _I, _P, _Z, _P, _P, _P, _I, _P, _P]),
    "imf_conv_first_bitgrid_flags": (_I, [_P, _L, _P, _I, _P, _Z, _P, _I, _P, _P, _I, _P, _P, _P]),
    "imf_fusion_attention_batched_flags": (_I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_void_p), _I, _I, C.POINTER(FusionWeights), C.c_float, _P, _P,
                                                _Z, _P, _P]),
    "imf_bitgrid_words": (_Z, [_P, _I]),
    "imf_conv_first_bitgrid": (_I, [_P, _L, _P, _I, _P, _Z, _P, _I, _P, _P, _I, _P, _P]),
    "imf_conv_first_fused": (_I, [_P, _L, _P, _L, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises ImfError (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImfError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C imfnet_amd/csrc`.  imfnet_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().imf_last_error().decode("utf-8", "replace")
        raise ImfError(f"{what} failed (rc={rc}): {msg}")
