"""ctypes binding of the C ABI declared in include/tspn_mi355x.h.

No torch types cross this boundary: only raw device pointers, sizes and a
hipStream_t.  There is NO fallback: if the shared library is missing or a
symbol cannot be resolved, importing callers get a RuntimeError.
"""
import ctypes
import os
import re
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# TSPN_LIB_PATH selects another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("TSPN_LIB_PATH") or os.path.join(HERE, "libtspn_mi355x.so")
HEADER_PATH = os.path.join(ROOT, "include", "tspn_mi355x.h")

ABI_VERSION = 7   # TSPN_ABI_VERSION of include/tspn_mi355x.h
TSPN_OK = 0
TSPN_EINVAL = -1
TSPN_EUNSUPPORTED = -2
TSPN_EWORKSPACE = -3
TSPN_ELAUNCH = -4
TSPN_EDEVICE = -5     # a kernel of an earlier launch raised a fault through the device status block
STATUS_WORDS = 16
STATUS_FAULT, STATUS_FAULT_INFO, STATUS_CONV_ERR, STATUS_CONV_CHECKS = 0, 1, 2, 3
FAULT_HANDOVER = 1
CONV_CHECK_HOT_OFFSET, CONV_CHECK_SCRATCH_BYTES = 256, 256 + 256 * 64   # tspn_conv3_spot_check_f32's scratch layout
GEOM_CHANNELS = 8
CONV_DIRECT, CONV_WINOGRAD63 = 0, 1   # tspn_fused_desc.conv_algo

_c_f32p = ctypes.c_void_p   # device pointers travel as integers
_c_i64p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class FusedDesc(ctypes.Structure):
    """struct tspn_fused_desc (include/tspn_mi355x.h)."""
    _fields_ = [
        ("B", _i64), ("N", _i64), ("T", _i64), ("D", _i64),
        ("A", _i64), ("K", _i64),
        ("feats", _vp), ("pairs", _vp), ("P", _i64), ("canonical_pairs", _i64),
        ("conv_packed", _vp), ("conv_algo", _i64), ("conv_bias", _vp),
        ("head_w", _vp), ("head_b", _vp),
        ("cls_w", _vp), ("cls_b", _vp),
        ("out_heads", _vp), ("out_logits", _vp),
        ("workspace", _vp), ("workspace_bytes", _sz),
        ("ev_conv_begin", _vp), ("ev_conv_end", _vp), ("ev_logits_ready", _vp),
        ("conv_weight", _vp), ("conv_check", _i64),
    ]


class FusedBf16Desc(ctypes.Structure):
    """struct tspn_fused_bf16_desc (include/tspn_mi355x.h)."""
    _fields_ = [
        ("B", _i64), ("N", _i64), ("T", _i64), ("D", _i64),
        ("A", _i64), ("K", _i64),
        ("feats", _vp), ("pairs", _vp), ("P", _i64),
        ("conv_packed", _vp), ("conv_bias", _vp),
        ("head_packed", _vp), ("head_b", _vp),
        ("cls_w", _vp), ("cls_b", _vp),
        ("out_heads", _vp), ("out_logits", _vp),
        ("workspace", _vp), ("workspace_bytes", _sz),
        ("ev_conv_begin", _vp), ("ev_conv_end", _vp), ("ev_logits_ready", _vp),
    ]


# name -> (restype, argtypes); mirrors the header one-to-one
PROTOTYPES = {
    "tspn_version": (_int, []),
    "tspn_fused_desc_size": (_sz, []),
    "tspn_fused_bf16_desc_size": (_sz, []),
    "tspn_last_error": (ctypes.c_char_p, []),
    "tspn_error_string": (ctypes.c_char_p, [_int]),
    "tspn_status_attach": (_int, [_vp]),
    "tspn_status_fault": (_int, []),
    "tspn_status_clear": (_int, []),
    "tspn_status_selftest": (_int, [_vp, _vp]),
    "tspn_conv3_spot_check_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _int, _vp, _i64, _vp, _i64, _vp]),
    "tspn_predicate_head_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "tspn_predicate_head_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _int, _vp, _sz, _vp]),
    "tspn_predicate_head_norm_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _i64]),
    "tspn_predicate_head_norm_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int,
                                            _vp, _sz, _vp]),
    "tspn_feature_preprocess_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "tspn_proposal_pair_filter_i64": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tspn_gather_rows_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp]),
    "tspn_ppn_pair_matrix_topk_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _i64, _vp, _vp, _vp]),
    "tspn_traj_iou_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_traj_iou_tail_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_pair_index_i64": (_int, [_i64, _i64, _vp, _vp]),
    "tspn_pair_gather_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp]),
    "tspn_pack_conv3_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv3_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp]),
    "tspn_conv3_tc_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp]),
    "tspn_heads_f32": (_int, [_int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64,
                              _i64, _vp, _vp]),
    "tspn_heads_pairgrid_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "tspn_temporal_mean_f32": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "tspn_temporal_sum_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_pair_rows_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "tspn_transpose_td_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_decode_topk_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "tspn_decode_topk_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                    _vp, _vp, _vp, _vp, _sz, _vp]),
    "tspn_decode_spans_f32": (_int, [_vp, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_float), _i64,
                                     ctypes.c_double, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_forward_fused_workspace_bytes": (_sz, [ctypes.POINTER(FusedDesc)]),
    "tspn_forward_fused_f32": (_int, [ctypes.POINTER(FusedDesc), _vp]),
    "tspn_temporal_encoder_heads_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64,
                                               _vp, _vp, _vp]),
    "tspn_pack_conv2d_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv2d_nhwc_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp,
                                    _int, _vp, _vp]),
    "tspn_pack_conv2d_frag_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv2d_nhwc_frag_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp,
                                         _int, _vp, _vp]),
    "tspn_pack_conv2d_frag_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv2d_nhwc_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp,
                                     _int, _vp, _vp]),
    "tspn_roi_align_nhwc_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, ctypes.c_float, _int, _int,
                                       _int, _vp, _vp]),
    "tspn_pack_conv2d_frag_cin4_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv2d_nhwc_cin4_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _int, _vp, _vp]),
    "tspn_max_pool_nhwc_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "tspn_bottleneck_tail_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_bottleneck_block_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_bottleneck_block_proj_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _vp]),
    "tspn_bottleneck_block_res_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_bottleneck_tail_io_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_bottleneck_tail_pipe_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tspn_bottleneck_tail_next_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tspn_stem_bf16_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "tspn_pack_stem_bf16": (_int, [_vp, _i64, _vp, _vp]),
    "tspn_stem_conv_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp, _vp]),
    "tspn_max_pool_nhwc_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tspn_stem_pool_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _sz, _vp, _vp]),
    "tspn_roi_align_nhwc_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, ctypes.c_float, _int, _int,
                                        _int, _vp, _vp]),
    "tspn_roi_align_nhwc_f32_bf16out": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, ctypes.c_float, _int, _int,
                                               _int, _vp, _vp]),
    "tspn_pack_conv3_wino63_frag_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_conv3_tc_wino63_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "tspn_conv3_tc_wino63_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp, _sz, _vp]),
    "tspn_conv3_tc_wino63_set_piece_form": (_int, [_int]),
    "tspn_span_predicate_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "tspn_span_predicate_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "tspn_cast_bf16": (_int, [_vp, _i64, _vp, _vp]),
    "tspn_pack_conv3_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_transpose_cast_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_heads_dense_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "tspn_temporal_encoder_heads_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "tspn_pack_heads_bf16": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "tspn_conv3_tc_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "tspn_heads_pairgrid_bf16": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "tspn_temporal_mean_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tspn_forward_fused_bf16_workspace_bytes": (_sz, [ctypes.POINTER(FusedBf16Desc)]),
    "tspn_forward_fused_bf16": (_int, [ctypes.POINTER(FusedBf16Desc), _vp]),
}

_lib = None
_lock = threading.Lock()


def header_symbols(path=HEADER_PATH):
    """Function names declared in the public header (used by the export test)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tspn_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the bound library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"TSPN HIP library not built: {LIB_PATH} is missing. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as exc:
                raise RuntimeError(f"TSPN HIP library {LIB_PATH} does not export {name}") from exc
            fn.restype = res
            fn.argtypes = args
        if handle.tspn_version() != ABI_VERSION:
            raise RuntimeError(f"TSPN ABI version mismatch: library {LIB_PATH} reports "
                               f"{handle.tspn_version()}, host expects {ABI_VERSION} (stale build? run "
                               "__graft_entry__.build())")
        for fn, struct in (("tspn_fused_desc_size", FusedDesc), ("tspn_fused_bf16_desc_size", FusedBf16Desc)):
            if getattr(handle, fn)() != ctypes.sizeof(struct):
                raise RuntimeError(f"TSPN ABI mismatch: {fn}() = {getattr(handle, fn)()} but the host binding "
                                   f"has {ctypes.sizeof(struct)} bytes (stale {LIB_PATH}?)")
        _lib = handle
    return _lib


class TspnError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"tspn error {code}: {message}")
        self.code = code


def check(rc):
    """Error-code -> exception mapping (no C++ exception crosses the ABI)."""
    if rc != TSPN_OK:
        l = lib()
        msg = l.tspn_last_error().decode("utf-8", "replace")
        if not msg:
            msg = l.tspn_error_string(rc).decode()
        raise TspnError(rc, msg)
