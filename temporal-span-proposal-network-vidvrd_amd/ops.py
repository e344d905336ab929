"""Tensor-level wrappers over the C ABI (device tensors in, device tensors out).

PyTorch is plumbing here: it owns HBM allocations and the HIP stream; every
operator below runs in the hand-written HIP library.  Inputs must already be
fp32 / int64, contiguous, on a HIP device — anything else raises; there is no
CPU path.
"""
import ctypes
import threading

import torch

from . import _abi

__all__ = [
    "predicate_head", "feature_preprocess_", "ppn_pair_matrix_topk", "traj_iou", "traj_iou_tail", "pair_index",
    "pair_gather", "pack_conv3", "conv3", "conv3_tc", "pack_conv3_wino63", "conv3_tc_wino63", "heads", "heads_pairgrid", "temporal_mean", "temporal_sum", "pair_rows", "transpose_td",
    "forward_fused", "temporal_encoder_heads", "fused_workspace_bytes", "fused_bf16_workspace_bytes", "decode_topk", "decode_spans",
    "cast_bf16", "pack_conv3_bf16", "pack_heads_bf16", "conv3_tc_bf16", "heads_pairgrid_bf16",
    "transpose_cast_bf16", "temporal_encoder_heads_bf16",
    "temporal_mean_bf16", "forward_fused_bf16", "span_predicate", "bottleneck_block_bf16", "bottleneck_block_proj_bf16", "bottleneck_block_res_bf16",
    "proposal_pair_filter", "gather_rows", "wino63_set_piece_form", "conv3_spot_check",
    "pack_conv2d", "pack_conv2d_frag", "conv2d_nhwc", "roi_align_nhwc", "pack_conv2d_frag_bf16", "conv2d_nhwc_bf16", "max_pool_nhwc", "pack_conv2d_frag_cin4", "conv2d_nhwc_cin4", "max_pool_nhwc_bf16", "pack_stem_bf16", "stem_conv_bf16", "stem_pool_bf16", "bottleneck_tail_bf16",
]


_status_blocks = {}          # device index -> numpy view of the pinned int32 [STATUS_WORDS] the kernels of that device can write
_status_lock = threading.Lock()


def _status_block(index=None):
    """The device status block of device `index` (default: the current one), allocated and attached on first use:
    `_abi.STATUS_WORDS` int32 of pinned host memory (include/tspn_mi355x.h, "device status block").  A kernel that
    detects a fault raises it there; the next launch entry of the library then returns TSPN_EDEVICE, which
    `_abi.check` turns into a `TspnError` -- without any synchronisation."""
    idx = torch.cuda.current_device() if index is None else int(index)
    blk = _status_blocks.get(idx)
    if blk is not None:
        return blk
    with _status_lock:
        blk = _status_blocks.get(idx)
        if blk is None:
            raw = torch.zeros(_abi.STATUS_WORDS + 16, dtype=torch.int32).pin_memory()
            off = (-(raw.data_ptr() // 4)) % 16                     # 64-byte aligned
            blk = raw[off:off + _abi.STATUS_WORDS]
            with torch.cuda.device(idx):
                _abi.check(_abi.lib().tspn_status_attach(ctypes.c_void_p(blk.data_ptr())))
            blk = _status_blocks[idx] = blk.numpy()             # live view of the pinned words (it keeps the tensor alive)
    return blk


def _stream():
    """Current stream of the CURRENT device; every public op runs under `_on_tensor_device`, which makes
    the device of its tensor arguments current for the duration of the call.  Every launch passes here: the
    device's status block is attached before its first kernel runs."""
    blk = _status_block()
    # fail fast: with a fault standing on this device nothing more is launched from here (the C entries report it too, but
    # only where they check the launch error, i.e. after enqueueing their kernel)
    fault = int(blk[_abi.STATUS_FAULT])
    if fault:
        raise _abi.TspnError(_abi.TSPN_EDEVICE,
                             f"device {torch.cuda.current_device()} reported fault 0x{fault:x}"
                             + (" (an LDS hand-over between waves timed out)" if fault & _abi.FAULT_HANDOVER else "")
                             + f", info 0x{int(blk[_abi.STATUS_FAULT_INFO]) & 0xffffffff:x}, in an earlier launch of this library: "
                             "results produced since are not to be trusted; ops.status_clear() re-arms")
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def status_words(device=None):
    """Live numpy view of a device's status block (words: _abi.STATUS_*)."""
    return _status_block(None if device is None else torch.device(device).index)


def status_fault(device=None):
    """The fault word of a device (0 = healthy), read from pinned host memory: no synchronisation."""
    return int(status_words(device)[_abi.STATUS_FAULT])


def status_clear(device=None):
    """Re-arm after a fault has been dealt with (results since the fault are not to be trusted)."""
    w = status_words(device)
    w[_abi.STATUS_FAULT_INFO] = 0
    w[_abi.STATUS_FAULT] = 0


def status_selftest(device=None):
    """Raise TSPN_FAULT_HANDOVER on purpose, through the device code of the role-split res4 tail's bounded wait
    (tspn_status_selftest).  Returns a device int32 that stays 0: the wave ends inside the wait."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        reached = torch.zeros(1, dtype=torch.int32, device=dev)
        _status_block()
        _abi.check(_abi.lib().tspn_status_selftest(_p(reached), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return reached


def _first_hip_device(obj, depth=0):
    if isinstance(obj, torch.Tensor):
        return obj.device if obj.is_cuda else None
    if depth < 2:
        if isinstance(obj, (list, tuple)):
            for v in obj:
                d = _first_hip_device(v, depth + 1)
                if d is not None:
                    return d
        elif isinstance(obj, dict):
            for v in obj.values():
                d = _first_hip_device(v, depth + 1)
                if d is not None:
                    return d
    return None


def _on_tensor_device(fn):
    """The C ABI launches on the stream it is handed and never calls hipSetDevice; kernels, the
    per-device LDS limits and the workspace allocations of an op must all belong to the device that
    holds its operands.  This wrapper makes that device current (a no-op when it already is) and
    refuses operands spread over several devices."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            d = _first_hip_device(a)
            if d is None:
                continue
            if dev is None:
                dev = d
            elif d != dev:
                raise RuntimeError(f"{fn.__name__}: operands live on different devices ({dev} and {d})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the TSPN HIP path needs a HIP device "
                           "tensor (there is no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _ws(nbytes, device):
    # torch's caching allocator returns >= 256-B aligned blocks
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def predicate_head(x, weight, bias, apply_sigmoid=True, norm=None):
    """sigmoid(x @ W.T + b) — RelationPredictor.forward (reference lib/modeling/model.py:85-88).

    `norm=(first, block, nblocks)`: x holds RAW features and the block-L1 normalisation of
    VRDataset._feature_preprocess (lib/dataset/vrdataset.py:219-243) is folded into the GEMM."""
    _dev(x, "x"); _dev(weight, "weight")
    if bias is not None:
        _dev(bias, "bias")
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"predicate_head: shapes {tuple(x.shape)} x {tuple(weight.shape)}^T do not match")
    if bias is not None and bias.shape != (weight.shape[0],):
        raise ValueError("predicate_head: bias shape mismatch")
    P, F = x.shape
    K = weight.shape[0]
    out = torch.empty((P, K), dtype=torch.float32, device=x.device)
    l = _abi.lib()
    if norm is not None:
        first, block, nblocks = (int(v) for v in norm)
        if first < 0 or block <= 0 or nblocks < 0 or first + block * nblocks > F:
            raise ValueError(f"predicate_head: normalisation blocks {norm} exceed F={F}")
        ws = _ws(l.tspn_predicate_head_norm_workspace_bytes(P, F, K, first, block, nblocks), x.device)
        _abi.check(l.tspn_predicate_head_norm_f32(_p(x), P, F, F, _p(weight), _p(bias), K, first, block,
                                                  nblocks, _p(out), 1 if apply_sigmoid else 0, _p(ws),
                                                  ws.numel(), _stream()))
        return out
    nbytes = l.tspn_predicate_head_workspace_bytes(P, F, K)
    ws = _ws(nbytes, x.device)
    _abi.check(l.tspn_predicate_head_f32(_p(x), P, F, F, _p(weight), _p(bias), K, _p(out),
                                         1 if apply_sigmoid else 0, _p(ws), ws.numel(), _stream()))
    return out


def feature_preprocess_(feats, first=70, block=1000, nblocks=8):
    """In-place block-L1 normalisation — VRDataset._feature_preprocess (lib/dataset/vrdataset.py:219-243)."""
    _dev(feats, "feats")
    if feats.dim() != 2:
        raise ValueError("feature_preprocess_: feats must be [P,F]")
    P, F = feats.shape
    _abi.check(_abi.lib().tspn_feature_preprocess_f32(_p(feats), P, F, F, first, block, nblocks, _stream()))
    return feats


def proposal_pair_filter(pairs, trackid, pair_off=None, track_off=None):
    """VRDataset._get_proposal_idx + _get_num_tracklet_proposals (lib/dataset/vrdataset.py:140-148).

    One segment: pairs int64 [P,2] (track indices as stored in the -relation.h5 file), trackid int64 [M]
    (-1 = proposal, >= 0 = ground truth) -> (proposal_idx int64 [P'], num_tracks int).
    Several segments stored back to back: pass int64 offsets pair_off [S+1], track_off [S+1] (device
    tensors) -> (idx int64 [P_total] with segment s's kept local indices at idx[pair_off[s]:pair_off[s]
    + count[s]], count int64 [S], num_tracks int64 [S]) as device tensors (no host sync).
    A pair naming a track outside its segment raises IndexError (single-segment form) or yields
    count -1 (batched form)."""
    _dev(pairs, "pairs", torch.int64); _dev(trackid, "trackid", torch.int64)
    if pairs.dim() != 2 or pairs.shape[1] != 2 or trackid.dim() != 1:
        raise ValueError("proposal_pair_filter: pairs must be [P,2] and trackid [M]")
    dev = pairs.device
    single = pair_off is None
    if single:
        pair_off = torch.tensor([0, pairs.shape[0]], dtype=torch.int64, device=dev)
        track_off = torch.tensor([0, trackid.shape[0]], dtype=torch.int64, device=dev)
    _dev(pair_off, "pair_off", torch.int64); _dev(track_off, "track_off", torch.int64)
    if pair_off.shape != track_off.shape or pair_off.dim() != 1 or pair_off.numel() < 1:
        raise ValueError("proposal_pair_filter: pair_off / track_off must be int64 [S+1]")
    S = pair_off.numel() - 1
    idx = torch.empty((pairs.shape[0],), dtype=torch.int64, device=dev)
    count = torch.empty((S,), dtype=torch.int64, device=dev)
    ntr = torch.empty((S,), dtype=torch.int64, device=dev)
    _abi.check(_abi.lib().tspn_proposal_pair_filter_i64(_p(pairs), _p(pair_off), _p(trackid), _p(track_off), S,
                                                        _p(idx), _p(count), _p(ntr), _stream()))
    if not single:
        return idx, count, ntr
    c = int(count[0])
    if c < 0:
        raise IndexError("proposal_pair_filter: a pair names a track index outside trackid")
    return idx[:c], int(ntr[0])


def gather_rows(src, idx, check_idx=True):
    """out[r] = src[idx[r]] for a 2-D fp32 matrix (feats[proposal_idx], lib/dataset/vrdataset.py:66-67).
    `check_idx=False` skips the range check (two host syncs) for indices a kernel of this library produced
    (proposal_pair_filter emits range-checked local row numbers)."""
    _dev(src, "src"); _dev(idx, "idx", torch.int64)
    if src.dim() != 2 or idx.dim() != 1:
        raise ValueError("gather_rows: src must be [R,F] and idx [R']")
    if check_idx and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= src.shape[0]):
        raise IndexError("gather_rows: row index out of range")
    out = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    if src.shape[1]:
        _abi.check(_abi.lib().tspn_gather_rows_f32(_p(src), src.shape[1], src.shape[1], _p(idx), 0, idx.numel(),
                                                   _p(out), _stream()))
    return out


def ppn_pair_matrix_topk(cls_logits, w, topk):
    """PPNHead + top-k (lib/modeling/relpn/ppn.py:107-112, 84-85).

    cls_logits [B,N,Cin] or [N,Cin]; `w` = dict with keys sub_emb.0.weight ... obj_emb.2.bias.
    Returns (pair_matrix [B,N,N], idx int64 [B,min(topk,N*N)]) (batch dim dropped for 2-D input).
    """
    squeeze = cls_logits.dim() == 2
    c = cls_logits.unsqueeze(0) if squeeze else cls_logits
    _dev(c, "cls_logits")
    B, N, Cin = c.shape
    ks = ["sub_emb.0.weight", "sub_emb.0.bias", "sub_emb.2.weight", "sub_emb.2.bias",
          "obj_emb.0.weight", "obj_emb.0.bias", "obj_emb.2.weight", "obj_emb.2.bias"]
    ts = [_dev(w[k], k) for k in ks]
    H, Cout = ts[0].shape[0], ts[2].shape[0]
    if ts[0].shape != (H, Cin) or ts[2].shape != (Cout, H) or ts[4].shape != (H, Cin) or ts[6].shape != (Cout, H):
        raise ValueError("ppn_pair_matrix_topk: weight shapes do not match the input")
    k = min(int(topk), N * N)
    mat = torch.empty((B, N, N), dtype=torch.float32, device=c.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=c.device)
    _abi.check(_abi.lib().tspn_ppn_pair_matrix_topk_f32(
        _p(c), B, N, Cin, H, Cout, *[_p(t) for t in ts], k, _p(mat), _p(idx), _stream()))
    return (mat[0], idx[0]) if squeeze else (mat, idx)


def traj_iou(boxes1, boxes2=None):
    """cubic_iou (lib/modeling/trajectory.py:127-141); boxes [N,T,4] or [B,N,T,4] -> [.., N1, N2]."""
    squeeze = boxes1.dim() == 3
    b1 = boxes1.unsqueeze(0) if squeeze else boxes1
    _dev(b1, "boxes1")
    B, N1, T, four = b1.shape
    if four != 4:
        raise ValueError("traj_iou: last dim must be 4")
    if boxes2 is None:
        b2, N2 = None, N1
    else:
        b2 = boxes2.unsqueeze(0) if squeeze else boxes2
        _dev(b2, "boxes2")
        if b2.shape[0] != B or b2.shape[2] != T or b2.shape[3] != 4:
            raise ValueError("traj_iou: boxes2 shape mismatch")
        N2 = b2.shape[1]
    out = torch.empty((B, N1, N2), dtype=torch.float32, device=b1.device)
    _abi.check(_abi.lib().tspn_traj_iou_f32(_p(b1), N1, _p(b2), N2, B, T, _p(out), _stream()))
    return out[0] if squeeze else out


@_on_tensor_device
def traj_iou_tail(a, len_a, b, out=None):
    """Association-time IoU, one launch per segment (replaces the per-candidate `_traj_iou` calls of reference
    lib/modeling/association.py:35-48,101-106): a [U,L,4] float64 = trajectories from the segment's first frame on,
    len_a [U] int32 = their number of common frames (<= L; 0 -> IoU 0), b [N,L,4] float64 = the segment's tracklets
    -> [U,N] float32 with the reference chain's roundings (tspn_traj_iou_tail_f64)."""
    _dev(a, "a", torch.float64), _dev(b, "b", torch.float64), _dev(len_a, "len_a", torch.int32)
    if a.dim() != 3 or b.dim() != 3 or a.shape[2] != 4 or b.shape[2] != 4 or a.shape[1] != b.shape[1]:
        raise ValueError(f"traj_iou_tail: a [U,L,4] and b [N,L,4] expected, got {tuple(a.shape)} and {tuple(b.shape)}")
    U, L, _ = a.shape
    N = b.shape[0]
    if len_a.shape != (U,):
        raise ValueError(f"traj_iou_tail: len_a must be [{U}], got {tuple(len_a.shape)}")
    if not (a.is_contiguous() and b.is_contiguous() and len_a.is_contiguous()):
        raise ValueError("traj_iou_tail: operands must be contiguous")
    if out is None:
        out = torch.empty((U, N), dtype=torch.float32, device=a.device)
    elif out.shape != (U, N) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != a.device:
        raise ValueError("traj_iou_tail: out must be a contiguous float32 [U,N] tensor on the operands' device")
    _abi.check(_abi.lib().tspn_traj_iou_tail_f64(_p(a), _p(len_a), _p(b), U, N, L, _p(out), _stream()))
    return out


def pair_index(n, device, base=0):
    """All ordered pairs (i,j), i != j, i-major (lib/modeling/predict.py:133-140), int64 [n(n-1), 2]."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("pair_index: needs a HIP device (no CPU fallback)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(dev):
        out = torch.empty((max(n * (n - 1), 0), 2), dtype=torch.int64, device=dev)
        _abi.check(_abi.lib().tspn_pair_index_i64(n, base, _p(out), _stream()))
    return out


def pair_gather(tracklet_feats, tracklet_boxes, pairs, want_feat=True, want_geom=True, check_pairs=True, out_geom=None):
    """N^2 pair builder: -> (pair_feats [P,2D,T] | None, pair_geom [P,8,T] | None).  `out_geom`: caller-held
    contiguous fp32 [P,8,T] the geometry is written into (e.g. a slice of a buffer another stream reads later)."""
    _dev(pairs, "pairs", torch.int64)
    P = pairs.shape[0]
    if pairs.dim() != 2 or pairs.shape[1] != 2:
        raise ValueError("pair_gather: pairs must be [P,2]")
    feat = geom = None
    NT = T = D = None
    if want_feat:
        _dev(tracklet_feats, "tracklet_feats")
        NT, T, D = tracklet_feats.shape
        feat = torch.empty((P, 2 * D, T), dtype=torch.float32, device=pairs.device)
    if want_geom:
        _dev(tracklet_boxes, "tracklet_boxes")
        if tracklet_boxes.dim() != 3 or tracklet_boxes.shape[2] != 4:
            raise ValueError("pair_gather: tracklet_boxes must be [N,T,4]")
        if NT is not None and tuple(tracklet_boxes.shape[:2]) != (NT, T):
            raise ValueError("pair_gather: feats / boxes shape mismatch")
        NT, T = tracklet_boxes.shape[:2]
        if out_geom is None:
            geom = torch.empty((P, _abi.GEOM_CHANNELS, T), dtype=torch.float32, device=pairs.device)
        else:
            _dev(out_geom, "out_geom")
            if tuple(out_geom.shape) != (P, _abi.GEOM_CHANNELS, T):
                raise ValueError(f"pair_gather: out_geom must be [{P},{_abi.GEOM_CHANNELS},{T}], got {tuple(out_geom.shape)}")
            geom = out_geom
        D = D or 1
    if NT is None:
        return None, None
    if check_pairs and P and (int(pairs.min()) < 0 or int(pairs.max()) >= NT):   # host sync: skip for tables built here
        raise IndexError("pair_gather: pair index out of range")
    _abi.check(_abi.lib().tspn_pair_gather_f32(
        _p(tracklet_feats if want_feat else None), _p(tracklet_boxes if want_geom else None),
        NT, T, D, _p(pairs), P, _p(feat), _p(geom), _stream()))
    return feat, geom


def pack_conv3(weight, split=0):
    """nn.Conv1d weight [M,Cin,3] -> MFMA staging layout [3][Cin'][M'] (see tspn_pack_conv3_f32)."""
    _dev(weight, "conv weight")
    if weight.dim() != 3 or weight.shape[2] != 3:
        raise ValueError("pack_conv3: weight must be [M,Cin,3]")
    M, Cin, _ = weight.shape
    if split:
        shape = (3, split, 2 * M)
    else:
        shape = (3, Cin, M)
    packed = torch.empty(shape, dtype=torch.float32, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv3_f32(_p(weight), M, Cin, split, _p(packed), _stream()))
    return packed


def conv3(x, packed, bias=None, relu=False):
    """y[B,M,T] = act(bias + conv1d_k3(x[B,Cin,T])) with packed weights [3,Cin,M]."""
    _dev(x, "x"); _dev(packed, "packed")
    if bias is not None:
        _dev(bias, "bias")
    B, Cin, T = x.shape
    if packed.dim() != 3 or packed.shape[0] != 3 or packed.shape[1] != Cin:
        raise ValueError(f"conv3: packed weights {tuple(packed.shape)} do not match Cin={Cin}")
    M = packed.shape[2]
    if bias is not None and bias.shape != (M,):
        raise ValueError("conv3: bias shape mismatch")
    y = torch.empty((B, M, T), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_conv3_f32(_p(x), B, Cin, T, _p(packed), M, _p(bias), 1 if relu else 0,
                                         _p(y), _stream()))
    return y


def conv3_tc(x, packed, bias=None, relu=False):
    """conv3 on channels-last x[B,T,Cin] (tracklet layout) -> y[B,M,T]; fast path only."""
    _dev(x, "x"); _dev(packed, "packed")
    if bias is not None:
        _dev(bias, "bias")
    B, T, Cin = x.shape
    if packed.dim() != 3 or packed.shape[0] != 3 or packed.shape[1] != Cin:
        raise ValueError(f"conv3_tc: packed weights {tuple(packed.shape)} do not match Cin={Cin}")
    M = packed.shape[2]
    y = torch.empty((B, M, T), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_conv3_tc_f32(_p(x), B, T, Cin, _p(packed), M, _p(bias), 1 if relu else 0,
                                            _p(y), _stream()))
    return y


def wino63_frag_dims(frag):
    """(Cin, M) of a fragment-major F(6,3) weight tensor (pack_conv3_wino63)."""
    if frag.dim() != 5 or tuple(frag.shape[2:]) != (8, 64, 4):
        raise ValueError(f"not a fragment-major F(6,3) weight tensor: {tuple(frag.shape)}")
    return frag.shape[1] * 8, frag.shape[0] * 32


def pack_conv3_wino63(weight, split=0):
    """nn.Conv1d weight [M,Cin,3] -> Winograd F(6,3) weights, fragment-major [M'/32][Cin'/8][8][64][4]
    (tspn_pack_conv3_wino63_frag_f32; split as in pack_conv3).  Needs Cin' % 8 == 0, M' % 32 == 0."""
    _dev(weight, "conv weight")
    if weight.dim() != 3 or weight.shape[2] != 3:
        raise ValueError("pack_conv3_wino63: weight must be [M,Cin,3]")
    M, Cin, _ = weight.shape
    Mp, Cp = (2 * M, split) if split else (M, Cin)
    if Cp % 8 or Mp % 32:
        raise ValueError(f"pack_conv3_wino63: needs Cin % 8 == 0 and M % 32 == 0 (Cin={Cp}, M={Mp})")
    frag = torch.empty((Mp // 32, Cp // 8, 8, 64, 4), dtype=torch.float32, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv3_wino63_frag_f32(_p(weight), M, Cin, split, _p(frag), _stream()))
    return frag


def conv3_tc_wino63(x, frag, bias=None, relu=False, workspace=None):
    """Winograd F(6,3) conv3 (tspn_wino63.hip): channels-last x[B,T,Cin] with pack_conv3_wino63 weights -> y[B,M,T].
    Needs Cin % 32 == 0; `workspace` (uint8, >= tspn_conv3_tc_wino63_workspace_bytes) holds the transformed input."""
    _dev(x, "x"); _dev(frag, "frag")
    if bias is not None:
        _dev(bias, "bias")
    B, T, Cin = x.shape
    cin_w, M = wino63_frag_dims(frag)
    if cin_w != Cin:
        raise ValueError(f"conv3_tc_wino63: weights are for Cin={cin_w}, x has Cin={Cin}")
    l = _abi.lib()
    need = l.tspn_conv3_tc_wino63_workspace_bytes(B, T, Cin)
    if workspace is None:
        workspace = _ws(need, x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"conv3_tc_wino63: workspace too small ({workspace.numel()} < {need})")
    y = torch.empty((B, M, T), dtype=torch.float32, device=x.device)
    _abi.check(l.tspn_conv3_tc_wino63_f32(_p(x), B, T, Cin, _p(frag), M, _p(bias), 1 if relu else 0, _p(y),
                                          _p(workspace), workspace.numel() * workspace.element_size(), _stream()))
    return y


def wino63_set_piece_form(form):
    """0 = buffer-load pieces where the workspace is below 4 GB (default), 1 = 64-bit pointer pieces everywhere
    (tspn_conv3_tc_wino63_set_piece_form).  Returns the previous setting."""
    prev = _abi.lib().tspn_conv3_tc_wino63_set_piece_form(int(form))
    if prev < 0:
        _abi.check(prev)
    return prev


def heads(a, head_w, head_b, b=None, ia=None, ib=None, bias=None, channels=None, num_pairs=None):
    """out[P,H,T] = head_b + head_w @ h_p; h_p = a[ia[p]] (dense) or relu(a[ia[p]] + b[ib[p]] + bias)."""
    _dev(a, "a"); _dev(head_w, "head_w")
    mode = 0 if b is None else 1
    lda, T = a.shape[1], a.shape[2]
    C = channels if channels is not None else lda
    H = head_w.shape[0]
    if head_w.shape[1] != C:
        raise ValueError("heads: head_w / channel mismatch")
    for nm, t in (("ia", ia), ("ib", ib)):
        if t is not None:
            _dev(t, nm, torch.int64)
    if b is not None:
        _dev(b, "b")
        if b.shape[1:] != a.shape[1:]:
            raise ValueError("heads: a / b shape mismatch")
    if num_pairs is None:
        num_pairs = ia.shape[0] if ia is not None else a.shape[0]
    out = torch.empty((num_pairs, H, T), dtype=torch.float32, device=a.device)
    _abi.check(_abi.lib().tspn_heads_f32(mode, _p(a), _p(b), lda, _p(ia), _p(ib), 1, _p(bias),
                                         _p(head_w), _p(head_b), H, num_pairs, C, T, _p(out), _stream()))
    return out


def heads_pairgrid(y, B, N, head_w, head_b):
    """Blocked pair stage on tracklet projections y[B*N, 2C, T] for the canonical pair table."""
    _dev(y, "y"); _dev(head_w, "head_w")
    NT, C2, T = y.shape
    C, H = C2 // 2, head_w.shape[0]
    if NT != B * N or head_w.shape[1] != C:
        raise ValueError("heads_pairgrid: shape mismatch")
    out = torch.empty((B * N * max(N - 1, 0), H, T), dtype=torch.float32, device=y.device)
    _abi.check(_abi.lib().tspn_heads_pairgrid_f32(_p(y), B, N, C, T, _p(head_w), _p(head_b), H, _p(out),
                                                  _stream()))
    return out


def temporal_mean(x, layout_tc):
    """Mean over frames: x[R,T,D] -> [R,D] (layout_tc=True) or x[R,C,T] -> [R,C]."""
    _dev(x, "x")
    if layout_tc:
        R, T, Cd = x.shape
    else:
        R, Cd, T = x.shape
    out = torch.empty((R, Cd), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_temporal_mean_f32(_p(x), R, T, Cd, 1 if layout_tc else 0, _p(out), _stream()))
    return out


def temporal_sum(x):
    """Sum over the middle axis: x[R,T,D] -> [R,D], frames added in order (tspn_temporal_sum_f32); with R = 1 the
    column sums of a matrix (bias gradients of the training step)."""
    _dev(x, "x")
    R, T, Cd = x.shape
    out = torch.empty((R, Cd), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_temporal_sum_f32(_p(x), R, T, Cd, _p(out), _stream()))
    return out


def pair_rows(src, pairs):
    """out[P,2D] = cat(src[pairs[:,0]], src[pairs[:,1]])."""
    _dev(src, "src"); _dev(pairs, "pairs", torch.int64)
    NT, D = src.shape
    P = pairs.shape[0]
    out = torch.empty((P, 2 * D), dtype=torch.float32, device=src.device)
    _abi.check(_abi.lib().tspn_pair_rows_f32(_p(src), NT, D, _p(pairs), P, _p(out), _stream()))
    return out


def transpose_td(x):
    """[R,T,D] -> [R,D,T]."""
    _dev(x, "x")
    R, T, D = x.shape
    out = torch.empty((R, D, T), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_transpose_td_f32(_p(x), R, T, D, _p(out), _stream()))
    return out


def decode_topk(rel_logit, pairs, cls_sub, cls_obj=None, row_mul=1, num_obj=35,
                topk_per_pair=20, topk_per_seg=200, check_pairs=True):
    """Top-k triplet decode (lib/modeling/predict.py:66-106) for a batch of equal-shape segments.

    rel_logit [S,P,K] (or [P,K]); pairs int64 [S,P,2] local tracklet ids; class logits:
      * reference quirk: cls_sub = feature matrix [S,P,F] with F >= 70 (columns 0:35 subject,
        35:70 object classeme), row_mul = N-1  (predict.py:88-89 reads row (N-1)*tid);
      * per-tracklet:    cls_sub = track_cls_logits [S,N,35], row_mul = 1.
    Returns (scores [S,M], triplets int64 [S,M,3], pair_tids int64 [S,M,2]).
    """
    squeeze = rel_logit.dim() == 2
    if squeeze:
        rel_logit, pairs, cls_sub = rel_logit.unsqueeze(0), pairs.unsqueeze(0), cls_sub.unsqueeze(0)
        cls_obj = cls_obj.unsqueeze(0) if cls_obj is not None else None
    _dev(rel_logit, "rel_logit"); _dev(pairs, "pairs", torch.int64); _dev(cls_sub, "cls_sub")
    S, P, K = rel_logit.shape
    if tuple(pairs.shape) != (S, P, 2):
        raise ValueError("decode_topk: pairs must be [S,P,2]")
    if cls_sub.dim() != 3 or cls_sub.shape[0] != S:
        raise ValueError("decode_topk: class logits must be [S,rows,cols]")
    seg_rows, ld = cls_sub.shape[1], cls_sub.shape[2]
    if cls_obj is None:
        if ld >= 2 * num_obj:      # feature matrix: object classeme in columns [NO, 2NO)
            obj_ptr = cls_sub.data_ptr() + 4 * num_obj
        else:                      # per-tracklet logits: same table for both roles
            obj_ptr = cls_sub.data_ptr()
    else:
        _dev(cls_obj, "cls_obj")
        if cls_obj.shape != cls_sub.shape:
            raise ValueError("decode_topk: cls_obj shape mismatch")
        obj_ptr = cls_obj.data_ptr()
    if check_pairs and P and (int(pairs.min()) < 0 or int(pairs.max()) * row_mul >= seg_rows):
        raise IndexError("decode_topk: row_mul * tracklet id exceeds the class-logit rows")
    R = min(topk_per_pair, K)
    M = min(topk_per_seg, P * R)
    dev = rel_logit.device
    scores = torch.empty((S, M), dtype=torch.float32, device=dev)
    trip = torch.empty((S, M, 3), dtype=torch.int64, device=dev)
    tids = torch.empty((S, M, 2), dtype=torch.int64, device=dev)
    l = _abi.lib()
    ws = _ws(l.tspn_decode_topk_workspace_bytes(S, P, R), dev)
    _abi.check(l.tspn_decode_topk_f32(_p(rel_logit), _p(pairs), _p(cls_sub), ctypes.c_void_p(obj_ptr), ld,
                                      seg_rows, row_mul, S, P, K, num_obj, topk_per_pair, topk_per_seg,
                                      _p(scores), _p(trip), _p(tids), _p(ws), ws.numel(), _stream()))
    return (scores[0], trip[0], tids[0]) if squeeze else (scores, trip, tids)


def decode_spans(heads, sizes, top_k=64, nms_threshold=0.5, pre_nms=1024):
    """Temporal span proposals per pair from heads [P,3A,T] (span decode + 1-D NMS).

    Returns dict(anchor int64 [P,top_k], span int64 [P,top_k,2], span_f fp32 [P,top_k,2],
    score fp32 [P,top_k], count int64 [P])."""
    _dev(heads, "heads")
    P, H, T = heads.shape
    A = len(sizes)
    if H != 3 * A:
        raise ValueError(f"decode_spans: heads has {H} channels, expected 3*A = {3 * A}")
    dev = heads.device
    out = {"anchor": torch.empty((P, top_k), dtype=torch.int64, device=dev),
           "span": torch.empty((P, top_k, 2), dtype=torch.int64, device=dev),
           "span_f": torch.empty((P, top_k, 2), dtype=torch.float32, device=dev),
           "score": torch.empty((P, top_k), dtype=torch.float32, device=dev),
           "count": torch.empty((P,), dtype=torch.int64, device=dev)}
    sz = (ctypes.c_float * A)(*[float(v) for v in sizes])
    _abi.check(_abi.lib().tspn_decode_spans_f32(_p(heads), P, A, T, sz, top_k, float(nms_threshold), pre_nms,
                                                _p(out["anchor"]), _p(out["span"]), _p(out["span_f"]),
                                                _p(out["score"]), _p(out["count"]), _stream()))
    return out


def _fused_desc(feats, pairs, B, N, conv_packed, conv_bias, head_w, head_b, cls_w, cls_b):
    _dev(feats, "tracklet_feats"); _dev(pairs, "pairs", torch.int64)
    for nm, t in (("conv_packed", conv_packed), ("conv_bias", conv_bias), ("head_w", head_w),
                  ("head_b", head_b), ("cls_w", cls_w), ("cls_b", cls_b)):
        _dev(t, nm)
    NT, T, D = feats.shape
    if NT != B * N:
        raise ValueError(f"forward_fused: feats has {NT} tracklets, expected B*N = {B * N}")
    C = 2 * D
    H = head_w.shape[0]
    if H % 3 or head_w.shape[1] != C or head_b.shape != (H,):
        raise ValueError("forward_fused: head_w must be [3A, 2D]")
    w63 = conv_packed.dim() == 5     # fragment-major Winograd F(6,3) weights (pack_conv3_wino63); else direct taps
    if w63 and wino63_frag_dims(conv_packed) != (D, 2 * C):
        raise ValueError(f"forward_fused: Winograd F(6,3) conv weights are for (Cin, M) = "
                         f"{wino63_frag_dims(conv_packed)}, expected ({D}, {2 * C})")
    if (not w63 and tuple(conv_packed.shape) != (3, D, 2 * C)) or conv_bias.shape != (C,):
        raise ValueError(f"forward_fused: conv_packed must be pack_conv3(conv.weight, split=D) = [3, D={D}, 4D={2 * C}] "
                         "or pack_conv3_wino63(conv.weight, split=D)")
    if cls_w.dim() != 2 or cls_w.shape[1] != C or cls_b.shape != (cls_w.shape[0],):
        raise ValueError("forward_fused: cls_w must be [K, 2D]")
    d = _abi.FusedDesc()
    d.B, d.N, d.T, d.D = B, N, T, D
    d.A, d.K = H // 3, cls_w.shape[0]
    d.feats, d.pairs, d.P = feats.data_ptr(), pairs.data_ptr(), pairs.shape[0]
    d.conv_packed, d.conv_bias = conv_packed.data_ptr(), conv_bias.data_ptr()
    d.conv_algo = _abi.CONV_WINOGRAD63 if w63 else _abi.CONV_DIRECT   # the packing is the choice of kernel
    d.head_w, d.head_b = head_w.data_ptr(), head_b.data_ptr()
    d.cls_w, d.cls_b = cls_w.data_ptr(), cls_b.data_ptr()
    return d


def fused_workspace_bytes(B, N, T, D, A, K, P):
    d = _abi.FusedDesc()
    d.B, d.N, d.T, d.D, d.A, d.K, d.P = B, N, T, D, A, K, P
    return _abi.lib().tspn_forward_fused_workspace_bytes(ctypes.byref(d))


def forward_fused(feats, pairs, B, N, conv_packed, conv_bias, head_w, head_b, cls_w, cls_b,
                  workspace=None, out_heads=None, out_logits=None, check_pairs=True,
                  conv_events=None, canonical_pairs=False, logits_event=None, conv_weight=None, conv_check=0):
    """Whole scoring pass on tracklet tensors (tspn_forward_fused_f32).

    `conv_weight` (raw conv.weight [C,C,3]) + `conv_check` (rows): the a-posteriori accuracy guard of the F(6,3)
    conv -- that many output rows are recomputed in float64 behind the conv and the largest deviation lands in the
    device's status block (status_words()[_abi.STATUS_CONV_ERR], float bits); ignored with direct-tap weights.

    feats [B*N,T,D]; pairs int64 [P,2] global tracklet ids.
    `canonical_pairs`: `pairs` is cat_b(pair_index(N, base=b*N)) — the caller built it with
    pair_index — which lets the pair stage use the blocked kernel with computed output slots.
    `conv_events`: optional (begin, end) torch.cuda.Event pair (enable_timing=True, already
    recorded once so the handles exist) re-recorded around the dominant kernel.
    `logits_event`: optional torch.cuda.Event (already recorded once) re-recorded as soon as `rel_logits` is
    complete — the logits are computed first, so decode / PPN / gather can run on another stream behind it.
    Returns (heads [P,3A,T], rel_logits [P,K]).
    """
    d = _fused_desc(feats, pairs, B, N, conv_packed, conv_bias, head_w, head_b, cls_w, cls_b)
    P, T = pairs.shape[0], feats.shape[1]
    if check_pairs and P and (int(pairs.min()) < 0 or int(pairs.max()) >= B * N):
        raise IndexError("forward_fused: pair index out of range")
    l = _abi.lib()
    need = l.tspn_forward_fused_workspace_bytes(ctypes.byref(d))
    if workspace is None:
        workspace = _ws(need, feats.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"forward_fused: workspace too small ({workspace.numel()} < {need})")
    if out_heads is None:
        out_heads = torch.empty((P, 3 * d.A, T), dtype=torch.float32, device=feats.device)
    if out_logits is None:
        out_logits = torch.empty((P, d.K), dtype=torch.float32, device=feats.device)
    d.out_heads, d.out_logits = out_heads.data_ptr(), out_logits.data_ptr()
    d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    d.canonical_pairs = 1 if canonical_pairs else 0
    if canonical_pairs and P != B * N * (N - 1):
        raise ValueError("forward_fused: canonical_pairs needs P == B*N*(N-1)")
    if conv_events is not None:
        d.ev_conv_begin, d.ev_conv_end = conv_events[0].cuda_event, conv_events[1].cuda_event
    if logits_event is not None:
        d.ev_logits_ready = logits_event.cuda_event
    if conv_weight is not None and conv_check > 0 and d.conv_algo == _abi.CONV_WINOGRAD63:
        _dev(conv_weight, "conv_weight")
        if tuple(conv_weight.shape) != (2 * d.D, 2 * d.D, 3):
            raise ValueError(f"forward_fused: conv_weight must be conv.weight [C, C, 3] with C = {2 * d.D}")
        d.conv_weight, d.conv_check = conv_weight.data_ptr(), int(conv_check)
    _abi.check(l.tspn_forward_fused_f32(ctypes.byref(d), _stream()))
    return out_heads, out_logits


def conv3_spot_check(x, weight, y, split=0, bias=None, relu=False, hot=None, rows=128, ldy=None):
    """A-posteriori accuracy check of a temporal conv launch (tspn_conv3_spot_check_f32): x [B,T,Cin], raw
    weight [M,Cw,3] (`split` as in pack_conv3), y = the launch's output [B, rows of y, ldy]; `rows` output rows are
    recomputed at up to 24 columns each in float64 (`hot`: optional int64 [1] naming a sextet every row must cover, as the
    F(6,3) input transform reports it).  The largest |y - y_ref| is max-ed into the device's status
    block (status_words()[_abi.STATUS_CONV_ERR] holds its float bits)."""
    _dev(x, "x"); _dev(weight, "weight"); _dev(y, "y")
    B, T, Cin = x.shape
    M, Cw = weight.shape[0], weight.shape[1]
    if bias is not None:
        _dev(bias, "bias")
    scratch = torch.zeros(_abi.CONV_CHECK_SCRATCH_BYTES // 8, dtype=torch.int64, device=x.device)
    if hot is not None:
        _dev(hot, "hot", torch.int64)
        scratch[_abi.CONV_CHECK_HOT_OFFSET // 8:_abi.CONV_CHECK_HOT_OFFSET // 8 + 1] = hot.reshape(-1)[0:1]
    ld = int(ldy) if ldy is not None else y.shape[-1]
    _abi.check(_abi.lib().tspn_conv3_spot_check_f32(_p(x), B, T, Cin, _p(weight), M, Cw, split, _p(bias), 1 if relu else 0,
                                                    _p(y), ld, _p(scratch), rows, _stream()))


def temporal_encoder_heads(x, conv_packed, conv_bias, head_w, head_b, h_ws=None):
    """DPNHead.forward on a materialised x[P,C,T] (lib/modeling/relpn/dpn.py:69-73) -> [P,H,T]."""
    _dev(x, "x"); _dev(conv_packed, "conv_packed"); _dev(head_w, "head_w")
    P, C, T = x.shape
    if tuple(conv_packed.shape) != (3, C, C):
        raise ValueError("temporal_encoder_heads: conv_packed must be [3,C,C]")
    H = head_w.shape[0]
    if h_ws is None:
        h_ws = torch.empty_like(x)
    out = torch.empty((P, H, T), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_temporal_encoder_heads_f32(
        _p(x), P, C, T, _p(conv_packed), _p(conv_bias), _p(head_w), _p(head_b), H, _p(h_ws), _p(out),
        _stream()))
    return out


# --------------------------------------------------------------------------- #
# bf16-operand path (BASELINE config 3); semantics in csrc/tspn_bf16.hip / oracle.forward_bf16
# --------------------------------------------------------------------------- #
def cast_bf16(x):
    """fp32 -> bf16, round to nearest even (tspn_cast_bf16)."""
    _dev(x, "x")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _abi.check(_abi.lib().tspn_cast_bf16(_p(x), x.numel(), _p(out), _stream()))
    return out


def pack_conv3_bf16(weight, split=0):
    """conv.weight [M,Cin,3] fp32 -> bf16 [3, Cp/8, Mp, 8] (split as in pack_conv3)."""
    _dev(weight, "weight")
    if weight.dim() != 3 or weight.shape[2] != 3:
        raise ValueError("pack_conv3_bf16: weight must be [M, Cin, 3]")
    M, Cin, _ = weight.shape
    Mp, Cp = (2 * M, split) if split else (M, Cin)
    out = torch.empty((3, Cp // 8, Mp, 8), dtype=torch.bfloat16, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv3_bf16(_p(weight), M, Cin, split, _p(out), _stream()))
    return out


def pack_heads_bf16(head_w):
    """[H<=16, C] fp32 -> bf16 [C/8, 16, 8] (zero rows above H)."""
    _dev(head_w, "head_w")
    H, C = head_w.shape
    out = torch.empty((C // 8, 16, 8), dtype=torch.bfloat16, device=head_w.device)
    _abi.check(_abi.lib().tspn_pack_heads_bf16(_p(head_w), H, C, _p(out), _stream()))
    return out


def conv3_tc_bf16(x, packed, bias=None, ldm=None):
    """k=3 conv over time, bf16 operands: x bf16 [B,T,Cin] -> y fp32 channels-last [B,T,M]
    (`ldm` > M: rows padded to ldm floats, returned as [B,T,ldm] with the pad uninitialised)."""
    _dev(x, "x", torch.bfloat16); _dev(packed, "packed", torch.bfloat16)
    if bias is not None:
        _dev(bias, "bias")
    B, T, Cin = x.shape
    if packed.dim() != 4 or packed.shape[0] != 3 or packed.shape[1] * 8 != Cin or packed.shape[3] != 8:
        raise ValueError(f"conv3_tc_bf16: packed {tuple(packed.shape)} does not match Cin={Cin}")
    M = packed.shape[2]
    ldm = M if ldm is None else int(ldm)
    y = torch.empty((B, T, ldm), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_conv3_tc_bf16(_p(x), B, T, Cin, _p(packed), M, _p(bias), _p(y), ldm, _stream()))
    return y


def transpose_cast_bf16(x):
    """x fp32 [P,C,T] (DPNHead's input layout) -> bf16 channels-last [P,T,C] (tspn_transpose_cast_bf16)."""
    _dev(x, "x")
    P, C, T = x.shape
    out = torch.empty((P, T, C), dtype=torch.bfloat16, device=x.device)
    _abi.check(_abi.lib().tspn_transpose_cast_bf16(_p(x), P, C, T, _p(out), _stream()))
    return out


def temporal_encoder_heads_bf16(x, conv_packed, conv_bias, head_packed, head_b, H, y_ws=None):
    """DPNHead.forward (dpn.py:69-73) on bf16 operands: x fp32 [P,C,T] (transposed and rounded here) or bf16
    channels-last [P,T,C]; conv_packed = pack_conv3_bf16(conv.weight) (no split), head_packed =
    pack_heads_bf16(cat(relness, duration weights)); biases fp32 holding bf16 values -> heads fp32 [P,H,T]."""
    if x.dtype == torch.float32:
        x = transpose_cast_bf16(x.contiguous())
    _dev(x, "x", torch.bfloat16); _dev(conv_packed, "conv_packed", torch.bfloat16)
    _dev(head_packed, "head_packed", torch.bfloat16); _dev(head_b, "head_b")
    if conv_bias is not None:
        _dev(conv_bias, "conv_bias")
    P, T, C = x.shape
    if tuple(conv_packed.shape) != (3, C // 8, C, 8) or tuple(head_packed.shape) != (C // 8, 16, 8):
        raise ValueError(f"temporal_encoder_heads_bf16: packed weights do not match C={C}")
    if head_b.numel() != H:
        raise ValueError("temporal_encoder_heads_bf16: head_b must have H entries")
    if y_ws is None:
        y_ws = torch.empty((P, T, C), dtype=torch.float32, device=x.device)
    out = torch.empty((P, H, T), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_temporal_encoder_heads_bf16(
        _p(x), P, C, T, _p(conv_packed), _p(conv_bias), _p(head_packed), _p(head_b), H, _p(y_ws), _p(out), _stream()))
    return out


def heads_pairgrid_bf16(y, B, N, head_packed, head_b, H):
    """Pair stage on y fp32 [B*N, T, ldm >= 2C] (U | V halves, C from head_packed) -> heads fp32
    [B*N*(N-1), H, T]."""
    _dev(y, "y"); _dev(head_packed, "head_packed", torch.bfloat16); _dev(head_b, "head_b")
    BN, T, ldm = y.shape
    C = head_packed.shape[0] * 8
    if BN != B * N or ldm < 2 * C or tuple(head_packed.shape[1:]) != (16, 8) or head_b.numel() != H:
        raise ValueError("heads_pairgrid_bf16: shape mismatch")
    out = torch.empty((B * N * (N - 1), H, T), dtype=torch.float32, device=y.device)
    _abi.check(_abi.lib().tspn_heads_pairgrid_bf16(_p(y), ldm, B, N, C, T, _p(head_packed), _p(head_b),
                                                   H, _p(out), _stream()))
    return out


def temporal_mean_bf16(x):
    """bf16 [R,T,D] -> fp32 [R,D] holding bf16-rounded means."""
    _dev(x, "x", torch.bfloat16)
    R, T, D = x.shape
    out = torch.empty((R, D), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_temporal_mean_bf16(_p(x), R, T, D, _p(out), _stream()))
    return out


def fused_bf16_workspace_bytes(B, N, T, D, A, K, P):
    d = _abi.FusedBf16Desc()
    d.B, d.N, d.T, d.D, d.A, d.K, d.P = B, N, T, D, A, K, P
    return _abi.lib().tspn_forward_fused_bf16_workspace_bytes(ctypes.byref(d))


def forward_fused_bf16(feats, pairs, B, N, conv_packed, conv_bias, head_packed, head_b, cls_w, cls_b,
                       workspace=None, conv_events=None, logits_event=None, out_heads=None, out_logits=None):
    """Whole scoring pass, bf16 operands (tspn_forward_fused_bf16), canonical pair table only.

    feats bf16 [B*N,T,D]; conv_packed = pack_conv3_bf16(conv.weight, split=D); head_packed =
    pack_heads_bf16([3A,C]); cls_w fp32 [K,C] holding bf16-rounded values.
    Returns (heads fp32 [P,3A,T], rel_logits fp32 [P,K])."""
    _dev(feats, "feats", torch.bfloat16); _dev(pairs, "pairs", torch.int64)
    _dev(conv_packed, "conv_packed", torch.bfloat16); _dev(conv_bias, "conv_bias")
    _dev(head_packed, "head_packed", torch.bfloat16); _dev(head_b, "head_b")
    _dev(cls_w, "cls_w"); _dev(cls_b, "cls_b")
    BN, T, D = feats.shape
    C = 2 * D
    if BN != B * N:
        raise ValueError("forward_fused_bf16: feats rows != B*N")
    if tuple(conv_packed.shape) != (3, D // 8, 2 * C, 8):
        raise ValueError(f"forward_fused_bf16: conv_packed {tuple(conv_packed.shape)} != {(3, D // 8, 2 * C, 8)}")
    if tuple(head_packed.shape) != (C // 8, 16, 8) or head_b.numel() % 3:
        raise ValueError("forward_fused_bf16: head weights do not match C")
    A, K = head_b.numel() // 3, cls_w.shape[0]
    if tuple(cls_w.shape) != (K, C) or conv_bias.numel() != C or cls_b.numel() != K:
        raise ValueError("forward_fused_bf16: classifier / bias shapes do not match")
    P = pairs.shape[0]
    if tuple(pairs.shape) != (B * N * (N - 1), 2):
        raise ValueError("forward_fused_bf16: needs the canonical pair table [B*N*(N-1), 2]")
    d = _abi.FusedBf16Desc()
    d.B, d.N, d.T, d.D, d.A, d.K, d.P = B, N, T, D, A, K, P
    d.feats, d.pairs = feats.data_ptr(), pairs.data_ptr()
    d.conv_packed, d.conv_bias = conv_packed.data_ptr(), conv_bias.data_ptr()
    d.head_packed, d.head_b = head_packed.data_ptr(), head_b.data_ptr()
    d.cls_w, d.cls_b = cls_w.data_ptr(), cls_b.data_ptr()
    l = _abi.lib()
    need = l.tspn_forward_fused_bf16_workspace_bytes(ctypes.byref(d))
    if workspace is None:
        workspace = _ws(need, feats.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("forward_fused_bf16: workspace too small")
    if out_heads is None:
        out_heads = torch.empty((P, 3 * A, T), dtype=torch.float32, device=feats.device)
    elif tuple(out_heads.shape) != (P, 3 * A, T):
        raise ValueError("forward_fused_bf16: out_heads shape mismatch")
    if out_logits is None:
        out_logits = torch.empty((P, K), dtype=torch.float32, device=feats.device)
    elif tuple(out_logits.shape) != (P, K):
        raise ValueError("forward_fused_bf16: out_logits shape mismatch")
    _dev(out_heads, "out_heads"); _dev(out_logits, "out_logits")
    d.out_heads, d.out_logits = out_heads.data_ptr(), out_logits.data_ptr()
    d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if conv_events is not None:
        d.ev_conv_begin, d.ev_conv_end = conv_events[0].cuda_event, conv_events[1].cuda_event
    if logits_event is not None:
        d.ev_logits_ready = logits_event.cuda_event
    _abi.check(l.tspn_forward_fused_bf16(ctypes.byref(d), _stream()))
    return out_heads, out_logits


def span_predicate(feats, pairs, spans, cls_w, cls_b):
    """Span-restricted RelOIPool + predicate head (tspn_span_predicate_f32): feats [NT,T,D], pairs [P,2]
    global tracklet ids, spans int64 [P,2] frames [start,end) -> sigmoid logits [P,K]."""
    _dev(feats, "feats"); _dev(pairs, "pairs", torch.int64); _dev(spans, "spans", torch.int64)
    _dev(cls_w, "cls_w")
    if cls_b is not None:
        _dev(cls_b, "cls_b")
    NT, T, D = feats.shape
    K = cls_w.shape[0]
    P = pairs.shape[0]
    if tuple(cls_w.shape) != (K, 2 * D) or tuple(pairs.shape) != (P, 2) or tuple(spans.shape) != (P, 2):
        raise ValueError("span_predicate: shape mismatch")
    if P and (int(pairs.min()) < 0 or int(pairs.max()) >= NT):
        raise IndexError("span_predicate: pair index out of range")
    l = _abi.lib()
    ws = _ws(l.tspn_span_predicate_workspace_bytes(NT, T, D, K), feats.device)
    out = torch.empty((P, K), dtype=torch.float32, device=feats.device)
    _abi.check(l.tspn_span_predicate_f32(_p(feats), NT, T, D, _p(pairs), _p(spans), P, _p(cls_w), _p(cls_b), K,
                                         _p(out), _p(ws), ws.numel(), _stream()))
    return out


# --------------------------------------------------------------------------- #
# f4 (first slice): RoI feature head operators (csrc/tspn_roi.hip)
# --------------------------------------------------------------------------- #
def pack_conv2d(weight):
    """nn.Conv2d weight [Cout,Cin,KH,KW] -> [KH*KW, Cin, Cout] (tspn_pack_conv2d_f32)."""
    _dev(weight, "conv2d weight")
    if weight.dim() != 4:
        raise ValueError("pack_conv2d: weight must be [Cout,Cin,KH,KW]")
    Cout, Cin, KH, KW = weight.shape
    packed = torch.empty((KH * KW, Cin, Cout), dtype=torch.float32, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv2d_f32(_p(weight), Cout, Cin, KH, KW, _p(packed), _stream()))
    return packed


def pack_conv2d_frag(weight):
    """nn.Conv2d weight [Cout,Cin,KH,KW] -> fragment-major [Cout/32, KH*KW, Cin/16, 64, 8]
    (tspn_pack_conv2d_frag_f32; the registers-direct kernel).  Needs Cout % 32 == 0, Cin % 16 == 0."""
    _dev(weight, "conv2d weight")
    if weight.dim() != 4:
        raise ValueError("pack_conv2d_frag: weight must be [Cout,Cin,KH,KW]")
    Cout, Cin, KH, KW = weight.shape
    if Cout % 32 or Cin % 16:
        raise ValueError(f"pack_conv2d_frag: needs Cout % 32 == 0 and Cin % 16 == 0 (Cout={Cout}, Cin={Cin})")
    frag = torch.empty((Cout // 32, KH * KW, Cin // 16, 64, 8), dtype=torch.float32, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv2d_frag_f32(_p(weight), Cout, Cin, KH, KW, _p(frag), _stream()))
    return frag


def conv2d_nhwc(x, packed, kernel_size, stride=1, padding=0, bias=None, residual=None, relu=False):
    """act(conv2d(x) + bias + residual) on channels-last tensors: x [NB,H,W,Cin] -> [NB,OH,OW,Cout];
    `packed` = pack_conv2d(weight) or pack_conv2d_frag(weight) (5-D: fast kernel), kernel_size = (KH, KW)."""
    _dev(x, "x"); _dev(packed, "packed")
    NB, H, W, Cin = x.shape
    KH, KW = kernel_size
    frag = packed.dim() == 5
    if frag:
        if tuple(packed.shape[1:]) != (KH * KW, Cin // 16, 64, 8) or Cin % 16:
            raise ValueError(f"conv2d_nhwc: fragment-major weights {tuple(packed.shape)} do not match "
                             f"taps={KH * KW}, Cin={Cin}")
        Cout = packed.shape[0] * 32
    else:
        if packed.dim() != 3 or packed.shape[0] != KH * KW or packed.shape[1] != Cin:
            raise ValueError(f"conv2d_nhwc: packed weights {tuple(packed.shape)} do not match taps={KH * KW}, Cin={Cin}")
        Cout = packed.shape[2]
    OH, OW = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    if OH <= 0 or OW <= 0:
        raise ValueError("conv2d_nhwc: empty output")
    if bias is not None:
        _dev(bias, "bias")
        if bias.shape != (Cout,):
            raise ValueError("conv2d_nhwc: bias shape mismatch")
    if residual is not None:
        _dev(residual, "residual")
        if tuple(residual.shape) != (NB, OH, OW, Cout):
            raise ValueError("conv2d_nhwc: residual shape mismatch")
    out = torch.empty((NB, OH, OW, Cout), dtype=torch.float32, device=x.device)
    fn = _abi.lib().tspn_conv2d_nhwc_frag_f32 if frag else _abi.lib().tspn_conv2d_nhwc_f32
    _abi.check(fn(_p(x), NB, H, W, Cin, _p(packed), Cout, KH, KW, stride, padding,
                  _p(bias), _p(residual), 1 if relu else 0, _p(out), _stream()))
    return out


def roi_align_nhwc(feat, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True, out_bf16=False,
                   bin_stride=1):
    """detectron2 ROIAlign on a channels-last map: feat [NF,H,W,C], rois [R,5] = (map index, x1, y1, x2, y2)
    -> [R,P,P,C]; `out_bf16`: the fp32 result rounded once to bf16.  A bf16 map is read as it is
    (interpolation in fp32) and always gives bf16.  `bin_stride` = bs: only the bins (bs i, bs j) -> [R,OP,OP,C],
    OP = ceil(P / bs)."""
    bf16_in = isinstance(feat, torch.Tensor) and feat.dtype == torch.bfloat16
    _dev(feat, "feat", torch.bfloat16 if bf16_in else torch.float32); _dev(rois, "rois")
    out_bf16 = out_bf16 or bf16_in
    NF, H, W, C = feat.shape
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise ValueError("roi_align_nhwc: rois must be [R,5]")
    R, P, bs = rois.shape[0], int(output_size), int(bin_stride)
    if bs < 1:
        raise ValueError("roi_align_nhwc: bin_stride must be >= 1")
    OP = (P + bs - 1) // bs
    out = torch.empty((R, OP, OP, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=feat.device)
    fn = _abi.lib().tspn_roi_align_nhwc_bf16 if bf16_in else (
        _abi.lib().tspn_roi_align_nhwc_f32_bf16out if out_bf16 else _abi.lib().tspn_roi_align_nhwc_f32)
    _abi.check(fn(_p(feat), NF, H, W, C, _p(rois), R, P, float(spatial_scale),
                  int(sampling_ratio), 1 if aligned else 0, bs, _p(out), _stream()))
    return out


def pack_conv2d_frag_bf16(weight):
    """fp32 nn.Conv2d weight [Cout,Cin,KH,KW] -> bf16 fragment-major [Cout/32, KH*KW, Cin/64, 4, 64, 8]
    (tspn_pack_conv2d_frag_bf16; rounded once, to nearest even).  Needs Cout % 32 == 0, Cin % 64 == 0."""
    _dev(weight, "conv2d weight")
    if weight.dim() != 4:
        raise ValueError("pack_conv2d_frag_bf16: weight must be [Cout,Cin,KH,KW]")
    Cout, Cin, KH, KW = weight.shape
    if Cout % 32 or Cin % 64:
        raise ValueError(f"pack_conv2d_frag_bf16: needs Cout % 32 == 0 and Cin % 64 == 0 (Cout={Cout}, Cin={Cin})")
    frag = torch.empty((Cout // 32, Cin // 64, KH * KW, 4, 64, 8), dtype=torch.bfloat16, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv2d_frag_bf16(_p(weight), Cout, Cin, KH, KW, _p(frag), _stream()))
    return frag


def conv2d_nhwc_bf16(x, frag, kernel_size, stride=1, padding=0, bias=None, residual=None, relu=False):
    """bf16-operand conv2d on channels-last tensors (tspn_conv2d_nhwc_bf16): x bf16 [NB,H,W,Cin], frag =
    pack_conv2d_frag_bf16(weight), bias fp32 [Cout], residual bf16 -> bf16 [NB,OH,OW,Cout]
    = bf16(act(fp32 sum of exact products + bias + residual))."""
    _dev(x, "x", torch.bfloat16); _dev(frag, "frag", torch.bfloat16)
    NB, H, W, Cin = x.shape
    KH, KW = kernel_size
    if frag.dim() != 6 or tuple(frag.shape[1:]) != (Cin // 64, KH * KW, 4, 64, 8) or Cin % 64:
        raise ValueError(f"conv2d_nhwc_bf16: weights {tuple(frag.shape)} do not match taps={KH * KW}, Cin={Cin}")
    Cout = frag.shape[0] * 32
    OH, OW = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    if OH <= 0 or OW <= 0:
        raise ValueError("conv2d_nhwc_bf16: empty output")
    if bias is not None:
        _dev(bias, "bias")
        if bias.shape != (Cout,):
            raise ValueError("conv2d_nhwc_bf16: bias shape mismatch")
    if residual is not None:
        _dev(residual, "residual", torch.bfloat16)
        if tuple(residual.shape) != (NB, OH, OW, Cout):
            raise ValueError("conv2d_nhwc_bf16: residual shape mismatch")
    out = torch.empty((NB, OH, OW, Cout), dtype=torch.bfloat16, device=x.device)
    _abi.check(_abi.lib().tspn_conv2d_nhwc_bf16(_p(x), NB, H, W, Cin, _p(frag), Cout, KH, KW, stride, padding,
                                                _p(bias), _p(residual), 1 if relu else 0, _p(out), _stream()))
    return out


def max_pool_nhwc(x, kernel_size=3, stride=2, padding=1, out_bf16=False):
    """max_pool2d on a channels-last fp32 map [NB,H,W,C] -> [NB,OH,OW,C] (fp32, or bf16 rounded once)."""
    _dev(x, "x")
    NB, H, W, C = x.shape
    OH, OW = (H + 2 * padding - kernel_size) // stride + 1, (W + 2 * padding - kernel_size) // stride + 1
    if OH <= 0 or OW <= 0:
        raise ValueError("max_pool_nhwc: empty output")
    out = torch.empty((NB, OH, OW, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_max_pool_nhwc_f32(_p(x), NB, H, W, C, kernel_size, stride, padding, _p(out),
                                                 1 if out_bf16 else 0, _stream()))
    return out


def bottleneck_block_bf16(x, frag1, bias1, frag2, bias2, frag3, bias3, out=None):
    """A whole identity-shortcut bottleneck block in one launch (tspn_bottleneck_block_bf16):
    relu(conv1x1(relu(conv3x3(relu(conv1x1(x) + b1)) + b2)) + b3 + x) for x bf16 [NB,H,W,4 CM], CM in (64, 128);
    frag1 / frag2 / frag3 = pack_conv2d_frag_bf16 of the folded conv1 [CM,4CM,1,1], conv2 [CM,CM,3,3], conv3 [4CM,CM,1,1].
    Bit-identical to conv2d_nhwc_bf16 (conv1) + bottleneck_tail_bf16; the 4 CM-channel map is read once."""
    _dev(x, "x", torch.bfloat16)
    for nm, t in (("frag1", frag1), ("frag2", frag2), ("frag3", frag3)):
        _dev(t, nm, torch.bfloat16)
    _dev(bias1, "bias1"); _dev(bias2, "bias2"); _dev(bias3, "bias3")
    NB, H, W, C4 = x.shape
    CM = C4 // 4
    if CM not in (64, 128) or C4 != 4 * CM:
        raise ValueError(f"bottleneck_block_bf16: needs 4 x 64 or 4 x 128 channels (got {C4})")
    if (tuple(frag1.shape) != (CM // 32, C4 // 64, 1, 4, 64, 8) or tuple(frag2.shape) != (CM // 32, CM // 64, 9, 4, 64, 8)
            or tuple(frag3.shape) != (CM // 8, CM // 64, 1, 4, 64, 8)):
        raise ValueError("bottleneck_block_bf16: frag1 / frag2 / frag3 must be pack_conv2d_frag_bf16 of [CM,4CM,1,1] / "
                         "[CM,CM,3,3] / [4CM,CM,1,1]")
    if bias1.shape != (CM,) or bias2.shape != (CM,) or bias3.shape != (C4,):
        raise ValueError("bottleneck_block_bf16: bias shape mismatch")
    if out is None:
        out = torch.empty_like(x)
    else:
        _dev(out, "out", torch.bfloat16)
        if tuple(out.shape) != tuple(x.shape):
            raise ValueError(f"bottleneck_block_bf16: out must be {tuple(x.shape)}, got {tuple(out.shape)}")
    _abi.check(_abi.lib().tspn_bottleneck_block_bf16(_p(x), NB, H, W, CM, _p(frag1), _p(bias1), _p(frag2), _p(bias2),
                                                     _p(frag3), _p(bias3), _p(out), _stream()))
    return out


def bottleneck_block_proj_bf16(x, stride, frag1, bias1, frag2, bias2, frag3, bias3, frags, biass, out=None):
    """The first block of a stage in one launch (tspn_bottleneck_block_proj_bf16): conv1 and the projection shortcut are
    1x1 convs of stride `stride` on x bf16 [NB,Hin,Win,CIN]; frags / biass = the folded shortcut [4CM,CIN,1,1].  Built for
    (CIN, CM, stride) = (64, 64, 1), detectron2's res2.0.  Bit-identical to conv1 + shortcut + fused tail; the shortcut map
    never goes to memory."""
    _dev(x, "x", torch.bfloat16)
    for nm, t in (("frag1", frag1), ("frag2", frag2), ("frag3", frag3), ("frags", frags)):
        _dev(t, nm, torch.bfloat16)
    for nm, t in (("bias1", bias1), ("bias2", bias2), ("bias3", bias3), ("biass", biass)):
        _dev(t, nm)
    NB, Hin, Win, CIN = x.shape
    CM = frag2.shape[0] * 32
    if (CIN, CM, int(stride)) != (64, 64, 1):
        raise ValueError(f"bottleneck_block_proj_bf16: built for (CIN, CM, stride) = (64, 64, 1), got {(CIN, CM, stride)}")
    if (tuple(frag1.shape) != (CM // 32, CIN // 64, 1, 4, 64, 8) or tuple(frag2.shape) != (CM // 32, CM // 64, 9, 4, 64, 8)
            or tuple(frag3.shape) != (CM // 8, CM // 64, 1, 4, 64, 8) or tuple(frags.shape) != (CM // 8, CIN // 64, 1, 4, 64, 8)):
        raise ValueError("bottleneck_block_proj_bf16: fragment shapes do not match [CM,CIN,1,1] / [CM,CM,3,3] / [4CM,CM,1,1] / [4CM,CIN,1,1]")
    if bias1.shape != (CM,) or bias2.shape != (CM,) or bias3.shape != (4 * CM,) or biass.shape != (4 * CM,):
        raise ValueError("bottleneck_block_proj_bf16: bias shape mismatch")
    H, W = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    if out is None:
        out = torch.empty((NB, H, W, 4 * CM), dtype=torch.bfloat16, device=x.device)
    else:
        _dev(out, "out", torch.bfloat16)
        if tuple(out.shape) != (NB, H, W, 4 * CM):
            raise ValueError(f"bottleneck_block_proj_bf16: out must be {(NB, H, W, 4 * CM)}, got {tuple(out.shape)}")
    _abi.check(_abi.lib().tspn_bottleneck_block_proj_bf16(_p(x), NB, Hin, Win, CIN, int(stride), CM, _p(frag1), _p(bias1),
                                                          _p(frag2), _p(bias2), _p(frag3), _p(bias3), _p(frags), _p(biass),
                                                          _p(out), _stream()))
    return out


def bottleneck_block_res_bf16(x, stride, frag1, bias1, frag2, bias2, frag3, bias3, residual, out=None):
    """conv1 (1x1, stride `stride`) + 3x3 + expand + `residual` + ReLU in one launch (tspn_bottleneck_block_res_bf16): x bf16
    [NB,Hin,Win,CIN], residual bf16 [NB,H,W,4CM] (e.g. the separately launched projection shortcut).  Built for res3.0:
    (CIN, CM, stride) = (256, 128, 2).  Bit-identical to conv1 + bottleneck_tail_bf16."""
    _dev(x, "x", torch.bfloat16); _dev(residual, "residual", torch.bfloat16)
    for nm, t in (("frag1", frag1), ("frag2", frag2), ("frag3", frag3)):
        _dev(t, nm, torch.bfloat16)
    for nm, t in (("bias1", bias1), ("bias2", bias2), ("bias3", bias3)):
        _dev(t, nm)
    NB, Hin, Win, CIN = x.shape
    CM = frag2.shape[0] * 32
    if (CIN, CM, int(stride)) != (256, 128, 2):
        raise ValueError(f"bottleneck_block_res_bf16: built for (CIN, CM, stride) = (256, 128, 2), got {(CIN, CM, stride)}")
    if (tuple(frag1.shape) != (CM // 32, CIN // 64, 1, 4, 64, 8) or tuple(frag2.shape) != (CM // 32, CM // 64, 9, 4, 64, 8)
            or tuple(frag3.shape) != (CM // 8, CM // 64, 1, 4, 64, 8)):
        raise ValueError("bottleneck_block_res_bf16: fragment shapes do not match [CM,CIN,1,1] / [CM,CM,3,3] / [4CM,CM,1,1]")
    H, W = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    if bias1.shape != (CM,) or bias2.shape != (CM,) or bias3.shape != (4 * CM,) or tuple(residual.shape) != (NB, H, W, 4 * CM):
        raise ValueError("bottleneck_block_res_bf16: bias / residual shape mismatch")
    if out is None:
        out = torch.empty((NB, H, W, 4 * CM), dtype=torch.bfloat16, device=x.device)
    else:
        _dev(out, "out", torch.bfloat16)
        if tuple(out.shape) != (NB, H, W, 4 * CM):
            raise ValueError(f"bottleneck_block_res_bf16: out must be {(NB, H, W, 4 * CM)}, got {tuple(out.shape)}")
    _abi.check(_abi.lib().tspn_bottleneck_block_res_bf16(_p(x), NB, Hin, Win, CIN, int(stride), CM, _p(frag1), _p(bias1), _p(frag2),
                                                         _p(bias2), _p(frag3), _p(bias3), _p(residual), _p(out), _stream()))
    return out


def bottleneck_tail_bf16(h1, frag2, bias2, frag3, bias3, residual, out=None, next_frag1=None, next_bias1=None,
                         persistent=False, max_workgroups=0, io_waves=False):
    """relu(conv1x1(relu(conv3x3(h1) + b2)) + b3 + residual) in one launch (tspn_bottleneck_tail_bf16): h1 bf16
    [NB,H,W,CM], CM in (64, 128, 256); frag2 / frag3 = pack_conv2d_frag_bf16 of the folded conv2 / conv3 weights;
    residual bf16 [NB,H,W,4 CM] -> bf16 [NB,H,W,4 CM] (written into `out` when given: a contiguous tensor of that
    shape, e.g. a slice of the caller's result along the first dimension).
    `next_frag1`, `next_bias1` (CM = 256): pack_conv2d_frag_bf16 of the FOLLOWING block's conv1 [CM,4 CM,1,1] and its
    bias: the launch also computes that conv1 on its own output (tspn_bottleneck_tail_next_bf16) and the call returns
    (out, h1_next [NB,H,W,CM]) -- both bit-identical to the separate launches.
    `persistent` (CM = 256): the persistent kernel pipelined across tiles (tspn_bottleneck_tail_pipe_bf16: one
    workgroup per CU, the 3x3 phase of tile t beside the expand / store phase of tile t - 1), same results;
    `max_workgroups` > 0 limits its grid (tests).
    `io_waves` (CM = 256): the role-split kernel (tspn_bottleneck_tail_io_bf16, round 5: four MFMA waves + four waves that
    own the h1 DMA, the residual rows and the stores; what the backbone uses for res4), same results."""
    _dev(h1, "h1", torch.bfloat16); _dev(frag2, "frag2", torch.bfloat16); _dev(frag3, "frag3", torch.bfloat16)
    _dev(bias2, "bias2"); _dev(bias3, "bias3"); _dev(residual, "residual", torch.bfloat16)
    NB, H, W, CM = h1.shape
    if CM not in (64, 128, 256):
        raise ValueError(f"bottleneck_tail_bf16: bottleneck channels must be 64, 128 or 256 (got {CM})")
    if tuple(frag2.shape) != (CM // 32, CM // 64, 9, 4, 64, 8) or tuple(frag3.shape) != (CM // 8, CM // 64, 1, 4, 64, 8):
        raise ValueError("bottleneck_tail_bf16: frag2 / frag3 must be pack_conv2d_frag_bf16 of [CM,CM,3,3] / [4CM,CM,1,1]")
    if bias2.shape != (CM,) or bias3.shape != (4 * CM,) or tuple(residual.shape) != (NB, H, W, 4 * CM):
        raise ValueError("bottleneck_tail_bf16: bias / residual shape mismatch")
    if out is None:
        out = torch.empty((NB, H, W, 4 * CM), dtype=torch.bfloat16, device=h1.device)
    else:
        _dev(out, "out", torch.bfloat16)
        if tuple(out.shape) != (NB, H, W, 4 * CM):
            raise ValueError(f"bottleneck_tail_bf16: out must be {(NB, H, W, 4 * CM)}, got {tuple(out.shape)}")
    if io_waves:
        # (CM = 256) the role-split kernel: four MFMA waves + four waves that do all HBM traffic (tspn_bottleneck_tail_io_bf16)
        if CM != 256 or next_frag1 is not None or persistent:
            raise ValueError("bottleneck_tail_bf16: io_waves is built for 256 bottleneck channels, without next_frag1 / persistent")
        _abi.check(_abi.lib().tspn_bottleneck_tail_io_bf16(_p(h1), NB, H, W, CM, _p(frag2), _p(bias2), _p(frag3), _p(bias3),
                                                           _p(residual), _p(out), _stream()))
        return out
    if persistent:
        if CM != 256 or next_frag1 is not None:
            raise ValueError("bottleneck_tail_bf16: the persistent form is built for 256 bottleneck channels, without next_frag1")
        _abi.check(_abi.lib().tspn_bottleneck_tail_pipe_bf16(_p(h1), NB, H, W, CM, _p(frag2), _p(bias2), _p(frag3), _p(bias3),
                                                             _p(residual), _p(out), int(max_workgroups), _stream()))
        return out
    if next_frag1 is not None:
        _dev(next_frag1, "next_frag1", torch.bfloat16); _dev(next_bias1, "next_bias1")
        if CM != 256:
            raise ValueError(f"bottleneck_tail_bf16: the fused next conv1 needs 256 bottleneck channels (got {CM})")
        if tuple(next_frag1.shape) != (CM // 32, 4 * CM // 64, 1, 4, 64, 8) or next_bias1.shape != (CM,):
            raise ValueError("bottleneck_tail_bf16: next_frag1 must be pack_conv2d_frag_bf16 of [CM,4CM,1,1], next_bias1 [CM]")
        h1n = torch.empty((NB, H, W, CM), dtype=torch.bfloat16, device=h1.device)
        _abi.check(_abi.lib().tspn_bottleneck_tail_next_bf16(
            _p(h1), NB, H, W, CM, _p(frag2), _p(bias2), _p(frag3), _p(bias3), _p(residual), _p(out),
            _p(next_frag1), _p(next_bias1), _p(h1n), _stream()))
        return out, h1n
    _abi.check(_abi.lib().tspn_bottleneck_tail_bf16(_p(h1), NB, H, W, CM, _p(frag2), _p(bias2), _p(frag3), _p(bias3),
                                                    _p(residual), _p(out), _stream()))
    return out


def max_pool_nhwc_bf16(x, kernel_size=3, stride=2, padding=1):
    """max_pool2d on a channels-last bf16 map [NB,H,W,C] -> bf16 [NB,OH,OW,C] (tspn_max_pool_nhwc_bf16; C % 8 == 0)."""
    _dev(x, "x", torch.bfloat16)
    NB, H, W, C = x.shape
    OH, OW = (H + 2 * padding - kernel_size) // stride + 1, (W + 2 * padding - kernel_size) // stride + 1
    if OH <= 0 or OW <= 0:
        raise ValueError("max_pool_nhwc_bf16: empty output")
    out = torch.empty((NB, OH, OW, C), dtype=torch.bfloat16, device=x.device)
    _abi.check(_abi.lib().tspn_max_pool_nhwc_bf16(_p(x), NB, H, W, C, kernel_size, stride, padding, _p(out), _stream()))
    return out


def pack_stem_bf16(weight):
    """Stem weight [Cout in (32, 64), 3, 7, 7] fp32 (batch norm folded) -> bf16 fragment-major [Cout/32, 16, 64, 8]
    (tspn_pack_stem_bf16: the 7x7/2 conv as a 4x4/1 conv on the 2x2 space-to-depth image)."""
    _dev(weight, "stem weight")
    if weight.dim() != 4 or tuple(weight.shape[1:]) != (3, 7, 7) or weight.shape[0] not in (32, 64):
        raise ValueError(f"pack_stem_bf16: weight must be [32 | 64, 3, 7, 7], got {tuple(weight.shape)}")
    frag = torch.empty((weight.shape[0] // 32, 16, 64, 8), dtype=torch.bfloat16, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_stem_bf16(_p(weight), weight.shape[0], _p(frag), _stream()))
    return frag


def stem_conv_bf16(x, frag, bias, workspace=None):
    """relu(conv7x7/2/pad3(x) + bias) with bf16 operands: x fp32 [NB,H,W,3] -> bf16 [NB,OH,OW,Cout]
    (tspn_stem_conv_bf16; `workspace` >= tspn_stem_bf16_workspace_bytes holds the space-to-depth image)."""
    _dev(x, "x"); _dev(frag, "frag", torch.bfloat16); _dev(bias, "bias")
    if x.dim() != 4 or x.shape[3] != 3 or frag.dim() != 4 or tuple(frag.shape[1:]) != (16, 64, 8):
        raise ValueError("stem_conv_bf16: x must be [NB,H,W,3] and frag = pack_stem_bf16(weight)")
    NB, H, W, _ = x.shape
    Cout = frag.shape[0] * 32
    if bias.shape != (Cout,):
        raise ValueError("stem_conv_bf16: bias shape mismatch")
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    l = _abi.lib()
    need = l.tspn_stem_bf16_workspace_bytes(NB, H, W)
    if workspace is None:
        workspace = _ws(need, x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("stem_conv_bf16: workspace too small")
    out = torch.empty((NB, OH, OW, Cout), dtype=torch.bfloat16, device=x.device)
    _abi.check(l.tspn_stem_conv_bf16(_p(x), NB, H, W, _p(frag), Cout, _p(bias), _p(workspace),
                                     workspace.numel() * workspace.element_size(), _p(out), _stream()))
    return out


def stem_pool_bf16(x, frag, bias, workspace=None):
    """BasicStem.forward on bf16 operands in one conv launch: max_pool2d(relu(conv7x7/2/pad3(x) + bias), 3, 2, 1),
    x fp32 [NB,H,W,3] -> bf16 [NB,PH,PW,Cout] (tspn_stem_pool_bf16); bit-identical to
    max_pool_nhwc_bf16(stem_conv_bf16(x, frag, bias), 3, 2, 1) without writing the conv map."""
    _dev(x, "x"); _dev(frag, "frag", torch.bfloat16); _dev(bias, "bias")
    if x.dim() != 4 or x.shape[3] != 3 or frag.dim() != 4 or tuple(frag.shape[1:]) != (16, 64, 8):
        raise ValueError("stem_pool_bf16: x must be [NB,H,W,3] and frag = pack_stem_bf16(weight)")
    NB, H, W, _ = x.shape
    Cout = frag.shape[0] * 32
    if bias.shape != (Cout,):
        raise ValueError("stem_pool_bf16: bias shape mismatch")
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    l = _abi.lib()
    need = l.tspn_stem_bf16_workspace_bytes(NB, H, W)
    if workspace is None:
        workspace = _ws(need, x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("stem_pool_bf16: workspace too small")
    out = torch.empty((NB, (OH - 1) // 2 + 1, (OW - 1) // 2 + 1, Cout), dtype=torch.bfloat16, device=x.device)
    _abi.check(l.tspn_stem_pool_bf16(_p(x), NB, H, W, _p(frag), Cout, _p(bias), _p(workspace),
                                     workspace.numel() * workspace.element_size(), _p(out), _stream()))
    return out


def pack_conv2d_frag_cin4(weight):
    """Stem weights [Cout, Cin <= 4, KH, KW] -> [Cout/32, ceil(KH*KW/4), 64, 8] (tspn_pack_conv2d_frag_cin4_f32)."""
    _dev(weight, "conv2d weight")
    Cout, Cin, KH, KW = weight.shape
    if Cin > 4 or Cout % 32:
        raise ValueError(f"pack_conv2d_frag_cin4: needs Cin <= 4 and Cout % 32 == 0 (Cin={Cin}, Cout={Cout})")
    frag = torch.empty((Cout // 32, (KH * KW + 3) // 4, 64, 8), dtype=torch.float32, device=weight.device)
    _abi.check(_abi.lib().tspn_pack_conv2d_frag_cin4_f32(_p(weight), Cout, Cin, KH, KW, _p(frag), _stream()))
    return frag


def conv2d_nhwc_cin4(x, frag, kernel_size, stride=1, padding=0, bias=None, relu=False):
    """Stem conv on a 4-channel channels-last image x [NB,H,W,4] (RGB + one zero channel) -> [NB,OH,OW,Cout]."""
    _dev(x, "x"); _dev(frag, "frag")
    NB, H, W, C4 = x.shape
    KH, KW = kernel_size
    if C4 != 4 or frag.dim() != 4 or tuple(frag.shape[1:]) != ((KH * KW + 3) // 4, 64, 8):
        raise ValueError("conv2d_nhwc_cin4: x must be [NB,H,W,4] and frag = pack_conv2d_frag_cin4(weight)")
    Cout = frag.shape[0] * 32
    OH, OW = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    if bias is not None:
        _dev(bias, "bias")
    out = torch.empty((NB, OH, OW, Cout), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().tspn_conv2d_nhwc_cin4_f32(_p(x), NB, H, W, _p(frag), Cout, KH, KW, stride, padding,
                                                    _p(bias), 1 if relu else 0, _p(out), _stream()))
    return out


# every public operator runs with the device of its operands made current (see _on_tensor_device)
for _name in __all__:
    if _name not in ("pair_index", "fused_workspace_bytes"):
        globals()[_name] = _on_tensor_device(globals()[_name])
del _name
