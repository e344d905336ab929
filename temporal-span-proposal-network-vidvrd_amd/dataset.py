"""The in-repo half of the pair-feature builder: from the arrays of a segment's `-relation.h5` file to the
`PairList` / `TargetList` that `BaseModel.forward` consumes — the device mirror of
`VRDataset.__getitem__` (reference lib/dataset/vrdataset.py:61-83) without the file access.

    trackid [N+G]              -1 = tracklet proposal, >= 0 = ground-truth track     (vrdataset.py:203-204)
    pairs   [(N+G)(N+G-1), 2]  all ordered pairs among the N+G tracks                 (vrdataset.py:206)
    feats   [.., 11070]        RAW relation feature of every pair, same order         (vrdataset.py:208)
    iou     [N+G, N+G]         vIoU between tracks                                    (vrdataset.py:210)

What the reference does on the host per segment, and where it runs here:
    proposal_idx = _get_proposal_idx(pairs, trackid)        -> tspn_proposal_pair_filter_i64
    feats, pairs, labels = x[proposal_idx] for each         -> tspn_gather_rows_f32
    num_tracks = sum(trackid < 0)                           -> same kernel
    feats = _feature_preprocess(feats)                      -> tspn_feature_preprocess_f32 (in place), or
                                                               left RAW for PREDICT.FUSE_PREPROCESS
Reading the HDF5 / JSON files (h5py, vrdataset.py:190-217) and the label construction
(vrdataset.py:85-138) stay outside this build (dataset I/O is out of scope).
"""
import numpy as np
import torch

from . import ops
from .pair_list import PairList, TargetList

__all__ = ["proposal_pair_list", "select_proposal_pairs"]


def _to_dev(x, device, dtype):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(x))
    return t.to(device=device, dtype=dtype).contiguous()


def select_proposal_pairs(pairs, trackid, device=None):
    """(proposal_idx int64 [P'] on the device, num_tracklets int) — vrdataset.py:140-148."""
    device = torch.device(device) if device is not None else (
        pairs.device if isinstance(pairs, torch.Tensor) and pairs.is_cuda
        else torch.device("cuda", torch.cuda.current_device()))
    p = _to_dev(pairs, device, torch.int64).reshape(-1, 2)
    return ops.proposal_pair_filter(p, _to_dev(trackid, device, torch.int64).reshape(-1))


def proposal_pair_list(pairs, feats, iou, trackid, cls_logits, pred_labels=None, preprocess=True,
                       device=None):
    """One segment's (PairList, TargetList | None) as `VRDataset.__getitem__` builds them
    (vrdataset.py:64-83), resident on the HIP device.

    `preprocess=False` leaves the kept feature rows RAW for a model built with
    PREDICT.FUSE_PREPROCESS (the block-L1 normalisation then runs inside the predicate GEMM)."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    p_all = _to_dev(pairs, device, torch.int64).reshape(-1, 2)
    f_all = _to_dev(feats, device, torch.float32)
    if f_all.dim() != 2 or f_all.shape[0] != p_all.shape[0]:
        raise ValueError(f"feats must be [P,F] with one row per pair (pairs {tuple(p_all.shape)}, "
                         f"feats {tuple(f_all.shape)})")
    tid = _to_dev(trackid, device, torch.int64).reshape(-1)
    idx, num_tracks = ops.proposal_pair_filter(p_all, tid)
    kept_feats = ops.gather_rows(f_all, idx)
    # int64 [P,2] rows are 16 bytes: moved by the same byte-mover as four fp32 columns
    kept_pairs = ops.gather_rows(p_all.view(torch.float32), idx).view(torch.int64)
    if preprocess and kept_feats.shape[0]:
        first, block, nblocks = 70, 1000, 8          # vrdataset.py:227-236
        if kept_feats.shape[1] < first + block * nblocks:
            raise ValueError(f"_feature_preprocess needs F >= {first + block * nblocks}, got {kept_feats.shape[1]}")
        ops.feature_preprocess_(kept_feats, first, block, nblocks)
    plist = PairList(kept_feats)
    plist.add_field("tracklet_pairs", kept_pairs)
    plist.add_field("track_cls_logits", _to_dev(cls_logits, device, torch.float32))
    plist.add_field("num_tracklets", num_tracks)
    plist.add_field("ious", iou)
    plist.add_field("track_ids", trackid)
    tlist = None
    if pred_labels is not None:
        lab = _to_dev(pred_labels, device, torch.float32)
        if lab.dim() != 2 or lab.shape[0] != p_all.shape[0]:
            raise ValueError("pred_labels must be [P,K] with one row per pair")
        tlist = TargetList(ops.gather_rows(lab, idx))
    return plist, tlist
