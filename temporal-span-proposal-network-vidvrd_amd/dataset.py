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
The label construction (vrdataset.py:85-138) stays outside this build (dataset I/O is out of scope).

On-disk side (the formats a feature-extraction front end has to honour): the small helpers
at the end of this file read and write the two per-segment files the reference's dataset opens —
`<vsig>-traj_cls.json` (a list of `Trajectory.serialize()` dicts, lib/modeling/trajectory.py:72-82, read at
vrdataset.py:162-188) and `<vsig>-relation.h5` (datasets `trackid`, `pairs`, `feats`, `iou`, read at
vrdataset.py:190-217) — under the reference's own naming (`get_segment_signature`, `get_feature_path`,
lib/modeling/__init__.py:5-24).  JSON needs nothing; HDF5 needs `h5py`, which is an optional import (absent in
the build image: the functions then raise ImportError, nothing is emulated).
"""
import json
import os

import numpy as np
import torch

from . import ops
from .pair_list import PairList, TargetList

__all__ = ["DevicePrefetcher", "proposal_pair_list", "proposal_pair_lists", "select_proposal_pairs", "segment_signature", "feature_path",
           "write_traj_cls_json", "read_traj_cls_json", "tracklets_to_traj_cls", "write_relation_h5",
           "read_relation_h5"]


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
    return _assemble(p_all, f_all, idx, num_tracks, iou, trackid, cls_logits, pred_labels, preprocess, device)


def _assemble(p_all, f_all, idx, num_tracks, iou, trackid, cls_logits, pred_labels, preprocess, device):
    """Gathers + normalisation + field wiring of one segment, given its kept row numbers `idx` (produced and
    range-checked by the filter kernel, hence check_idx=False: no host sync in here)."""
    kept_feats = ops.gather_rows(f_all, idx, check_idx=False)
    # int64 [P,2] rows are 16 bytes: moved by the same byte-mover as four fp32 columns
    kept_pairs = ops.gather_rows(p_all.view(torch.float32), idx, check_idx=False).view(torch.int64)
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
        tlist = TargetList(ops.gather_rows(lab, idx, check_idx=False))
    return plist, tlist


def proposal_pair_lists(segments, preprocess=True, device=None):
    """A batch of segments at once: `segments` = sequence of dicts with the keys of `proposal_pair_list`
    (pairs, feats, iou, trackid, cls_logits[, pred_labels]).  The pair filter of ALL segments is ONE launch
    (`pair_off` / `track_off` form of tspn_proposal_pair_filter_i64) and the loader pays ONE host sync per
    batch (the kept-row counts) instead of several per segment.  Returns [(PairList, TargetList | None)]."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if not len(segments):
        return []
    p_all = [_to_dev(s["pairs"], device, torch.int64).reshape(-1, 2) for s in segments]
    f_all = [_to_dev(s["feats"], device, torch.float32) for s in segments]
    tids = [_to_dev(s["trackid"], device, torch.int64).reshape(-1) for s in segments]
    for p, f in zip(p_all, f_all):
        if f.dim() != 2 or f.shape[0] != p.shape[0]:
            raise ValueError(f"feats must be [P,F] with one row per pair (pairs {tuple(p.shape)}, feats {tuple(f.shape)})")
    pair_off = np.concatenate([[0], np.cumsum([p.shape[0] for p in p_all])]).astype(np.int64)
    track_off = np.concatenate([[0], np.cumsum([t.shape[0] for t in tids])]).astype(np.int64)
    idx, count, ntr = ops.proposal_pair_filter(torch.cat(p_all), torch.cat(tids),
                                               torch.from_numpy(pair_off).to(device), torch.from_numpy(track_off).to(device))
    counts = torch.stack([count, ntr]).cpu().numpy()          # the one sync of the batch
    if (counts[0] < 0).any():
        raise IndexError(f"proposal_pair_lists: segment {int(np.argmax(counts[0] < 0))}: a pair names a track index "
                         "outside trackid")
    out = []
    for k, s in enumerate(segments):
        lo = int(pair_off[k])
        out.append(_assemble(p_all[k], f_all[k], idx[lo:lo + int(counts[0, k])], int(counts[1, k]), s["iou"], s["trackid"],
                             s["cls_logits"], s.get("pred_labels"), preprocess, device))
    return out


# ---------------------------------------------------------------------------------------------------------
# on-disk formats of the reference's per-segment files
# ---------------------------------------------------------------------------------------------------------
def segment_signature(vid, fstart, fend):
    """'<vid>-<fstart:04d>-<fend:04d>' — lib/modeling/__init__.py:5-9."""
    return "{}-{:04d}-{:04d}".format(vid, int(fstart), int(fend))


def feature_path(root, name, vid, vsig=None, ext=None, create=False):
    """<root>/features/<name>/<vid>[/<vsig>-<name>.<ext>] — the layout of `get_feature_path` + the file names of
    vrdataset.py:168-170, 194-196 (`root` = './vidvrd-baseline-output' in the reference)."""
    d = os.path.join(root, "features", name, vid)
    if create:
        os.makedirs(d, exist_ok=True)
    return d if vsig is None else os.path.join(d, "{}-{}.{}".format(vsig, name, ext or "json"))


def tracklets_to_traj_cls(tracklet_boxes, track_cls_logits, fstart, vsig=None, scores=None, gt_trackids=None):
    """[N,T,4] boxes (l,t,r,b) + [N,35] classeme logits of one segment -> the list of `Trajectory.serialize()`
    dicts (trajectory.py:72-82) that `<vsig>-traj_cls.json` holds: what a feature-extraction front end writes
    for the dataset to read (`track_cls_logits` = the 'classeme' entries, vrdataset.py:150-160)."""
    boxes = np.asarray(tracklet_boxes.detach().cpu() if isinstance(tracklet_boxes, torch.Tensor) else tracklet_boxes,
                       dtype=np.float64)
    cls = np.asarray(track_cls_logits.detach().cpu() if isinstance(track_cls_logits, torch.Tensor) else track_cls_logits,
                     dtype=np.float64)
    n, t = boxes.shape[:2]
    if boxes.shape != (n, t, 4) or cls.shape[0] != n:
        raise ValueError("tracklets_to_traj_cls: boxes must be [N,T,4] and logits [N,K]")
    out = []
    for i in range(n):
        out.append({"pstart": int(fstart), "pend": int(fstart) + t,
                    "rois": [tuple(float(v) for v in b) for b in boxes[i]],
                    "score": float(scores[i]) if scores is not None else float(cls[i].max()),
                    "category": int(cls[i].argmax()),
                    "classeme": [float(x) for x in cls[i]],
                    "vsig": vsig,
                    "gt_trackid": int(gt_trackids[i]) if gt_trackids is not None else -1})
    return out


def write_traj_cls_json(path, trajs):
    with open(path, "w") as fout:
        json.dump([dict(t, rois=[list(r) for r in t["rois"]]) for t in trajs], fout)


def read_traj_cls_json(path, logit_only=True):
    """`VRDataset._get_object_trajectory_proposal` (vrdataset.py:162-188): the classeme rows as a float32
    [N,35] array (`logit_only`, what configs/baseline.yaml uses), or the list of dicts; [] if the file is absent."""
    if not os.path.exists(path):
        return np.zeros((0, 0), dtype=np.float32) if logit_only else []
    with open(path, "r") as fin:
        trajs = json.load(fin)
    if not logit_only:
        return trajs
    return np.asarray([t["classeme"] for t in trajs], dtype=np.float32)


def _h5py():
    try:
        import h5py
    except ImportError as exc:
        raise ImportError("the -relation.h5 files need h5py, which is not installed here; nothing emulates it") from exc
    return h5py


def write_relation_h5(path, trackid, pairs, feats, iou):
    """One segment's `-relation.h5` with the four datasets the reference reads (vrdataset.py:203-212)."""
    h5py = _h5py()
    with h5py.File(path, "w") as fout:
        fout.create_dataset("trackid", data=np.asarray(trackid))
        fout.create_dataset("pairs", data=np.asarray(pairs))
        fout.create_dataset("feats", data=np.asarray(feats, dtype=np.float32))
        fout.create_dataset("iou", data=np.asarray(iou, dtype=np.float32))


def read_relation_h5(path):
    """(pairs, feats, iou, trackid) as `VRDataset._get_rel_feature` returns them (vrdataset.py:190-217), or
    None if the file is absent; feed them to `proposal_pair_list`."""
    if not os.path.exists(path):
        return None
    h5py = _h5py()
    with h5py.File(path, "r") as fin:
        return fin["pairs"][:], fin["feats"][:], fin["iou"][:], fin["trackid"][:]


# ---- host-resident batches (what the reference's loader yields) one step ahead on the device -------------------
class DevicePrefetcher:
    """Wraps the reference's test / train loader (an iterable of `(pair_list, target_list, indexs)` batches with HOST
    tensors, lib/dataset/build.py:84-93, consumed at lib/modeling/predict.py:50-57):

        for pair_list, target_list, indexs in DevicePrefetcher(data_loader, device):
            pair_proposals, duration_proposals, rel_logits = model(pair_list, target_list)

    yields the same batches with every tensor already in HBM, and uploads batch i+1 on a copy stream while the caller
    works on batch i - the whole-batch upload (39 MB per cfg2 video) hides under the previous batch's encoder, and
    `forward` runs its resident path (one launch over the whole batch; results stay on the device, where `decode`
    wants them).  Pinned host tensors (`DataLoader(pin_memory=True)` -> `PairList.pin_memory()`) are DMA'd in place;
    pageable ones are first copied into pinned memory by a few copy workers.  Ordinary stream semantics for the
    consumer: the yielded tensors are ready on the CURRENT stream of `device` at the time of the `next()`."""

    def __init__(self, loader, device=None):
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher needs a HIP device")
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _pin(self, t):
        from .model import _stage_pool
        if t.is_pinned():
            return t
        out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        src = t if t.is_contiguous() else t.contiguous()
        if t.numel() * t.element_size() < (1 << 20) or t.dim() == 0:
            out.copy_(src)
            return out
        pool, nt = _stage_pool()
        rows = t.shape[0]
        step = max(1, -(-rows // nt))
        list(pool.map(lambda r: out[r:r + step].copy_(src[r:r + step]), range(0, rows, step)))
        return out

    def _upload(self, obj, keep):
        """Mirror `obj` (tensor / PairList-like / list / tuple / dict) with host tensors replaced by device copies issued
        on the copy stream; `keep` collects the pinned sources so that they outlive their DMA."""
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                return obj
            pinned = self._pin(obj)
            keep.append(pinned)
            return pinned.to(self.device, non_blocking=True)
        if hasattr(obj, "extra_fields") and hasattr(obj, "_derive"):           # PairList / TargetList of this package
            return obj._derive(self._upload(obj._primary_value(), keep), lambda v: self._upload(v, keep))
        if hasattr(obj, "extra_fields") and hasattr(obj, "to"):                # the reference's own list types
            out = obj.to(self.device)
            return out
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._upload(v, keep) for v in obj)
        if isinstance(obj, dict):
            return {k: self._upload(v, keep) for k, v in obj.items()}
        return obj                                                               # numpy fields, ints, index triples

    def _device_tensors(self, obj, out):
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                out.append(obj)
        elif hasattr(obj, "extra_fields"):
            if hasattr(obj, "_primary"):
                self._device_tensors(getattr(obj, obj._primary), out)
                self._device_tensors(list(obj.extra_fields.values()), out)
            else:
                # the reference's own list types (lib/dataset/list_pair.py:3-31): the primary tensor sits under an
                # attribute name this package does not know -- take every attribute (extra_fields included)
                self._device_tensors(list(vars(obj).values()), out)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                self._device_tensors(v, out)
        elif isinstance(obj, dict):
            self._device_tensors(list(obj.values()), out)

    def _stage(self, batch):
        keep = []
        with torch.cuda.stream(self._stream):
            dev_batch = self._upload(batch, keep)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return dev_batch, ev, keep

    def __iter__(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        held = None
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._stage(next(it))        # batch i+1 starts uploading before batch i is handed out
            except StopIteration:
                nxt = None
            dev_batch, ev, keep = cur
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            tensors = []
            self._device_tensors(dev_batch, tensors)
            for t in tensors:
                t.record_stream(main)              # allocated under the copy stream, consumed on the caller's
            held = keep                            # pinned sources of the batch in flight stay alive one more round
            yield dev_batch
        del held
