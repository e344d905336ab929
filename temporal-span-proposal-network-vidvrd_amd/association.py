"""Greedy relational association of short-term relation predictions into video-level relations.

Host-side mirror of the reference's `lib/modeling/association.py` (same entry point, same argument
meaning, same output dicts) without dlib: trajectories are plain lists of (left, top, right, bottom)
float tuples.  It is downstream of the GPU path: the inputs are the per-segment
top-k triplets of `BaseModel.decode` / predict.py:59-117.  The greedy loop is sequential by construction
and stays on the host; its arithmetic - the trajectory IoUs, two per (prediction, candidate relation) in
the reference - is batched: with `device=` every live trajectory of the previous segment's relations
meets every tracklet of the current segment in ONE launch of `tspn_traj_iou_tail_f64` per segment
(csrc/tspn_iou.hip, the reference chain's float32 / float64 roundings kept to the bit), and only rows
whose trajectory a merge has changed since are recomputed, again in one launch.  Two things the
reference does per prediction are done once per segment because they cannot change in between: the
stable sort of the previous segment's relations by mean confidence (a relation's confidences change only
when it is extended, and then it leaves that list), and the scan for relations with the same triplet
(grouped once, order kept).  Without `device` the same loop calls the host IoU per candidate.

Behaviour reproduced as is (each checked against the reference's own code by tests/golden/g9):
  * segments are visited in order of `int(fstart)`; per segment the predictions are sorted by score,
    descending and stable, and cut to `max_traj_num_in_clip` (association.py:117-130);
  * in the first segment every prediction opens a relation with its score; in later segments a
    prediction is merged into the best-scoring (mean confidence, stable order) relation modified in the
    PREVIOUS segment that has the same triplet, starts before that relation's end and overlaps both
    trajectories with IoU >= 0.5 on the common frames (association.py:141-170); otherwise it opens a
    relation whose confidence list starts with 1 — the constructor default, not the prediction's
    score (association.py:166, 66);
  * merging averages the boxes over the overlap and appends the rest (association.py:16-31), IN PLACE
    on the trajectory object of the earlier segment — relations that share a tracklet share the object,
    so one merge lengthens the trajectory seen by the others (association.py:101-106); `fend` of a
    relation is only refreshed when that relation itself is extended, from the OBJECT trajectory;
  * the IoU arithmetic follows lib/modeling/trajectory.py:85-141: intersections in float32 (+1
    inclusive pixels), areas in float64, quotient stored as float32.
"""
import json
import os

import numpy as np

__all__ = ["Track", "VideoRelation", "greedy_relational_association", "load_trajectories", "AssociationWorker",
           "short_term_relations_from_arrays"]


class Track:
    """Box trajectory over frames [pstart, pend) (reference lib/modeling/trajectory.py:12-83).  `rois` is a float64
    array [length, 4] of (left, top, right, bottom) (round 6; a list of tuples before: building, slicing, merging and
    serialising 4 000 of them per video was most of the association's time)."""

    __slots__ = ("pstart", "pend", "rois", "score", "category", "classeme", "vsig", "gt_trackid", "_ver", "_ser")

    def __init__(self, pstart, pend, rois, score=0.0, category=-1, classeme=(), vsig=None, gt_trackid=-1):
        rois = np.array(rois, dtype=np.float64).reshape(-1, 4)     # always a private copy: merges write into it
        if rois.shape[0] != pend - pstart:
            raise ValueError(f"Track: {rois.shape[0]} boxes for frames [{pstart}, {pend})")
        self.pstart, self.pend, self.rois = int(pstart), int(pend), rois
        self.score, self.category, self.classeme = score, category, classeme
        self.vsig, self.gt_trackid = vsig, gt_trackid
        self._ver = 0          # bumped by every in-place merge: the batched IoU rows of this trajectory are stale
        self._ser = None       # (version, serialised boxes)

    def length(self):
        return self.pend - self.pstart

    def serialize_rois(self):
        """[(l, t, r, b), ...] of Python floats: a new list per call (as in the reference), converted once per trajectory
        version -- relations that share a trajectory share the (immutable) tuples, not the list."""
        if self._ser is None or self._ser[0] != self._ver:
            self._ser = (self._ver, list(map(tuple, self.rois.tolist())))
        return list(self._ser[1])


def _cubic_iou_1x1(boxes1, boxes2):
    """Volumetric IoU of one trajectory against one ([L,4] float64 each) with the roundings of
    trajectory.py:85-141: per-frame overlap extents and products in float32 (+1 inclusive pixels),
    accumulated frame by frame in float32; areas and their sums in float64; the quotient is formed in
    float64 and stored as float32."""
    b1 = np.asarray(boxes1, dtype=np.float64).reshape(-1, 4)
    b2 = np.asarray(boxes2, dtype=np.float64).reshape(-1, 4)
    if b1.shape[0] != b2.shape[0]:
        raise AssertionError("trajectories of different length")  # trajectory.py:89
    f32 = np.float32
    one = f32(1)
    w = np.clip((np.minimum(b1[:, 2], b2[:, 2]).astype(f32) + one) - np.maximum(b1[:, 0], b2[:, 0]).astype(f32), 0, None)
    h = np.clip((np.minimum(b1[:, 3], b2[:, 3]).astype(f32) + one) - np.maximum(b1[:, 1], b2[:, 1]).astype(f32), 0, None)
    inter = np.cumsum(w * h, dtype=f32)[-1] if b1.shape[0] else f32(0)   # cumsum = strictly sequential
    area = lambda b: np.sum((b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1))   # noqa: E731
    union = (area(b1) + area(b2)) - np.float64(inter)
    return f32(np.float64(inter) / union)


def _cubic_iou_1xn(a, b):
    """`_cubic_iou_1x1` of one trajectory a [k,4] against n trajectories b [n,k,4] at once -> float32 [n], the same
    roundings in the same order: per-frame overlaps and products in float32, summed frame by frame in float32 (cumsum
    along the contiguous axis), areas and their per-trajectory sums in float64 (np.sum over a contiguous row = the
    1-D pairwise sum), quotient in float64, stored as float32."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, a.shape[0], 4)
    f32 = np.float32
    if a.shape[0] == 0:
        raise ValueError("_cubic_iou_1xn: no common frames (the caller returns 0 there, association.py:36-37)")
    one = f32(1)
    w = np.clip((np.minimum(a[None, :, 2], b[:, :, 2]).astype(f32) + one) - np.maximum(a[None, :, 0], b[:, :, 0]).astype(f32), 0, None)
    h = np.clip((np.minimum(a[None, :, 3], b[:, :, 3]).astype(f32) + one) - np.maximum(a[None, :, 1], b[:, :, 1]).astype(f32), 0, None)
    inter = np.cumsum(np.ascontiguousarray(w * h), axis=1, dtype=f32)[:, -1]
    area_a = np.sum((a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1))
    area_b = np.sum(np.ascontiguousarray((b[:, :, 2] - b[:, :, 0] + 1) * (b[:, :, 3] - b[:, :, 1] + 1)), axis=1)
    union = (area_a + area_b) - inter.astype(np.float64)
    return (inter.astype(np.float64) / union).astype(f32)


def _traj_iou(t1, t2):
    """IoU on the common frames of two trajectories (association.py:34-50)."""
    if t1.pend <= t2.pstart or t2.pend <= t1.pstart:
        return 0
    if t1.pstart > t2.pstart:
        t1, t2 = t2, t1
    a = t1.rois[t2.pstart - t1.pstart:t1.pend - t1.pstart]
    b = t2.rois[0:t1.pend - t2.pstart]
    return _cubic_iou_1x1(a, b)


def _merge_trajs(t1, t2):
    """association.py:16-31: average over the overlap, append the rest; t1 is modified in place."""
    overlap = max(t1.pend - t2.pstart, 0)
    n1, n2 = t1.length(), t2.length()
    if overlap > n1 or overlap > n2:
        # (only with segments out of order, which the sort by fstart excludes: the reference's index arithmetic then runs off
        # the end of t2 -- IndexError, as here -- or wraps around inside t1)
        raise IndexError("merge of trajectories whose overlap exceeds one of them")
    if overlap:
        t1.rois[n1 - overlap:n1] = (t1.rois[n1 - overlap:n1] + t2.rois[:overlap]) / 2
    if n2 > overlap:
        t1.rois = np.concatenate([t1.rois, t2.rois[overlap:]])
        t1.pend += n2 - overlap
    t1._ver += 1
    return t1


class VideoRelation:
    """association.py:53-112."""

    def __init__(self, vid, s_cid, pid, o_cid, straj, otraj, confs=1):
        self.vid, self.s_cid, self.pid, self.o_cid = vid, s_cid, pid, o_cid
        self.straj, self.otraj = straj, otraj
        self.confs_list = [confs]
        self.fstart, self.fend = straj.pstart, straj.pend

    def triplet(self):
        return (self.s_cid, self.pid, self.o_cid)

    def mean_confs(self):
        # np.mean as in the reference (association.py:84-85; its summation order decides ties), cached per list length
        n = len(self.confs_list)
        if getattr(self, "_mean_n", -1) != n:
            self._mean, self._mean_n = np.mean(self.confs_list), n
        return self._mean

    def both_overlap(self, straj, otraj, iou_thr=0.5, table=None, s_idx=None, o_idx=None):
        if table is not None:      # batched device IoUs of this segment (same values, same short circuit)
            return bool(table.iou(self.straj, s_idx) >= iou_thr and table.iou(self.otraj, o_idx) >= iou_thr)
        return bool(_traj_iou(self.straj, straj) >= iou_thr and _traj_iou(self.otraj, otraj) >= iou_thr)

    def extend(self, straj, otraj, confs):
        self.straj = _merge_trajs(self.straj, straj)
        self.otraj = _merge_trajs(self.otraj, otraj)
        self.confs_list.append(confs)
        self.fstart = self.straj.pstart
        self.fend = self.otraj.pend

    def serialize(self, dataset=None):
        name = (lambda f, i: getattr(dataset, f)(i)) if dataset is not None else (lambda f, i: int(i))
        return {"triplet": [name("get_object_name", self.s_cid), name("get_predicate_name", self.pid),
                            name("get_object_name", self.o_cid)],
                "score": float(self.mean_confs()),
                "duration": [int(self.fstart), int(self.fend)],
                "sub_traj": self.straj.serialize_rois(),
                "obj_traj": self.otraj.serialize_rois()}


def load_trajectories(vid, fstart, fend, root="./vidvrd-baseline-output"):
    """The reference's on-disk tracklet proposals of a segment (`traj_cls` JSON written by its
    preprocessing; lib/modeling/trajectory.py:169-196, path from lib/modeling/__init__.py:6-22)."""
    vsig = "{}-{:04d}-{:04d}".format(vid, fstart, fend)
    path = os.path.join(root, "features", "traj_cls", vid, f"{vsig}-traj_cls.json")
    if not os.path.exists(path):
        return []
    with open(path, "r") as fin:
        return [Track(**t) for t in json.load(fin)]


def _as_tracks(trajs, fstart, fend):
    out = []
    for t in trajs:
        if isinstance(t, Track):
            out.append(t)
        elif isinstance(t, dict):
            out.append(Track(**t))
        else:  # [L,4] boxes
            out.append(Track(fstart, fend, t))
    return out


class _SegmentIoUTable:
    """IoUs of a segment: (live trajectory of an earlier segment) x (tracklet of this segment), on their common
    frames, computed on the device in one launch for all pairs (`ops.traj_iou_tail`) and again - one launch for
    every stale row at once - when a merge has changed trajectories since (up to HOST_ROWS stale rows: on the host,
    vectorised, `_cubic_iou_1xn`).  Values equal `_traj_iou`'s bit for bit."""

    HOST_ROWS = 16

    def __init__(self, device, cur_tracks, fstart, fend):
        import torch

        from . import ops
        self._torch, self._ops = torch, ops
        self.device = torch.device(device)
        self.fstart, self.fend, self.L = int(fstart), int(fend), int(fend) - int(fstart)
        b = np.empty((len(cur_tracks), self.L, 4), dtype=np.float64)
        for j, t in enumerate(cur_tracks):
            if t.rois.shape[0] != self.L:
                raise ValueError(f"tracklet {j} of segment [{fstart}, {fend}) has {t.rois.shape[0]} boxes")
            b[j] = t.rois
        self.n = len(cur_tracks)
        self.b_host = b
        self.b = torch.from_numpy(b).to(self.device)
        self.live = {}         # id(track) -> track
        self.rows = {}         # id(track) -> (version, float32 [N] or AssertionError)
        self.launches = 0
        self.host_rows = 0

    def add(self, tracks):
        for t in tracks:
            self.live[id(t)] = t

    def _refresh(self):
        stale = [t for k, t in self.live.items() if self.rows.get(k, (None,))[0] != t._ver]
        if not stale or self.n == 0:
            for t in stale:
                self.rows[id(t)] = (t._ver, np.zeros((0,), dtype=np.float32))
            return
        if self.launches > 0 and len(stale) <= self.HOST_ROWS:
            # a few rows gone stale through merges: on the host, vectorised over the segment's tracklets -- same bits
            # (_cubic_iou_1xn), no launch and no synchronisation (round 6: 323 -> 59 launches per VidOR-scale video)
            for t in stale:
                if t.pstart > self.fstart:
                    raise AssertionError("segments out of order")
                k = t.pend - self.fstart
                if k <= 0:
                    row = np.zeros((self.n,), dtype=np.float32)
                elif k > self.L:
                    row = AssertionError("trajectories of different length")
                else:
                    row = _cubic_iou_1xn(t.rois[self.fstart - t.pstart:t.pend - t.pstart], self.b_host[:, :k])
                self.rows[id(t)] = (t._ver, row)
            self.host_rows += len(stale)
            return
        torch = self._torch
        a = np.zeros((len(stale), self.L, 4), dtype=np.float64)
        ln = np.zeros((len(stale),), dtype=np.int32)
        bad = {}
        for u, t in enumerate(stale):
            if t.pstart > self.fstart:
                raise AssertionError("segments out of order")       # cannot happen after the sort by fstart
            k = t.pend - self.fstart                                 # common frames [fstart, t.pend)
            if k <= 0:
                continue                                             # no overlap -> 0 (association.py:36-37)
            if k > self.L:
                bad[u] = AssertionError("trajectories of different length")    # trajectory.py:89, raised on use
                continue
            a[u, :k] = t.rois[self.fstart - t.pstart:t.pend - t.pstart]
            ln[u] = k
        out = self._ops.traj_iou_tail(torch.from_numpy(a).to(self.device), torch.from_numpy(ln).to(self.device), self.b)
        out = out.cpu().numpy()
        self.launches += 1
        for u, t in enumerate(stale):
            self.rows[id(t)] = (t._ver, bad.get(u, out[u]))

    def iou(self, track, j):
        ent = self.rows.get(id(track))
        if ent is None or ent[0] != track._ver:
            self.live[id(track)] = track
            self._refresh()
            ent = self.rows[id(track)]
        if isinstance(ent[1], AssertionError):
            raise ent[1]
        return ent[1][int(j)]


def greedy_relational_association(dataset, short_term_relations, max_traj_num_in_clip=100, trajectories=None,
                                  device=None, stats=None):
    """Reference `greedy_relational_association` (association.py:117-175).

    `short_term_relations`: list of `((vid, fstart, fend), (pred_list, iou, trackid))` with
    `pred_list` = [(score, (s_cid, pid, o_cid), (s_idx, o_idx)), ...] as predict.py:106-116 emits.
    `trajectories`: callable `(vid, fstart, fend) -> tracklets` (Track objects, `traj_cls` dicts or
    [L,4] box arrays, indexed like the predictions' tracklet ids) or a dict keyed by that triple;
    default = the reference's on-disk proposals (`load_trajectories`).
    `device`: a HIP device ("cuda", "cuda:0", torch.device): the trajectory IoUs of a segment come from one
    batched launch (module docstring) instead of one numpy evaluation per (prediction, candidate); same results.
    `stats`: optional dict, receives counters (`iou_launches`, `iou_lookups`, `iou_host_rows`, `segments`).
    Returns the list of serialised video relations (`dataset` supplies the names; None keeps ids)."""
    if trajectories is None:
        provider = load_trajectories
    elif callable(trajectories):
        provider = trajectories
    else:
        provider = lambda vid, fs, fe: trajectories[(vid, fs, fe)]  # noqa: E731
    short_term_relations.sort(key=lambda x: int(x[0][1]))   # in place, like the reference
    video_relation_list = []
    last_modify_rel_list = []
    launches = lookups = host_rows = 0
    for i, (index, prediction) in enumerate(short_term_relations):
        vid, fstart, fend = index
        pred_list = prediction[0]
        sorted_pred_list = sorted(pred_list, key=lambda x: x[0], reverse=True)[:max_traj_num_in_clip]
        trajs = _as_tracks(provider(vid, fstart, fend), fstart, fend)
        for traj in trajs:
            traj.pstart, traj.pend = fstart, fend
            traj.vsig = "{}-{:04d}-{:04d}".format(vid, fstart, fend)
        cur_modify_rel_list = []
        # the reference re-sorts the previous segment's relations before EVERY prediction (association.py:150); the
        # keys of the relations still in that list cannot change inside a segment (an extended relation leaves it), and
        # the sort is stable, so once per segment gives the same order; likewise the same-triplet scan, grouped once
        by_triplet = {}
        table = None
        if i > 0 and sorted_pred_list:
            last_modify_rel_list.sort(key=lambda r: r.mean_confs(), reverse=True)
            for r in last_modify_rel_list:
                by_triplet.setdefault(_triplet_key(r.triplet()), []).append(r)
            if device is not None and last_modify_rel_list:
                table = _SegmentIoUTable(device, trajs, fstart, fend)
                wanted = {_triplet_key(p[1]) for p in sorted_pred_list}
                for key in wanted:                                   # only relations a prediction can meet
                    for r in by_triplet.get(key, ()):
                        table.add((r.straj, r.otraj))
        for pred in sorted_pred_list:
            conf_score = pred[0]
            s_cid, pid, o_cid = pred[1]
            s_idx, o_idx = pred[2]
            straj, otraj = trajs[int(s_idx)], trajs[int(o_idx)]
            if i == 0:
                r = VideoRelation(vid, s_cid, pid, o_cid, straj, otraj, confs=conf_score)
                video_relation_list.append(r)
                cur_modify_rel_list.append(r)
                continue
            merged = False
            candidates = by_triplet.get(_triplet_key(pred[1]), ())
            for r in candidates:
                if straj.pstart < r.fend and otraj.pstart < r.fend:
                    lookups += 1
                    if r.both_overlap(straj, otraj, table=table, s_idx=s_idx, o_idx=o_idx):
                        r.extend(straj, otraj, conf_score)
                        candidates.remove(r)
                        cur_modify_rel_list.append(r)
                        merged = True
                        break
            if not merged:
                r = VideoRelation(vid, s_cid, pid, o_cid, straj, otraj)   # confs = 1 (reference default)
                video_relation_list.append(r)
                cur_modify_rel_list.append(r)
        if table is not None:
            launches += table.launches
            host_rows += table.host_rows
        last_modify_rel_list = cur_modify_rel_list
    if stats is not None:
        stats.update(iou_launches=launches, iou_lookups=lookups, iou_host_rows=host_rows, segments=len(short_term_relations))
    return [rel.serialize(dataset) for rel in video_relation_list]


def _triplet_key(triplet):
    """Hashable form of a prediction's / relation's (s_cid, pid, o_cid): the reference compares them element-wise
    with numpy (association.py:152), so 3, np.int64(3) and a 0-d array holding 3 are the same triplet."""
    return tuple(np.asarray(triplet).tolist())


# ------------------------------------------------------------------------------------------------------------------
# The association beside the GPU pipeline (round 6).  The greedy loop is ~0.3 s of pure Python per VidOR-scale video
# (59 segments x 200 predictions, 11 000 relations) -- about the time the GPU needs for the next video's frames ->
# triplets.  On a THREAD of the process that drives the GPU it holds the interpreter lock while that process issues its
# ~2 000 launches per video: measured 489 ms per video against 385 without it (profiles/r6/cfg5_associate.md).  Videos are
# independent, so it runs in a worker PROCESS: the driver hands over a video's short-term relations (a few MB, pickled)
# and carries on.
def short_term_relations_from_arrays(segments, scores, triplets, pairs, boxes=None):
    """The decoded top-k of a video's segments as arrays -> the structures `greedy_relational_association` takes.
    segments: [(vid, fstart, fend)] x S; scores [S][K]; triplets [S][K,3]; pairs [S][K,2] (tracklet ids inside the segment),
    as `BaseModel.decode` / predict.py:106-116 produce them per segment; boxes (optional) [S][N,L,4] -> the trajectories
    dict.  Arrays keep their dtypes (a relation's score is the mean of the scores it absorbed).  Returns
    (short_term_relations, trajectories | None)."""
    rels, trajs = [], ({} if boxes is not None else None)
    for s, index in enumerate(segments):
        index = (index[0], int(index[1]), int(index[2]))
        sc, tr, pr = np.asarray(scores[s]), np.asarray(triplets[s]), np.asarray(pairs[s])
        if tr.shape != (sc.shape[0], 3) or pr.shape != (sc.shape[0], 2):
            raise ValueError(f"segment {index}: scores {sc.shape}, triplets {tr.shape}, pairs {pr.shape} do not belong together")
        rels.append((index, (list(zip(sc, tr, pr)), None, None)))
        if boxes is not None:
            trajs[index] = np.asarray(boxes[s], dtype=np.float64)
    return rels, trajs


def _worker_main(conn, device, max_traj_num_in_clip):
    import contextlib
    import time
    scope = contextlib.nullcontext
    if device is not None:
        import torch
        # the IoU launches are tiny and the host waits for each: a high-priority stream, so that they do not queue behind
        # the long kernels of the process that drives the GPU
        stream = torch.cuda.Stream(device=torch.device(device), priority=-1)
        scope = lambda: torch.cuda.stream(stream)              # noqa: E731
    while True:
        try:
            job = conn.recv()
        except EOFError:
            return
        if job is None:
            return
        key, rels, trajs, out_path, want = job
        t0 = time.perf_counter()
        try:
            stats = {}
            if isinstance(rels, dict):                          # submit_arrays: a handful of arrays instead of 10^4 tuples
                rels, trajs = short_term_relations_from_arrays(**rels)
            with scope():
                out = greedy_relational_association(None, rels, max_traj_num_in_clip=max_traj_num_in_clip,
                                                    trajectories=trajs, device=device, stats=stats)
            if out_path:
                with open(out_path, "w") as fh:                 # what the reference's driver stores per video (base.py:107-113)
                    json.dump(out, fh)
            conn.send((key, out if want else len(out), (time.perf_counter() - t0) * 1e3, stats, None))
        except Exception as exc:                                 # noqa: BLE001 -- reported to the caller, the worker lives on
            conn.send((key, None, (time.perf_counter() - t0) * 1e3, {}, f"{type(exc).__name__}: {exc}"))


def _worker_connect(address, authkey, device, max_traj_num_in_clip):
    """Entry point of the worker process (started by AssociationWorker)."""
    from multiprocessing.connection import Client
    with Client(address, family="AF_UNIX", authkey=authkey) as conn:
        _worker_main(conn, device, max_traj_num_in_clip)


class AssociationWorker:
    """`greedy_relational_association` of whole videos in a worker process (same results: it IS that function).

        w = AssociationWorker(device="cuda:0")            # device=None: host IoUs
        w.submit(vid, short_term_relations, trajectories) # returns at once; jobs run in order
        vid, relations, ms, stats = w.result()            # blocks (without the interpreter lock) until the oldest job is done

    `trajectories`: a dict keyed by (vid, fstart, fend) or None for the reference's on-disk proposals (a callable does not
    travel).  `out_path`: the worker also writes the relations as JSON; with `return_relations=False` only their number
    comes back (a VidOR-scale video's relations are ~30 MB of Python lists: leave them in the worker when the file is
    what is wanted).  The worker owns its own HIP context and stream when `device` is given."""

    def __init__(self, device=None, max_traj_num_in_clip=100):
        # A plain child process that connects back over a Unix socket -- not multiprocessing.Process: its spawn method
        # re-imports the parent's __main__ in the child (a second run of the driver script unless it is guarded; impossible
        # from an interactive parent), and fork would copy a process that holds a HIP context.
        import secrets
        import subprocess
        import sys
        import tempfile
        from multiprocessing.connection import Listener
        self._dir = tempfile.mkdtemp(prefix="tspn_assoc_")
        address = os.path.join(self._dir, "sock")
        authkey = secrets.token_bytes(16)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import sys; sys.path.insert(0, %r); import tspn_mi355x; "
                "tspn_mi355x.association._worker_connect(%r, bytes.fromhex(%r), %r, %d)"
                % (root, address, authkey.hex(), None if device is None else str(device), int(max_traj_num_in_clip)))
        with Listener(address, family="AF_UNIX", authkey=authkey) as listener:
            self._proc = subprocess.Popen([sys.executable, "-c", code])
            listener._listener._socket.settimeout(120)          # a child that dies before connecting must not hang us
            try:
                self._conn = listener.accept()
            except Exception:
                self._proc.kill()
                raise RuntimeError("AssociationWorker: the worker process did not start "
                                   f"(exit code {self._proc.poll()})") from None
        self._inflight = 0

    def submit(self, key, short_term_relations, trajectories=None, out_path=None, return_relations=True):
        if callable(trajectories):
            raise TypeError("AssociationWorker.submit: trajectories must be a dict keyed by (vid, fstart, fend) or None")
        self._conn.send((key, short_term_relations, trajectories, out_path, bool(return_relations)))
        self._inflight += 1

    def submit_arrays(self, key, segments, scores, triplets, pairs, boxes, out_path=None, return_relations=True):
        """`submit` with the video's decoded results as arrays (`short_term_relations_from_arrays` runs in the worker):
        pickling a VidOR-scale video's 12 000 predictions as tuples of small arrays costs the driving process ~80 ms
        and the worker as much again; as five arrays it is under a millisecond."""
        self._conn.send((key, {"segments": list(segments), "scores": scores, "triplets": triplets, "pairs": pairs,
                               "boxes": boxes}, None, out_path, bool(return_relations)))
        self._inflight += 1

    def result(self):
        if self._inflight == 0:
            raise RuntimeError("AssociationWorker.result: nothing in flight")
        try:
            key, out, ms, stats, err = self._conn.recv()
        except (EOFError, ConnectionError, OSError) as exc:
            self._inflight = 0
            raise RuntimeError(f"AssociationWorker: the worker process is gone (exit code {self._proc.poll()}): {exc}") from None
        self._inflight -= 1
        if err is not None:
            raise RuntimeError(f"association of {key!r} failed in the worker: {err}")
        return key, out, ms, stats

    def close(self):
        import shutil
        if getattr(self, "_closed", False) or not hasattr(self, "_proc"):
            return
        self._closed = True
        if self._proc.poll() is None:
            try:
                self._conn.send(None)
            except (BrokenPipeError, OSError):
                pass
            try:
                self._proc.wait(timeout=10)
            except Exception:                                    # noqa: BLE001
                self._proc.kill()
        self._conn.close()
        shutil.rmtree(self._dir, ignore_errors=True)

    def __del__(self):
        try:
            self.close()
        except Exception:                                        # noqa: BLE001 -- interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
