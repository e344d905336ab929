"""Input builders shared by make_golden.py (which feeds them to the reference) and the
tests (which feed them to the oracle / the HIP path).  Everything is regenerated from the
hash RNG, so the committed .npz files hold only the reference's outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tspn_mi355x as tspn  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# effective values of the reference's configs/baseline.yaml on top of lib/config/defaults.py
# (SURVEY.md §5 "Config / flags"); the tests must not read /root/reference.
BASELINE_OVERRIDES = {
    "DATASET.TRAIN_BATCH_SIZE": 1, "DATASET.TEST_BATCH_SIZE": 1, "DATASET.LOGIT_ONLY": True,
    "RELPN.USE_PPN": False, "RELPN.USE_DPN": False,
    "RELPN.PPN.BATCH_SIZE_PER_SEGMENT": 256, "RELPN.PPN.POSITIVE_FRACTION": 0.25,
    "RELPN.PPN.NUM_PAIR_PROPOSALS": 256, "RELPN.DPN.NUM_DURATION_PROPOSALS": 64,
    "PREDICT.TOPK_PER_PAIR": 20, "PREDICT.TOPK_PER_SEG": 200,
}


def baseline_cfg(**extra):
    over = dict(BASELINE_OVERRIDES)
    over.update(extra)
    return tspn.load_cfg(None, **over)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def ref_pairs(n):
    return np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int64)


def g1_inputs():
    """cfg1: N=8 -> P=56 rows of the 11070-d baseline feature, one zero-norm block."""
    n, p, f, k = 8, 56, 11070, 132
    raw = tspn.synth.make_baseline_features(1, p, f)
    raw[3, 70:1070] = 0.0
    sd = tspn.synth.make_weights(0, c=1024, k=k, feat_dim=f, bias_std=0.02)
    cls = tspn.synth.make_video(1, n, 2, 2)["track_cls_logits"]
    tgt = (tspn.hashrng.uniform(1, "targets", (p, k)) < 0.02).astype(np.float32)
    return {"n": n, "raw": raw, "state_dict": sd, "cls": cls, "targets": tgt}


def g2_inputs(seed):
    n, f, k = 32, 64, 132
    sd = tspn.synth.make_weights(0, c=1024, k=k, feat_dim=f, bias_std=0.02)
    feats = tspn.hashrng.uniform(2, "ppn_feats", (n * (n - 1), f))
    cls = 8.0 * tspn.synth.make_video(int(seed), n, 2, 2)["track_cls_logits"]
    tgt = (tspn.hashrng.uniform(2, "targets", (n * (n - 1), k)) < 0.01).astype(np.float32)
    return {"n": n, "feats": feats, "state_dict": sd, "cls": cls, "targets": tgt,
            "pairs": ref_pairs(n)}


G3_SHAPES = {"small": (56, 64, 30, 3), "mid": (96, 128, 150, 4)}


def g3_inputs(tag):
    p, c, t, seed = G3_SHAPES[tag]
    sd = tspn.synth.make_weights(0, c=c, bias_std=0.02)
    x = tspn.hashrng.uniform(seed, "dpn_x", (p, c, t))
    return {"x": x, "state_dict": sd, "c": c}


def g4_inputs():
    return (tspn.synth.make_video(5, 32, 150, 2)["tracklet_boxes"],
            tspn.synth.make_video(6, 12, 150, 2)["tracklet_boxes"])


G5_SPECS = [((15, 30, 45, 60), 7.5, 60), ((8, 16, 32, 64), 8, 150), ((4, 8, 16), 8, 30)]


def g6_inputs():
    n, k = 12, 132
    p = n * (n - 1)
    perm = np.argsort(tspn.hashrng.bits(7, "decode_logits", p * k), kind="stable")
    rel_logit = ((perm.astype(np.float64) + 0.5) / (p * k)).astype(np.float32).reshape(p, k)
    feat70 = tspn.hashrng.uniform(7, "decode_feat70", (p, 70))
    return {"n": n, "rel_logit": rel_logit, "feat70": feat70, "pairs": ref_pairs(n)}


def full_pairs(m):
    """All ordered pairs among m tracks, i-major — the `pairs` dataset of a -relation.h5 file
    (vrdataset.py:203-212: "all possible pairs among N+M object trajectories")."""
    return ref_pairs(m)


def g10_tables():
    """(pairs, trackid) cases for VRDataset._get_proposal_idx / _get_num_tracklet_proposals:
    proposals first then ground truth (the layout of the h5 files), interleaved, all proposals, no
    proposals, a single proposal, and a shuffled pair table."""
    out = []
    out.append((full_pairs(10), np.array([-1] * 8 + [0, 1], dtype=np.int64)))
    out.append((full_pairs(9), np.array([-1, 3, -1, -1, 0, -1, 2, -1, -1], dtype=np.int64)))
    out.append((full_pairs(6), -np.ones(6, dtype=np.int64)))
    out.append((full_pairs(4), np.arange(4, dtype=np.int64)))
    out.append((full_pairs(3), np.array([-1, 0, 1], dtype=np.int64)))
    perm = np.argsort(tspn.hashrng.bits(10, "pair_perm", 12 * 11), kind="stable")
    out.append((full_pairs(12)[perm], np.array([-1, -1, 5, -1, -1, -1, 0, -1, -1, 2, -1, -1], dtype=np.int64)))
    return out


def g10_segments():
    """Three cfg1-shaped segments as the -relation.h5 files hold them (N proposals followed by G
    ground-truth tracks, all ordered pairs, RAW 11070-d features), plus the cfg1 weights."""
    f, k = 11070, 132
    sd = tspn.synth.make_weights(0, c=1024, k=k, feat_dim=f, bias_std=0.02)
    segs, indexs = [], []
    for s, (n, g) in enumerate(((8, 2), (5, 1), (1, 2))):
        m = n + g
        pairs = full_pairs(m)
        raw = tspn.synth.make_baseline_features(20 + s, pairs.shape[0], f)
        raw[1, 1070:2070] = 0.0
        trackid = np.array([-1] * n + list(range(g)), dtype=np.int64)
        cls = tspn.synth.make_video(20 + s, n, 2, 2)["track_cls_logits"]
        iou = tspn.hashrng.uniform(20 + s, "iou", (m, m))
        segs.append({"n": n, "pairs": pairs, "raw": raw, "trackid": trackid, "cls": cls, "iou": iou})
        indexs.append((f"vid{s}", 15 * s, 15 * s + 30))
    return {"state_dict": sd, "segments": segs, "indexs": indexs}


def g9_scenario(seed=11, n_seg=5, n_trk=7, n_pred=40):
    """Synthetic multi-segment video for the association: 30-frame segments with stride 15
    (lib/modeling/__init__.py:35-41), tracklets that mostly continue from segment to segment (so the
    overlap IoU is high), a few that jump, predictions that repeat triplets and share tracklets (so the
    in-place trajectory aliasing of the reference is exercised), and score ties.
    Returns (short_term_relations list, trajectories dict keyed by (vid, fstart, fend))."""
    rs = np.random.RandomState(seed)
    vid = "vid0"
    total = 30 + 15 * (n_seg - 1)
    base = np.zeros((n_trk, total, 4))
    for k in range(n_trk):
        x, y = rs.randint(0, 400), rs.randint(0, 300)
        w, h = rs.randint(40, 200), rs.randint(40, 200)
        dx, dy = rs.uniform(-2, 2), rs.uniform(-2, 2)
        f = np.arange(total)
        base[k, :, 0] = np.round(x + dx * f)
        base[k, :, 1] = np.round(y + dy * f)
        base[k, :, 2] = base[k, :, 0] + w
        base[k, :, 3] = base[k, :, 1] + h
    rels, trajs = [], {}
    for s in range(n_seg):
        fs, fe = 15 * s, 15 * s + 30
        boxes = base[:, fs:fe].copy() + rs.randint(-3, 4, size=(n_trk, 30, 4))   # detector jitter
        perm = np.arange(n_trk)
        if s % 2 == 1:
            perm = np.roll(perm, 1)                                              # tracklet ids change
        boxes = boxes[perm]
        if s >= 2:
            boxes[0] += 500                                                      # one track jumps away
        trajs[(vid, fs, fe)] = boxes
        inv = np.argsort(perm)
        preds = []
        for _ in range(n_pred):
            a, b = rs.choice(n_trk, 2, replace=False)                            # base-track ids
            trip = np.array([a % 5, (a * 3 + b) % 6, b % 5])
            score = np.array(np.round(rs.uniform(0.05, 0.95), 2))                # 2 decimals: ties occur
            preds.append((score, trip, np.array([inv[a], inv[b]])))
        rels.append(((vid, fs, fe), (preds, np.zeros((n_trk, n_trk)), -np.ones(n_trk))))
    order = rs.permutation(n_seg)                                                # unsorted on purpose
    return [rels[i] for i in order], trajs
