#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; it never ships to the
GPU box).  Inputs and weights come from the build-owned hash RNG
(temporal-span-proposal-network-vidvrd_amd/hashrng.py, synth.py), so only the
reference's OUTPUTS are stored; tests regenerate the inputs.

What is imported from the reference (torch + numpy only):
    lib.modeling.model            BaseModel, RelationPredictor
    lib.modeling.relpn.ppn        PPN, PPNHead
    lib.modeling.relpn.dpn        DPNHead
    relpn/dpn_anchor.py           DPNHead with relness_pred + duration_pred (the file uses sibling-relative imports
                                  `from rel_nms import ...`, so its directory goes on sys.path; g11)
    lib.modeling.relpn.anchor_generator   AnchorGenerator
    lib.modeling.relpn.sampler    BalancedPositiveNegativePairSampler
    lib.modeling.trajectory       cubic_iou
    lib.dataset.vrdataset         VRDataset._feature_preprocess, _get_proposal_idx,
                                  _get_num_tracklet_proposals
    lib.modeling.predict          predict   (the whole prediction loop + top-k decode)
    lib.dataset.list_pair / list_target   PairList, TargetList
    lib.modeling                  segment_video
Third-party modules the reference imports but does not use on these code paths
(dlib, h5py, IPython, torchvision) are absent here; empty module objects stand in for the
*import statement only*.  `np.float` (removed in numpy 1.24, used by
anchor_generator.py:72,80) is aliased to `float`.  The decode goldens (g6, g10) come from the reference's own
`lib.modeling.predict.predict()` run as a whole: only its checkpoint directory and its
data loader (the h5py / dataset side) are replaced, see `run_ref_predict`.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TSPN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402
sys.path.insert(0, HERE)
import cases  # noqa: E402

if not hasattr(np, "float"):
    np.float = float

class _Rect:
    """Stand-in for dlib.drectangle (dlib is absent here), used ONLY by g9: a value class with the
    accessors the reference's association / trajectory code calls on it.  The association's control
    flow, merging and IoU arithmetic that g9 pins are the reference's own code."""

    def __init__(self, left, top, right, bottom):
        self._v = (float(left), float(top), float(right), float(bottom))

    def left(self):
        return self._v[0]

    def top(self):
        return self._v[1]

    def right(self):
        return self._v[2]

    def bottom(self):
        return self._v[3]

    def width(self):
        return self._v[2] - self._v[0] + 1

    def height(self):
        return self._v[3] - self._v[1] + 1


for _name, _attrs in (("dlib", ("drectangle", "correlation_tracker")), ("h5py", ()), ("IPython", ()),
                      ("torchvision", ()), ("torchvision.transforms", ("functional",))):
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        for _a in _attrs:
            setattr(_m, _a, _Rect if _a == "drectangle" else object)
        sys.modules[_name] = _m
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

from lib.modeling import segment_video  # noqa: E402
from lib.modeling.model import BaseModel as RefBaseModel  # noqa: E402
from lib.modeling.model import RelationPredictor as RefRelationPredictor  # noqa: E402
from lib.modeling.relpn.dpn import DPNHead as RefDPNHead  # noqa: E402
from lib.modeling.relpn.anchor_generator import AnchorGenerator as RefAnchorGenerator  # noqa: E402
from lib.modeling.relpn.sampler import BalancedPositiveNegativePairSampler as RefSampler  # noqa: E402
from lib.modeling.trajectory import cubic_iou as ref_cubic_iou  # noqa: E402
from lib.modeling.trajectory import Trajectory as RefTrajectory  # noqa: E402
import lib.modeling.association as ref_association  # noqa: E402
from lib.dataset.list_pair import PairList as RefPairList  # noqa: E402
from lib.dataset.list_target import TargetList as RefTargetList  # noqa: E402
from lib.dataset.vrdataset import VRDataset as RefVRDataset  # noqa: E402
import lib.modeling.predict as ref_predict  # noqa: E402
sys.path.insert(0, os.path.join(REF, "lib", "modeling", "relpn"))   # dpn_anchor.py imports its siblings by bare name
from dpn_anchor import DPNHead as RefAnchorDPNHead  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}: " + ", ".join(f"{k}{tuple(np.asarray(v).shape)}" for k, v in arrays.items()))


def load_sd(model, sd, strict=True):
    sd = {k: t(v) for k, v in sd.items()}
    if not strict:
        own = model.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}
    model.load_state_dict(sd, strict=True)


def min_gap_desc(values):
    v = np.sort(np.asarray(values, dtype=np.float64).ravel())[::-1]
    return float(np.min(v[:-1] - v[1:]))


def ref_cfg(**over):
    """defaults restated in the build's config module + the reference's own configs/baseline.yaml."""
    cfg = tspn.load_cfg(os.path.join(REF, "configs", "baseline.yaml"), **over)
    chk = cases.baseline_cfg(**over)
    for sec in ("RELPN", "PREDICT", "DATASET"):  # the sections the hot path reads
        assert cfg[sec] == chk[sec], f"tests/golden/cases.py BASELINE_OVERRIDES drifted from configs/baseline.yaml ({sec})"
    return cfg


# --------------------------------------------------------------------------- #
def g1_baseline():
    """cfg1: configs/baseline.yaml, N=8 -> P=56, F=11070: BaseModel eval + train forward."""
    cfg = ref_cfg()
    c = cases.g1_inputs()
    feats = RefVRDataset._feature_preprocess(None, t(c["raw"].copy()))
    feats = (feats if isinstance(feats, torch.Tensor) else t(feats)).float()
    model = RefBaseModel(cfg)
    load_sd(model, c["state_dict"], strict=False)
    model.eval()
    plist = RefPairList(feats)
    plist.add_field("track_cls_logits", t(c["cls"]))
    with torch.no_grad():
        pp, dp, logits = model([plist], None)
    assert pp is None and dp is None
    model.train()
    loss = model([plist], [RefTargetList(t(c["targets"]))])
    save("g1_baseline_cfg1.npz", preprocessed_rows=feats[:4, :1200].numpy(),
         preprocessed_checksum=np.array([feats.double().sum().item(), feats.double().abs().max().item()]),
         rel_logits=logits[0].numpy(), loss_rel=np.array(loss["loss_rel"].item()))


def g2_ppn():
    """USE_PPN=True, N=32: pair matrix [32,32] + top-256 flat indices; plus the train-mode losses.
    Only the order of the first 256 entries (and the 256/257 boundary) is observable, so the first
    input seed whose top-257 values are separated by much more than fp32 GEMM noise is used."""
    cfg = ref_cfg(**{"RELPN.USE_PPN": True, "PREDICT.FEATURE_DIM": 64})
    model = RefBaseModel(cfg)
    for seed in range(2, 200):
        c = cases.g2_inputs(seed)
        load_sd(model, c["state_dict"], strict=False)
        model.eval()
        plist = RefPairList(t(c["feats"]))
        plist.add_field("track_cls_logits", t(c["cls"]))
        plist.add_field("tracklet_pairs", c["pairs"])
        plist.add_field("num_tracklets", c["n"])
        with torch.no_grad():
            pp, dp, logits = model([plist], None)
            mat = model.relpn.pair_proposal_network.ppn_head(t(c["cls"]), t(c["cls"]))
        top = np.sort(mat.numpy().astype(np.float64).ravel())[::-1][:257]
        gap = float(np.min(top[:-1] - top[1:]))
        if gap > 2e-5:
            break
    else:
        raise AssertionError("no tie-free PPN input found")
    stable = torch.sort(mat.view(-1), descending=True, stable=True)[1][:256]
    assert torch.equal(pp[0], stable)
    model.train()
    loss = model([plist], [RefTargetList(t(c["targets"]))])
    save("g2_ppn_n32.npz", pair_matrix=mat.numpy(), topk=pp[0].numpy(), min_gap=np.array(gap),
         input_seed=np.array(seed), rel_logits_head=logits[0][:8].numpy(),
         loss_pair=np.array(loss["loss_pair"].item()), loss_rel=np.array(loss["loss_rel"].item()))


def g3_dpn():
    """DPNHead (reference relpn/dpn.py:55-73) at small and mid shapes."""
    out = {}
    pre = "relpn.duration_proposal_network.dpn_head."
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        head = RefDPNHead(c["c"], 4)
        head.load_state_dict({k[len(pre):]: t(v) for k, v in c["state_dict"].items()
                              if k.startswith(pre) and "relness" not in k})
        with torch.no_grad():
            y = head(t(c["x"]))
        out[f"{tag}_duration"] = y.numpy()
    save("g3_dpn_head.npz", **out)


def g4_iou():
    """cubic_iou (reference trajectory.py:127-141), N=32, T=150, integer-valued boxes; and a
    second call with two different trajectory lists."""
    b, b2 = cases.g4_inputs()
    iou = ref_cubic_iou(b, b)
    iou12 = ref_cubic_iou(b, b2)
    assert iou.dtype == np.float32
    save("g4_cubic_iou.npz", iou=iou, iou_cross=iou12)


def g5_anchors():
    """AnchorGenerator (reference relpn/anchor_generator.py:31-64) for three (sizes, stride, T)."""
    out = {}
    for i, (sizes, stride, tw) in enumerate(cases.G5_SPECS):
        gen = RefAnchorGenerator(sizes, stride)
        out[f"anchors_{i}"] = gen(torch.zeros(2, 4, tw))[0].numpy()
    save("g5_anchors.npz", **out)


class _Log:
    def info(self, *a, **k):
        pass


def run_ref_predict(cfg, state_dict, batches, model_factory=None):
    """Run the reference's own `predict()` (lib/modeling/predict.py:14-123) as a whole.  Replaced,
    and nothing else: the checkpoint directory (`get_model_path`, a temporary directory holding a real
    `torch.save`d checkpoint that the reference's `torch.load` + `load_checkpoint` read) and the data
    loader (`build_data_loader`, which would open the h5py dataset) — `batches` is a list of
    `(pair_list, target_list, indexs)` as the reference's collate yields them.  `model_factory`
    replaces `BaseModel` when the golden needs the decode on given logits (g6)."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="tspn_golden_")
    saved = (ref_predict.get_model_path, ref_predict.build_data_loader, ref_predict.BaseModel)
    try:
        torch.save({"model": state_dict, "iter": 0, "loss": 0.0}, os.path.join(tmp, cfg.ETC.MODEL_DUMP_FILE))
        ref_predict.get_model_path = lambda: tmp
        ref_predict.build_data_loader = lambda *a, **k: batches
        if model_factory is not None:
            ref_predict.BaseModel = model_factory
        return ref_predict.predict(cfg, None, _Log())
    finally:
        ref_predict.get_model_path, ref_predict.build_data_loader, ref_predict.BaseModel = saved
        shutil.rmtree(tmp, ignore_errors=True)


def _predictions_arrays(preds):
    return (np.array([np.asarray(s) for s, _, _ in preds], dtype=np.float32),
            np.array([np.asarray(tr) for _, tr, _ in preds], dtype=np.int64).reshape(-1, 3),
            np.array([np.asarray(p) for _, _, p in preds], dtype=np.int64).reshape(-1, 2))


def g6_decode():
    """Top-k triplet decode = the reference's own predict() loop (predict.py:39-123) on tie-free
    logits, N=12.  The model is a stand-in that returns the given logits (the decode is what g6 pins;
    g10 runs the loop with the reference's BaseModel)."""
    c = cases.g6_inputs()
    rel_logit = t(c["rel_logit"])
    assert min_gap_desc(rel_logit.numpy()) > 0, "ties in decode input"

    class GivenLogits(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, pair_list, _):
            return None, None, [rel_logit]

    feat = torch.zeros(rel_logit.shape[0], 70)
    feat[:, :70] = t(c["feat70"])
    plist = RefPairList(feat)
    plist.add_field("tracklet_pairs", t(c["pairs"]))
    plist.add_field("track_cls_logits", torch.zeros(c["n"], 35))
    plist.add_field("num_tracklets", np.int64(c["n"]))
    plist.add_field("ious", np.zeros((c["n"], c["n"]), dtype=np.float32))
    plist.add_field("track_ids", -np.ones(c["n"], dtype=np.int64))
    index = ("g6", 0, 30)
    res = run_ref_predict(ref_cfg(), {"w": torch.zeros(1)}, [([plist], None, [index])], model_factory=GivenLogits)
    scores, trip, tids = _predictions_arrays(res[index][0])
    assert scores.shape == (200,)
    save("g6_decode.npz", scores=scores, triplets=trip, pair_tids=tids)


def g10_dataset_and_predict():
    """(a) VRDataset._get_proposal_idx / _get_num_tracklet_proposals (lib/dataset/vrdataset.py:140-148)
    on the pair tables of cases.g10_tables (proposal + ground-truth tracks mixed).
    (b) the reference's whole prediction path on cfg1-shaped segments: VRDataset's own pair filter and
    _feature_preprocess -> reference BaseModel (configs/baseline.yaml) -> predict()'s decode, for three
    segments (N = 8, 5 and 1 proposal tracklets; the last is skipped by predict.py:61-64)."""
    out = {}
    for i, (pairs, trackid) in enumerate(cases.g10_tables()):
        out[f"proposal_idx_{i}"] = np.array(RefVRDataset._get_proposal_idx(None, pairs, trackid), dtype=np.int64)
        out[f"num_tracks_{i}"] = np.array(int(RefVRDataset._get_num_tracklet_proposals(None, trackid)))
    cfg = ref_cfg()
    segs = cases.g10_segments()
    sd = {k: t(v) for k, v in segs["state_dict"].items()}
    own = RefBaseModel(cfg).state_dict()
    sd = {"module." + k: v for k, v in sd.items() if k in own}      # checkpoints are saved from DDP (train.py:114)
    batches = []
    for index, seg in zip(segs["indexs"], segs["segments"]):
        pairs, trackid, raw = seg["pairs"], seg["trackid"], seg["raw"]
        keep = RefVRDataset._get_proposal_idx(None, pairs, trackid)
        feats = RefVRDataset._feature_preprocess(None, torch.tensor(raw, dtype=torch.float32)[keep])
        plist = RefPairList(feats)
        plist.add_field("tracklet_pairs", pairs[keep])
        plist.add_field("track_cls_logits", t(seg["cls"]))
        plist.add_field("num_tracklets", RefVRDataset._get_num_tracklet_proposals(None, trackid))
        plist.add_field("ious", seg["iou"])
        plist.add_field("track_ids", trackid)
        batches.append(([plist], None, [index]))
    captured = []

    class Recording(RefBaseModel):           # the reference's model; only records what it returns
        def forward(self, pair_list, target_list=None):
            res = super().forward(pair_list, target_list)
            captured.append(res[2][0].clone())
            return res

    res = run_ref_predict(cfg, sd, batches, model_factory=Recording)
    assert len(captured) == 3 and captured[2].shape[0] == 0
    assert set(res.keys()) == set(segs["indexs"][:2]), res.keys()
    for i, index in enumerate(segs["indexs"][:2]):
        scores, trip, tids = _predictions_arrays(res[index][0])
        gaps = scores[:-1] - scores[1:]
        assert gaps.min() > 0, "ties in the g10 decode"
        out[f"seg{i}_scores"], out[f"seg{i}_triplets"], out[f"seg{i}_pair_tids"] = scores, trip, tids
        out[f"seg{i}_min_gap"] = np.array(float(gaps.min()))
        out[f"seg{i}_rel_logits"] = captured[i].numpy()
        np.testing.assert_array_equal(res[index][2], segs["segments"][i]["trackid"])
    save("g10_dataset_predict.npz", **out)


def g7_misc():
    """Known answers: segment_video (lib/modeling/__init__.py:35-41), sampler mask counts
    (relpn/sampler.py:3-66), PairList.to / __getitem__ behaviour."""
    segs = {f"segs_{a}_{b}": np.array(segment_video(a, b), dtype=np.int64).reshape(-1, 2)
            for a, b in ((0, 30), (0, 45), (0, 150), (0, 29))}
    sampler = RefSampler(256, 0.25)
    labels = torch.cat([torch.ones(100), torch.zeros(900)]).long()
    pos, neg = sampler([labels])
    # the drawn masks themselves under a fixed seed (ragged case: fewer positives than the quota)
    torch.manual_seed(1234)
    lab2 = [torch.cat([torch.ones(100), torch.zeros(900), -torch.ones(24)]).long(),
            torch.cat([torch.zeros(300), 2 * torch.ones(7)]).long(), torch.zeros(5).long()]
    pos2, neg2 = sampler(lab2)
    masks = {f"sampler_{k}_{i}": m.numpy() for k, ms in (("pos", pos2), ("neg", neg2)) for i, m in enumerate(ms)}
    save("g7_misc.npz", sampler_counts=np.array([int(pos[0].sum()), int(neg[0].sum())]), **segs, **masks)


def g11_relness():
    """The relationness head as the reference states it: relpn/dpn_anchor.py:82-108 `DPNHead(in_channels, num_anchors)`
    = conv + ReLU, then `relness_pred` Conv1d(C, A, 1) AND `duration_pred` Conv1d(C, 2A, 1), looping over a list of
    feature maps.  Run at the two G3 shapes in fp32 and cast with .bfloat16() (torch CPU bf16 kernels); both outputs
    fp32 / bf16 relness stored.  Its `duration` must equal g3 / g8's (same weights, same arithmetic) - asserted here."""
    out = {}
    pre = "relpn.duration_proposal_network.dpn_head."
    g3 = np.load(os.path.join(HERE, "g3_dpn_head.npz"))
    g8 = np.load(os.path.join(HERE, "g8_bf16.npz"))
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        head = RefAnchorDPNHead(c["c"], 4)
        head.load_state_dict({k[len(pre):]: t(v) for k, v in c["state_dict"].items() if k.startswith(pre)}, strict=True)
        with torch.no_grad():
            rel, dur = head([t(c["x"])])
        assert len(rel) == 1 and len(dur) == 1
        assert np.array_equal(dur[0].numpy(), g3[f"{tag}_duration"])
        out[f"{tag}_relness"] = rel[0].numpy()       # duration: identical to g3 (asserted above), not stored twice
        head = head.bfloat16()
        with torch.no_grad():
            rel, dur = head([t(c["x"]).bfloat16()])
        assert rel[0].dtype == torch.bfloat16
        assert np.array_equal(dur[0].float().numpy(), g8[f"{tag}_duration"])
        out[f"{tag}_relness_bf16"] = rel[0].float().numpy()
    save("g11_relness_head.npz", **out)


def g8_bf16():
    """The reference's own DPNHead and RelationPredictor cast with .bfloat16() and run by torch's CPU
    bf16 kernels on the (bf16-rounded) g3 / g1 inputs: pins the rounding points of the build's bf16
    semantics (oracle.dpn_head_bf16 / predicate_head_bf16).  Outputs are stored as fp32."""
    out = {}
    pre = "relpn.duration_proposal_network.dpn_head."
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        head = RefDPNHead(c["c"], 4)
        head.load_state_dict({k[len(pre):]: t(v) for k, v in c["state_dict"].items()
                              if k.startswith(pre) and "relness" not in k})
        head = head.bfloat16()
        with torch.no_grad():
            y = head(t(c["x"]).bfloat16())
        assert y.dtype == torch.bfloat16
        out[f"{tag}_duration"] = y.float().numpy()
    c = cases.g1_inputs()
    feats = RefVRDataset._feature_preprocess(None, t(c["raw"].copy()))
    feats = (feats if isinstance(feats, torch.Tensor) else t(feats)).float()
    pred = RefRelationPredictor(feats.shape[1], 132)
    pred.load_state_dict({"rel_predictor.weight": t(c["state_dict"]["classifier.rel_predictor.weight"]),
                          "rel_predictor.bias": t(c["state_dict"]["classifier.rel_predictor.bias"])})
    pred = pred.bfloat16()
    with torch.no_grad():
        lg = pred(feats.bfloat16())
    out["cfg1_rel_logits"] = lg.float().numpy()
    save("g8_bf16.npz", **out)


def g9_association():
    """The reference's greedy_relational_association (lib/modeling/association.py:117-175, with its
    VideoRelation / _merge_trajs / _traj_iou and trajectory.py's cubic IoU) on the synthetic
    multi-segment scenario of cases.g9_scenario.  Only the file access of
    `object_trajectory_proposal` is replaced (it reads per-segment JSON written by the reference's
    preprocessing) and dlib.drectangle is the stand-in above."""
    import copy
    rels, trajs = cases.g9_scenario()

    def proposals(dataset, vid, fstart, fend, gt=False, verbose=False):
        return [RefTrajectory(fstart, fend, b.tolist(), 0.0, 0, [], None) for b in trajs[(vid, fstart, fend)]]

    class Names:
        def get_object_name(self, i):
            return int(i)

        def get_predicate_name(self, i):
            return int(i)

    ref_association.object_trajectory_proposal = proposals
    out = {}
    for cap in (100, 25):
        res = ref_association.greedy_relational_association(Names(), copy.deepcopy(rels), max_traj_num_in_clip=cap)
        out[f"cap{cap}_triplet"] = np.array([r["triplet"] for r in res], dtype=np.int64)
        out[f"cap{cap}_score"] = np.array([r["score"] for r in res], dtype=np.float64)
        out[f"cap{cap}_duration"] = np.array([r["duration"] for r in res], dtype=np.int64)
        for side in ("sub_traj", "obj_traj"):
            out[f"cap{cap}_{side}_len"] = np.array([len(r[side]) for r in res], dtype=np.int64)
            out[f"cap{cap}_{side}"] = np.array([b for r in res for b in r[side]], dtype=np.float64).reshape(-1, 4)
        lens = out[f"cap{cap}_sub_traj_len"]
        print(f"  cap {cap}: {len(res)} relations, {int((lens > 30).sum())} extended, longest {int(lens.max())} frames")
    save("g9_association.npz", **out)


def main():
    g1_baseline()
    g2_ppn()
    g3_dpn()
    g4_iou()
    g5_anchors()
    g6_decode()
    g7_misc()
    g8_bf16()
    g11_relness()
    g9_association()
    g10_dataset_and_predict()


if __name__ == "__main__":
    main()
