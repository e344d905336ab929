"""a15 on the GPU: the proposal pair filter (VRDataset._get_proposal_idx, vrdataset.py:140-148), the row
gathers of vrdataset.py:66-67 and the whole dataset -> model -> decode path of predict.py against golden
g10 (the reference's own run) and the oracle.  Indices bit-exact; logits within 1e-5."""
import numpy as np
import pytest
import torch

import cases
import oracle

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_proposal_pair_filter_golden(tspn, device):
    g = cases.load("g10_dataset_predict.npz")
    for i, (pairs, trackid) in enumerate(cases.g10_tables()):
        idx, n = tspn.ops.proposal_pair_filter(t(pairs).to(device), t(trackid).to(device))
        assert idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f"proposal_idx_{i}"])
        assert n == int(g[f"num_tracks_{i}"])


def test_proposal_pair_filter_batched_ragged_and_large(tspn, device):
    """Several segments in one launch (ragged, one empty, one larger than a workgroup pass) against the
    oracle; out-of-range track indices are reported per segment."""
    rs = np.random.RandomState(3)
    tables = cases.g10_tables()
    m_big = 70                                                # 4830 pairs: 19 passes of 256
    tid_big = np.where(rs.rand(m_big) < 0.8, -1, rs.randint(0, 9, m_big)).astype(np.int64)
    tables.append((cases.full_pairs(m_big), tid_big))
    tables.insert(2, (np.zeros((0, 2), np.int64), np.zeros((0,), np.int64)))   # empty segment
    pair_off = np.cumsum([0] + [len(p) for p, _ in tables]).astype(np.int64)
    track_off = np.cumsum([0] + [len(x) for _, x in tables]).astype(np.int64)
    pairs = np.concatenate([p for p, _ in tables])
    tids = np.concatenate([x for _, x in tables])
    idx, cnt, ntr = tspn.ops.proposal_pair_filter(t(pairs).to(device), t(tids).to(device),
                                                  t(pair_off).to(device), t(track_off).to(device))
    idx, cnt, ntr = idx.cpu().numpy(), cnt.cpu().numpy(), ntr.cpu().numpy()
    for s, (p, x) in enumerate(tables):
        ref = oracle.proposal_pair_idx(p, x)
        assert cnt[s] == len(ref) and ntr[s] == oracle.num_tracklet_proposals(x)
        np.testing.assert_array_equal(idx[pair_off[s]:pair_off[s] + cnt[s]], ref)
    bad = pairs.copy()
    bad[pair_off[1] + 3, 1] = 99                              # segment 1 has 9 tracks
    _, cnt2, _ = tspn.ops.proposal_pair_filter(t(bad).to(device), t(tids).to(device),
                                               t(pair_off).to(device), t(track_off).to(device))
    cnt2 = cnt2.cpu().numpy()
    assert cnt2[1] == -1 and np.array_equal(np.delete(cnt2, 1), np.delete(cnt, 1))
    with pytest.raises(IndexError):
        tspn.ops.proposal_pair_filter(t(np.array([[0, 5]], np.int64)).to(device), t(-np.ones(3, np.int64)).to(device))


@pytest.mark.parametrize("R,F", [(7, 11070), (300, 5), (1, 1), (33, 64)])
def test_gather_rows_bit_exact(tspn, device, R, F):
    src = tspn.hashrng.uniform(5, "rows", (R, F), -4, 4)
    idx = np.argsort(tspn.hashrng.bits(5, "perm", 2 * R), kind="stable") % R
    out = tspn.ops.gather_rows(t(src).to(device), t(idx.astype(np.int64)).to(device))
    np.testing.assert_array_equal(out.cpu().numpy(), src[idx])
    empty = tspn.ops.gather_rows(t(src).to(device), torch.zeros(0, dtype=torch.int64, device=device))
    assert empty.shape == (0, F)
    with pytest.raises(IndexError):
        tspn.ops.gather_rows(t(src).to(device), torch.tensor([R], device=device))


def _g10_loader(tspn, device, preprocess=True):
    segs = cases.g10_segments()
    batches = []
    for index, seg in zip(segs["indexs"], segs["segments"]):
        plist, _ = tspn.dataset.proposal_pair_list(seg["pairs"], seg["raw"], seg["iou"], seg["trackid"], seg["cls"],
                                                   preprocess=preprocess, device=device)
        batches.append(([plist], None, [index]))
    return segs, batches


@pytest.mark.parametrize("fuse", [False, True])
def test_dataset_to_predictions_like_the_reference(tspn, device, fuse):
    """h5-shaped arrays -> proposal_pair_list -> BaseModel -> decode == the reference's own predict() run
    (golden g10), with the block-L1 preprocessing either applied by the dataset step or fused into the GEMM."""
    g = cases.load("g10_dataset_predict.npz")
    segs, batches = _g10_loader(tspn, device, preprocess=not fuse)
    model = tspn.BaseModel(cases.baseline_cfg(**{"PREDICT.FUSE_PREPROCESS": fuse}))
    own = model.state_dict()
    model.load_state_dict({k: t(v) for k, v in segs["state_dict"].items() if k in own})
    model.eval()
    # dataset step
    for i, (pl, _, _) in enumerate(batches[:2]):
        seg = segs["segments"][i]
        keep = oracle.proposal_pair_idx(seg["pairs"], seg["trackid"])
        assert int(pl[0].get_field("num_tracklets")) == seg["n"]
        np.testing.assert_array_equal(pl[0].get_field("tracklet_pairs").cpu().numpy(), seg["pairs"][keep])
        if not fuse:
            ref = oracle.feature_preprocess(t(seg["raw"])[keep])
            np.testing.assert_allclose(pl[0].features.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=1e-9)
        else:
            np.testing.assert_array_equal(pl[0].features.cpu().numpy(), seg["raw"][keep])
    assert batches[2][0][0].features.shape == (0, 11070) and int(batches[2][0][0].get_field("num_tracklets")) == 1
    # model + decode through the prediction loop
    res = tspn.predict.predict_short_term_relations(model, batches)
    assert set(res.keys()) == set(segs["indexs"][:2])           # the 1-tracklet segment is skipped
    for i, index in enumerate(segs["indexs"][:2]):
        preds, iou, tid = res[index]
        sc = np.array([p[0] for p in preds], dtype=np.float32)
        trip = np.array([p[1] for p in preds]); pt = np.array([p[2] for p in preds])
        np.testing.assert_allclose(sc, g[f"seg{i}_scores"], rtol=0, atol=1e-5)
        # entries whose score is separated from both neighbours by more than the logit tolerance must
        # be the same triplet at the same rank; closer ones may swap places
        gs = g[f"seg{i}_scores"].astype(np.float64)
        gap = np.minimum(np.append(gs[:-1] - gs[1:], 1.0), np.append(1.0, gs[:-1] - gs[1:]))
        firm = gap > 2e-5
        assert firm.sum() > 100
        np.testing.assert_array_equal(trip[firm], g[f"seg{i}_triplets"][firm])
        np.testing.assert_array_equal(pt[firm], g[f"seg{i}_pair_tids"][firm])
        np.testing.assert_array_equal(np.asarray(tid), segs["segments"][i]["trackid"])
    # the decode kernel on the reference's own logits: bit-exact
    for i, (pl, _, _) in enumerate(batches[:2]):
        if fuse:
            continue
        out = model.decode(pl, [t(g[f"seg{i}_rel_logits"]).to(device)])[0]
        np.testing.assert_array_equal(out[0].cpu().numpy(), g[f"seg{i}_scores"])
        np.testing.assert_array_equal(out[1].cpu().numpy(), g[f"seg{i}_triplets"])
        np.testing.assert_array_equal(out[2].cpu().numpy(), g[f"seg{i}_pair_tids"])


def test_batched_proposal_pair_lists_equal_per_segment_form(tspn, device):
    """dataset.proposal_pair_lists (one filter launch + one host sync per batch) == proposal_pair_list per
    segment, bit for bit, on the h5-shaped segments of golden g10 (incl. the 1-tracklet segment); a pair that
    names a track outside its segment raises like the per-segment form."""
    segs = cases.g10_segments()
    batch = [{"pairs": s["pairs"], "feats": s["raw"], "iou": s["iou"], "trackid": s["trackid"], "cls_logits": s["cls"]}
             for s in segs["segments"]]
    got = tspn.dataset.proposal_pair_lists(batch, preprocess=True, device=device)
    _, one = _g10_loader(tspn, device, preprocess=True)
    assert len(got) == len(one)
    for (pl, tl), (ref, _, _) in zip(got, one):
        assert tl is None
        assert torch.equal(pl.features, ref[0].features)
        assert torch.equal(pl.get_field("tracklet_pairs"), ref[0].get_field("tracklet_pairs"))
        assert int(pl.get_field("num_tracklets")) == int(ref[0].get_field("num_tracklets"))
    bad = dict(batch[0])
    bad["pairs"] = batch[0]["pairs"].copy()
    bad["pairs"][3, 1] = 10 ** 6
    with pytest.raises(IndexError):
        tspn.dataset.proposal_pair_lists([batch[1], bad], device=device)
    assert tspn.dataset.proposal_pair_lists([], device=device) == []
