"""The CPU oracle against the golden vectors produced by the REFERENCE's own modules
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import cases
import oracle


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def sd_t(sd):
    return {k: t(v) for k, v in sd.items()}


PPN_PRE = "relpn.pair_proposal_network.ppn_head."
DPN_PRE = "relpn.duration_proposal_network.dpn_head."


def test_g1_preprocess_and_predicate_head():
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    feats = oracle.feature_preprocess(t(c["raw"]))
    # numpy's pairwise L1 sum vs torch's: a few ulp
    np.testing.assert_allclose(feats[:4, :1200].numpy(), g["preprocessed_rows"], rtol=2e-6, atol=1e-9)
    assert float(feats[3, 70:1070].abs().max()) == 0.0  # zero-norm block stays zero
    chk = np.array([feats.double().sum().item(), feats.double().abs().max().item()])
    np.testing.assert_allclose(chk, g["preprocessed_checksum"], rtol=1e-6)
    sd = sd_t(c["state_dict"])
    logits = oracle.predicate_head(feats, sd["classifier.rel_predictor.weight"],
                                   sd["classifier.rel_predictor.bias"])
    np.testing.assert_allclose(logits.numpy(), g["rel_logits"], rtol=0, atol=1e-6)
    loss = torch.nn.functional.binary_cross_entropy(logits, t(c["targets"]))
    np.testing.assert_allclose(loss.item(), g["loss_rel"], rtol=1e-5)


def test_g2_ppn_matrix_and_topk():
    g = cases.load("g2_ppn_n32.npz")
    c = cases.g2_inputs(int(g["input_seed"]))
    sd = sd_t(c["state_dict"])
    p = {k[len(PPN_PRE):]: v for k, v in sd.items() if k.startswith(PPN_PRE)}
    mat = oracle.ppn_pair_matrix(t(c["cls"]), p)
    np.testing.assert_allclose(mat.numpy(), g["pair_matrix"], rtol=0, atol=1e-6)
    assert float(g["min_gap"]) > 2e-5
    idx = oracle.ppn_topk(mat, 256)
    assert idx.dtype == torch.int64
    np.testing.assert_array_equal(idx.numpy(), g["topk"])  # indices bit-exact
    assert int(idx.max()) < 32 * 32  # flat N*N incl. diagonal


def test_g3_dpn_head():
    """duration: reference relpn/dpn.py:55-73 (g3); relness: reference relpn/dpn_anchor.py:82-108 (g11) - both legs of
    oracle.dpn_head are pinned by the reference's own modules."""
    g = cases.load("g3_dpn_head.npz")
    g11 = cases.load("g11_relness_head.npz")
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        sd = sd_t(c["state_dict"])
        rel, dur, _ = oracle.dpn_head(t(c["x"]), sd[DPN_PRE + "conv.weight"], sd[DPN_PRE + "conv.bias"],
                                      sd[DPN_PRE + "duration_pred.weight"], sd[DPN_PRE + "duration_pred.bias"],
                                      sd[DPN_PRE + "relness_pred.weight"], sd[DPN_PRE + "relness_pred.bias"])
        np.testing.assert_allclose(dur.numpy(), g[f"{tag}_duration"], rtol=0, atol=1e-6)
        assert rel.shape == (c["x"].shape[0], 4, c["x"].shape[2])
        np.testing.assert_allclose(rel.numpy(), g11[f"{tag}_relness"], rtol=0, atol=1e-6)


def test_g4_cubic_iou_bit_exact():
    g = cases.load("g4_cubic_iou.npz")
    b, b2 = cases.g4_inputs()
    iou = oracle.cubic_iou(b)
    assert iou.dtype == np.float32
    np.testing.assert_array_equal(iou, g["iou"])
    np.testing.assert_array_equal(oracle.cubic_iou(b, b2), g["iou_cross"])
    np.testing.assert_array_equal(np.diag(iou), np.ones(32, np.float32))
    np.testing.assert_array_equal(iou, iou.T)


def test_g5_anchors():
    g = cases.load("g5_anchors.npz")
    for i, (sizes, stride, tw) in enumerate(cases.G5_SPECS):
        a = oracle.grid_anchors(sizes, stride, tw)
        np.testing.assert_array_equal(a.numpy(), g[f"anchors_{i}"])
    a0 = g["anchors_0"]
    assert a0.shape == (36, 2)
    np.testing.assert_array_equal(a0[:5], [[-7.5, 7.5], [-15, 15], [-22.5, 22.5], [-30, 30], [0, 15]])
    np.testing.assert_array_equal(a0[-1], [30, 90])


def test_g6_decode():
    g = cases.load("g6_decode.npz")
    c = cases.g6_inputs()
    sc, trip, tids = oracle.decode_topk(t(c["rel_logit"]), t(c["feat70"]), t(c["pairs"]), c["n"])
    np.testing.assert_array_equal(sc.numpy(), g["scores"])
    np.testing.assert_array_equal(trip.numpy(), g["triplets"])
    np.testing.assert_array_equal(tids.numpy(), g["pair_tids"])


def test_g10_proposal_filter_and_predict_pipeline():
    """g10 = the reference's own VRDataset._get_proposal_idx / _get_num_tracklet_proposals, and its whole
    predict() run (dataset filter + _feature_preprocess + BaseModel + top-k decode) on three segments."""
    g = cases.load("g10_dataset_predict.npz")
    for i, (pairs, trackid) in enumerate(cases.g10_tables()):
        np.testing.assert_array_equal(oracle.proposal_pair_idx(pairs, trackid), g[f"proposal_idx_{i}"])
        assert oracle.num_tracklet_proposals(trackid) == int(g[f"num_tracks_{i}"])
    segs = cases.g10_segments()
    sd = sd_t(segs["state_dict"])
    for i, seg in enumerate(segs["segments"][:2]):
        keep = oracle.proposal_pair_idx(seg["pairs"], seg["trackid"])
        n = oracle.num_tracklet_proposals(seg["trackid"])
        assert n == seg["n"] and len(keep) == n * (n - 1)
        feats = oracle.feature_preprocess(t(seg["raw"])[keep])
        logits = oracle.predicate_head(feats, sd["classifier.rel_predictor.weight"],
                                       sd["classifier.rel_predictor.bias"])
        np.testing.assert_allclose(logits.numpy(), g[f"seg{i}_rel_logits"], rtol=0, atol=1e-6)
        # the decode itself on the reference's own logits: exact
        sc, trip, tids = oracle.decode_topk(t(g[f"seg{i}_rel_logits"]), feats[:, :70], t(seg["pairs"][keep]), n)
        np.testing.assert_array_equal(sc.numpy(), g[f"seg{i}_scores"])
        np.testing.assert_array_equal(trip.numpy(), g[f"seg{i}_triplets"])
        np.testing.assert_array_equal(tids.numpy(), g[f"seg{i}_pair_tids"])
    # third segment: one proposal among ground-truth tracks -> no proposal pair, skipped by predict()
    seg = segs["segments"][2]
    assert len(oracle.proposal_pair_idx(seg["pairs"], seg["trackid"])) == 0
    assert oracle.num_tracklet_proposals(seg["trackid"]) == 1


def test_g7_known_answers():
    g = cases.load("g7_misc.npz")
    for a, b in ((0, 30), (0, 45), (0, 150), (0, 29)):
        got = np.array(oracle.segment_video(a, b), dtype=np.int64).reshape(-1, 2)
        np.testing.assert_array_equal(got, g[f"segs_{a}_{b}"])
    assert len(oracle.segment_video(0, 150)) == 9
    np.testing.assert_array_equal(g["sampler_counts"], [64, 192])
    np.testing.assert_array_equal(oracle.pair_index(4).numpy(), cases.ref_pairs(4))
    assert oracle.pair_index(1).shape == (0, 2)


def test_pair_gather_layout_and_geometry():
    v = tspn_video(3, 5, 7, 6)
    pairs = oracle.pair_index(5)
    pf, geom = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)
    assert pf.shape == (20, 12, 7) and geom.shape == (20, 8, 7)
    p = 6  # pair (1, 3)
    assert tuple(pairs[p].tolist()) == (1, 3)
    np.testing.assert_array_equal(pf[p, :6].numpy(), v["tracklet_feats"][1].T)
    np.testing.assert_array_equal(pf[p, 6:].numpy(), v["tracklet_feats"][3].T)
    # per-frame IoU channel agrees with a T=1 cubic_iou of the same boxes
    b = v["tracklet_boxes"]
    for tt in (0, 3):
        ref = oracle.cubic_iou(b[:, tt:tt + 1], b[:, tt:tt + 1])
        np.testing.assert_allclose(geom[p, 4, tt].item(), ref[1, 3], rtol=1e-6)
    assert torch.all(geom[:, 5:7, 0] == 0)


def tspn_video(seed, n, tt, d):
    import tspn_mi355x as tspn
    return tspn.synth.make_video(seed, n, tt, d)


@pytest.mark.parametrize("n,tt,d", [(6, 9, 8), (4, 30, 16)])
def test_factorised_equals_dense(n, tt, d):
    """conv(cat(f_s, f_o)) == conv_s(f_s) + conv_o(f_o): the algebra behind the fused HIP path."""
    import tspn_mi355x as tspn
    v = tspn_video(11, n, tt, d)
    sd = sd_t(tspn.synth.make_weights(0, c=2 * d, k=17, bias_std=0.05))
    w = {"conv_w": sd[DPN_PRE + "conv.weight"], "conv_b": sd[DPN_PRE + "conv.bias"],
         "dur_w": sd[DPN_PRE + "duration_pred.weight"], "dur_b": sd[DPN_PRE + "duration_pred.bias"],
         "rel_w": sd[DPN_PRE + "relness_pred.weight"], "rel_b": sd[DPN_PRE + "relness_pred.bias"],
         "cls_w": sd["classifier.rel_predictor.weight"], "cls_b": sd["classifier.rel_predictor.bias"]}
    pairs = oracle.pair_index(n)
    a = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs, w)
    b = oracle.forward_factorised(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs, w)
    for k in ("relness", "duration", "rel_logits"):
        np.testing.assert_allclose(a[k].numpy(), b[k].numpy(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("n,tt,d", [(6, 9, 8), (5, 30, 32)])
def test_bf16_factorised_oracle_equals_dense_bf16_oracle(n, tt, d):
    """forward_bf16_factorised (the whole-video yardstick of the full-size cfg3 GPU test) lands every bf16 rounding where
    forward_bf16 (pinned by g8 / g11) does: identical activations, so identical outputs up to float64 summation order;
    the float64 / chunked form of forward_factorised equals its fp32 form within fp32 rounding."""
    import tspn_mi355x as tspn
    v = tspn_video(12, n, tt, d)
    sd = sd_t(tspn.synth.make_weights(0, c=2 * d, k=17, bias_std=0.05))
    w = {"conv_w": sd[DPN_PRE + "conv.weight"], "conv_b": sd[DPN_PRE + "conv.bias"],
         "dur_w": sd[DPN_PRE + "duration_pred.weight"], "dur_b": sd[DPN_PRE + "duration_pred.bias"],
         "rel_w": sd[DPN_PRE + "relness_pred.weight"], "rel_b": sd[DPN_PRE + "relness_pred.bias"],
         "cls_w": sd["classifier.rel_predictor.weight"], "cls_b": sd["classifier.rel_predictor.bias"]}
    pairs = oracle.pair_index(n)
    a = oracle.forward_bf16(t(v["tracklet_feats"]), pairs, w)
    b = oracle.forward_bf16_factorised(t(v["tracklet_feats"]), pairs, w, pair_chunk=7)
    for k in ("relness", "duration", "rel_logits"):
        np.testing.assert_allclose(a[k].numpy(), b[k].numpy(), rtol=0, atol=1e-6)
    f32 = oracle.forward_factorised(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs, w)
    f64 = oracle.forward_factorised(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs, w, dtype=torch.float64, pair_chunk=5)
    for k in ("relness", "duration", "rel_logits"):
        assert f64[k].dtype == torch.float64
        np.testing.assert_allclose(f32[k].numpy(), f64[k].numpy(), rtol=0, atol=2e-6)


def test_rel_oi_pool_spans():
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    np.testing.assert_allclose(oracle.rel_oi_pool(x).numpy(), x.mean(2).numpy())
    spans = torch.tensor([[1, 3], [0, 5]])
    out = oracle.rel_oi_pool(x, spans)
    np.testing.assert_allclose(out[0].numpy(), x[0, :, 1:3].mean(1).numpy())
    np.testing.assert_allclose(out[1].numpy(), x[1].mean(1).numpy())
    assert oracle.rel_oi_pool(torch.ones(4, 7)).shape == (4, 7)


def test_g8_bf16_semantics_pinned_by_reference_bf16_modules():
    """The bf16 restatement (operands bf16, exact products, wide sums, encoder activation rounded
    once) against the reference's DPNHead / RelationPredictor cast with .bfloat16() (torch CPU
    bf16 kernels): rounding the oracle's unrounded output to bf16 reproduces the reference's output
    on (nearly) every element, and never differs by more than two bf16 ulps."""
    g = cases.load("g8_bf16.npz")
    g11 = cases.load("g11_relness_head.npz")
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        sd = sd_t(c["state_dict"])
        rel, dur, h = oracle.dpn_head_bf16(t(c["x"]), sd[DPN_PRE + "conv.weight"], sd[DPN_PRE + "conv.bias"],
                                           sd[DPN_PRE + "duration_pred.weight"], sd[DPN_PRE + "duration_pred.bias"],
                                           sd[DPN_PRE + "relness_pred.weight"], sd[DPN_PRE + "relness_pred.bias"])
        # duration: dpn.py's DPNHead.bfloat16() (g8); relness: dpn_anchor.py's DPNHead.bfloat16() (g11)
        for got, ref in ((dur, g[f"{tag}_duration"]), (rel, g11[f"{tag}_relness_bf16"])):
            same = (oracle.bf16_round(got).numpy() == ref).mean()
            assert same > 0.999, same
            ulp = np.maximum(np.abs(ref), 2.0 ** -126) * 2.0 ** -7   # spacing of bf16 at |ref| (upper bound)
            assert np.all(np.abs(got.numpy() - ref) <= 2 * ulp)
        np.testing.assert_array_equal(oracle.bf16_round(h).numpy(), h.numpy())   # activation is bf16
    c = cases.g1_inputs()
    feats = oracle.feature_preprocess(t(c["raw"].copy()))
    lg = oracle.predicate_head_bf16(feats, t(c["state_dict"]["classifier.rel_predictor.weight"]),
                                    t(c["state_dict"]["classifier.rel_predictor.bias"]))
    # the reference rounds the pre-sigmoid logit AND the sigmoid to bf16: within 1.5 ulp at 0.5
    np.testing.assert_allclose(lg.numpy(), g["cfg1_rel_logits"], rtol=0, atol=1.5 * 2.0 ** -9)


def test_forward_bf16_is_forward_dense_on_rounded_operands_up_to_one_rounding():
    """forward_bf16 differs from the fp32 dense forward on bf16-rounded operands only by the single
    rounding of the encoder activation and of the pooled feature."""
    import tspn_mi355x as tspn
    n, tt, d = 5, 12, 16
    v = tspn.synth.make_video(31, n, tt, d)
    sd = tspn.synth.make_weights(0, c=2 * d, bias_std=0.05)
    w = {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
         "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
         "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
         "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}
    pairs = oracle.pair_index(n)
    out = oracle.forward_bf16(t(v["tracklet_feats"]), pairs, w)
    wr = {k: oracle.bf16_round(x) for k, x in w.items()}
    ref = oracle.forward_dense(oracle.bf16_round(t(v["tracklet_feats"])), t(v["tracklet_boxes"]), pairs, wr)
    for k in ("relness", "duration", "rel_logits"):
        assert out[k].shape == ref[k].shape
        scale = float(ref[k].abs().max())
        assert float((out[k] - ref[k]).abs().max()) <= 2.0 ** -8 * max(scale, 1.0)


def test_roi_align_oracle_known_answers():
    """oracle.roi_head_oracle.roi_align_nhwc (restated detectron2 CPU kernel): hand-computed cases."""
    from oracle import roi_head_oracle as ro
    # map value = 10*y + x; legacy (aligned=False) box [0,0,2,2] at scale 1, P=2, one sample per bin:
    # sample points (0.5,0.5), (0.5,1.5), (1.5,0.5), (1.5,1.5) -> bilinear of a linear map = its value there
    fmap = np.fromfunction(lambda y, x: 10.0 * y + x, (4, 4), dtype=np.float32)[None, :, :, None]
    out = ro.roi_align_nhwc(fmap, np.array([[0, 0, 0, 2, 2]], dtype=np.float32), 2, 1.0, 1, False)
    np.testing.assert_allclose(out[0, :, :, 0].numpy(), [[5.5, 6.5], [15.5, 16.5]], rtol=1e-6)
    # aligned=True shifts the sampling grid by half a pixel
    out = ro.roi_align_nhwc(fmap, np.array([[0, 0, 0, 2, 2]], dtype=np.float32), 2, 1.0, 1, True)
    np.testing.assert_allclose(out[0, :, :, 0].numpy(), [[0.0, 1.0], [10.0, 11.0]], atol=1e-6)
    # a box entirely outside the map pools to zero; a constant map pools to the constant
    assert float(ro.roi_align_nhwc(fmap, np.array([[0, 50, 50, 60, 60]], dtype=np.float32), 2, 1.0, 0, True).abs().max()) == 0.0
    const = ro.roi_align_nhwc(np.full((1, 5, 5, 3), 2.5, dtype=np.float32), np.array([[0, 1, 1, 30, 40]], dtype=np.float32),
                              3, 0.1, 0, True)
    np.testing.assert_allclose(const.numpy(), 2.5, rtol=1e-6)
