"""`BaseModel.forward()` drop-in behaviour on the GPU against the reference-generated goldens
and the oracle (the tests read like calls the reference's train.py / predict.py make)."""
import numpy as np
import pytest
import torch

import cases
import oracle

pytestmark = pytest.mark.gpu

DPN_PRE = "relpn.duration_proposal_network.dpn_head."


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load(model, sd):
    own = model.state_dict()
    model.load_state_dict({k: t(v) for k, v in sd.items() if k in own})


def test_baseline_yaml_forward_cpu_inputs_like_predict_py(tspn, device):
    """predict.py:21-57: model never .cuda()'d, CPU PairList in, CPU results out."""
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    model = tspn.BaseModel(cases.baseline_cfg())
    load(model, c["state_dict"])
    model.eval()
    feats = oracle.feature_preprocess(t(c["raw"]))
    plist = tspn.PairList(feats)
    plist.add_field("track_cls_logits", t(c["cls"]))
    with torch.no_grad():
        pp, dp, logits = model([plist], None)
    assert pp is None and dp is None and len(logits) == 1
    assert logits[0].device.type == "cpu" and logits[0].shape == (56, 132)
    np.testing.assert_allclose(logits[0].numpy(), g["rel_logits"], rtol=0, atol=2e-6)


def test_baseline_train_loss_and_grads_like_train_py(tspn, device):
    """train.py:44-78: model.cuda(), inputs .to(gpu), loss dict, backward."""
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    model = tspn.BaseModel(cases.baseline_cfg()).to(device)
    load(model, c["state_dict"])
    model.train()
    feats = oracle.feature_preprocess(t(c["raw"]))
    plist = tspn.PairList(feats).to(device)
    tlist = tspn.TargetList(t(c["targets"])).to(device)
    loss = model([plist], [tlist])
    assert set(loss) == {"loss_rel"}
    np.testing.assert_allclose(loss["loss_rel"].item(), g["loss_rel"], rtol=2e-5)
    loss["loss_rel"].backward()
    w = model.classifier.rel_predictor.weight
    # gradient against torch autograd of the same math
    wr = t(c["state_dict"]["classifier.rel_predictor.weight"]).clone().requires_grad_(True)
    br = t(c["state_dict"]["classifier.rel_predictor.bias"]).clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(feats @ wr.t() + br), t(c["targets"]))
    ref.backward()
    np.testing.assert_allclose(w.grad.cpu().numpy(), wr.grad.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(model.classifier.rel_predictor.bias.grad.cpu().numpy(), br.grad.numpy(),
                               rtol=0, atol=1e-7)


def test_use_ppn_eval_and_train(tspn, device):
    g = cases.load("g2_ppn_n32.npz")
    c = cases.g2_inputs(int(g["input_seed"]))
    cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": True, "PREDICT.FEATURE_DIM": 64})
    model = tspn.BaseModel(cfg)
    load(model, c["state_dict"])
    model.eval()

    def sample():
        plist = tspn.PairList(t(c["feats"]))
        plist.add_field("track_cls_logits", t(c["cls"]))
        plist.add_field("tracklet_pairs", c["pairs"])
        plist.add_field("num_tracklets", np.int64(c["n"]))
        return plist

    with torch.no_grad():
        pp, dp, logits = model([sample(), sample()], None)
    assert dp is None and len(pp) == 2
    for idx in pp:
        assert idx.dtype == torch.int64 and idx.shape == (256,)
        np.testing.assert_array_equal(idx.numpy(), g["topk"])
    np.testing.assert_allclose(logits[0][:8].numpy(), g["rel_logits_head"], rtol=0, atol=2e-6)
    model.to(device).train()
    loss = model([sample().to(device)], [tspn.TargetList(t(c["targets"])).to(device)])
    assert set(loss) == {"loss_pair", "loss_rel"}
    np.testing.assert_allclose(loss["loss_pair"].item(), g["loss_pair"], rtol=2e-5)
    np.testing.assert_allclose(loss["loss_rel"].item(), g["loss_rel"], rtol=2e-5)
    sum(loss.values()).backward()
    head = model.relpn.pair_proposal_network.ppn_head
    assert head.sub_emb[0].weight.grad is not None
    # round 4: every GEMM of the PPN training step (two MLPs + the pair matrix), forward and backward, runs on the
    # library's HIP GEMM -- gradients against float64 autograd of the same formula (ppn.py:92-112, 57-71)
    sd = {k: t(v).double().requires_grad_(True) for k, v in c["state_dict"].items() if k.startswith("relpn.pair_proposal_network.")}
    pre = "relpn.pair_proposal_network.ppn_head."
    x = t(c["cls"]).double()

    def mlp(name):
        h = torch.relu(x @ sd[pre + name + ".0.weight"].t() + sd[pre + name + ".0.bias"])
        return h @ sd[pre + name + ".2.weight"].t() + sd[pre + name + ".2.bias"]
    pm = torch.sigmoid(mlp("sub_emb") @ mlp("obj_emb").t())
    gt = tspn.PPN._gt_matrices([sample()], [tspn.TargetList(t(c["targets"]))])[0].double()
    ref_loss = torch.nn.functional.binary_cross_entropy(pm, gt)
    ref_loss.backward()
    assert abs(float(loss["loss_pair"]) - float(ref_loss)) <= 1e-6 * max(1.0, float(ref_loss))   # fp32 loss vs float64
    for name in ("sub_emb.0.weight", "sub_emb.0.bias", "sub_emb.2.weight", "sub_emb.2.bias",
                 "obj_emb.0.weight", "obj_emb.0.bias", "obj_emb.2.weight", "obj_emb.2.bias"):
        got = dict(head.named_parameters())[name].grad.cpu().double()
        want = sd[pre + name].grad
        assert float((got - want).abs().max()) <= 2e-7 + 1e-5 * float(want.abs().max()), name


def temporal_cfg(D, use_ppn=True):
    return cases.baseline_cfg(**{"RELPN.USE_PPN": use_ppn, "RELPN.USE_DPN": True,
                                 "RELPN.DPN.IN_CHANNELS": 2 * D, "PREDICT.FEATURE_DIM": 2 * D})


def oracle_weights(sd):
    return {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
            "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
            "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
            "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}


def test_temporal_forward_tracklets_fused(tspn, device):
    """USE_DPN=True on tracklet tensors: ragged batch (different N, T per segment)."""
    D = 24
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D))
    load(model, sd)
    model.eval()
    shapes = [(6, 30), (4, 17), (6, 30)]
    vids = [tspn.synth.make_video(70 + i, n, tt, D) for i, (n, tt) in enumerate(shapes)]
    plists = [tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]),
                                           t(v["track_cls_logits"])) for v in vids]
    pp, dp, logits = model(plists, None)
    w = oracle_weights(sd)
    for i, v in enumerate(vids):
        n = shapes[i][0]
        ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(n), w)
        assert dp[i].relness.shape == ref["relness"].shape and dp[i].duration.shape == ref["duration"].shape
        np.testing.assert_allclose(dp[i].relness.numpy(), ref["relness"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(dp[i].duration.numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(logits[i].numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
        assert pp[i].shape == (min(256, n * n),)
    geo = model.pair_geometry(plists)
    ref_g = oracle.pair_geometry(t(vids[1]["tracklet_boxes"]), oracle.pair_index(4))
    np.testing.assert_allclose(geo[1].numpy(), ref_g.numpy(), rtol=2e-6, atol=2e-6)
    # the bbox half of the pair builder is an opt-in output of forward (RELPN.DPN.PAIR_GEOMETRY; nothing downstream
    # consumes it): off by default, one launch per group of segments when on
    assert all(d.geom is None for d in dp)
    gmodel = tspn.BaseModel(cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                                  "PREDICT.FEATURE_DIM": 2 * D, "RELPN.DPN.PAIR_GEOMETRY": True}))
    load(gmodel, sd)
    gmodel.eval()
    _, dp, logits_g = gmodel(plists, None)
    assert all(torch.equal(a, b) for a, b in zip(logits, logits_g))
    for i, v in enumerate(vids):
        n = shapes[i][0]
        assert dp[i].geom.shape == (n * (n - 1), 8, shapes[i][1]) and dp[i].geom.device.type == "cpu"
        np.testing.assert_allclose(dp[i].geom.numpy(), oracle.pair_geometry(t(v["tracklet_boxes"]), oracle.pair_index(n)).numpy(),
                                   rtol=2e-6, atol=2e-6)
        assert torch.equal(dp[i].geom, geo[i])
    no_boxes = tspn.PairList.from_tracklets(t(vids[0]["tracklet_feats"]), None, t(vids[0]["track_cls_logits"]))
    assert gmodel([no_boxes], None)[1][0].geom is None


def test_temporal_forward_materialised_features_dense(tspn, device):
    """USE_DPN=True with PairList.features = [P,C,T] (the layout DPNHead is documented to take)."""
    D, N, T = 16, 5, 30
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False))
    load(model, sd)
    model.eval()
    v = tspn.synth.make_video(80, N, T, D)
    pairs = oracle.pair_index(N)
    pf, _ = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)
    pp, dp, logits = model([tspn.PairList(pf)], None)
    ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs, oracle_weights(sd))
    assert pp is None
    np.testing.assert_allclose(dp[0].duration.numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(dp[0].relness.numpy(), ref["relness"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(logits[0].numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)


def test_temporal_custom_pairs_and_errors(tspn, device):
    D, N, T = 8, 5, 12
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False))
    load(model, sd)
    model.eval()
    v = tspn.synth.make_video(81, N, T, D)
    sub = oracle.pair_index(N)[[3, 0, 17]]
    plist = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), tracklet_pairs=sub.numpy())
    _, dp, logits = model([plist], None)
    ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), sub, oracle_weights(sd))
    np.testing.assert_allclose(dp[0].duration.numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(logits[0].numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
    with pytest.raises(ValueError):  # 2-D features with USE_DPN=True (reference: NameError)
        model([tspn.PairList(torch.zeros(4, 2 * D))], None)
    bad = tspn.PairList.from_tracklets(torch.zeros(3, 4, D + 1))
    with pytest.raises(ValueError):
        model([bad], None)


def test_reference_checkpoint_without_relness_loads(tspn, device):
    model = tspn.BaseModel(cases.baseline_cfg())
    sd = {("module." + k): v for k, v in model.state_dict().items() if "relness_pred" not in k}
    stripped = {k[len("module."):]: v for k, v in sd.items()}  # lib/utils/serialize.py:13-16
    model.load_state_dict(stripped)  # strict


def test_decode_like_predict_py(tspn, device):
    """predict.py:57-117: forward, then top-20 / top-200 decode — baseline (feature-row quirk) and
    tracklet segments, batch of two."""
    g = cases.load("g6_decode.npz")
    c = cases.g6_inputs()
    model = tspn.BaseModel(cases.baseline_cfg())
    feats = torch.zeros(c["rel_logit"].shape[0], 11070)
    feats[:, :70] = t(c["feat70"])
    plist = tspn.PairList(feats)
    plist.add_field("tracklet_pairs", c["pairs"])
    plist.add_field("num_tracklets", np.int64(c["n"]))
    single = tspn.PairList(torch.zeros(0, 11070))
    single.add_field("num_tracklets", np.int64(1))
    res = model.decode([plist, single, plist], [t(c["rel_logit"]), torch.zeros(0, 132), t(c["rel_logit"])])
    for k in (0, 2):
        sc, trip, tids = res[k]
        assert sc.device.type == "cpu"
        np.testing.assert_array_equal(sc.numpy(), g["scores"])
        np.testing.assert_array_equal(trip.numpy(), g["triplets"])
        np.testing.assert_array_equal(tids.numpy(), g["pair_tids"])
    assert res[1][0].shape == (0,) and res[1][1].shape == (0, 3)
    # tracklet segment: class labels come from track_cls_logits
    D = 8
    m2 = tspn.BaseModel(temporal_cfg(D, use_ppn=False)).eval()
    v = tspn.synth.make_video(90, 5, 12, D)
    pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), t(v["track_cls_logits"]))
    _, _, logits = m2([pl], None)
    sc, trip, tids = m2.decode([pl], logits)[0]
    rs, rt, ri = oracle.decode_topk(logits[0], torch.cat([t(v["track_cls_logits"])] * 2, 1)[
        oracle.pair_index(5)[:, 0] * 0 + torch.arange(20) // 4], oracle.pair_index(5), 5)
    np.testing.assert_array_equal(sc.numpy(), rs.numpy())
    np.testing.assert_array_equal(tids.numpy(), ri.numpy())
    lab = v["track_cls_logits"].argmax(-1)
    np.testing.assert_array_equal(trip[:, 0].numpy(), lab[tids[:, 0].numpy()])
    np.testing.assert_array_equal(trip[:, 2].numpy(), lab[tids[:, 1].numpy()])
    np.testing.assert_array_equal(trip[:, 1].numpy(), rt[:, 1].numpy())


def test_decode_spans_from_forward(tspn, device):
    """forward -> decode_spans: integer spans bit-exact vs the oracle run on the same heads."""
    D, N, T = 16, 6, 30
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    sd = {k: (v * 40 if "relness_pred" in k or "duration_pred" in k else v) for k, v in sd.items()}
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False))
    load(model, sd)
    model.eval()
    v = tspn.synth.make_video(91, N, T, D)
    pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]))
    _, dp, _ = model([pl], None)
    spans = model.decode_spans(dp)[0]
    sizes = model.anchor_sizes(T)
    assert sizes == [7.5, 15.0, 22.5, 30.0]
    ref = oracle.decode_spans(dp[0].relness, dp[0].duration, sizes, top_k=64)
    for k in ("count", "anchor", "span"):
        np.testing.assert_array_equal(spans[k].numpy(), ref[k].numpy(), err_msg=k)
    assert spans["span"].shape == (N * (N - 1), 64, 2) and int(spans["count"].min()) >= 1


@pytest.mark.parametrize("n,tt,d", [(1, 30, 16), (2, 1, 16), (3, 7, 10), (2, 30, 16), (9, 31, 32), (9, 30, 24)])
def test_temporal_forward_edge_shapes(tspn, device, n, tt, d):
    """Ragged / degenerate segments through every dispatch of the fused path: a single tracklet (no
    pairs), one frame, odd T (direct conv instead of Winograd), D not a multiple of 16 (transpose +
    general conv kernel), D % 16 == 8."""
    sd = tspn.synth.make_weights(0, c=2 * d, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(d))
    load(model, sd)
    model.eval()
    v = tspn.synth.make_video(95, n, tt, d)
    pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), t(v["track_cls_logits"]))
    pp, dp, logits = model([pl], None)
    p = n * (n - 1)
    assert logits[0].shape == (p, 132) and dp[0].relness.shape == (p, 4, tt) and pp[0].shape == (min(256, n * n),)
    if p:
        ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(n),
                                   oracle_weights(sd))
        np.testing.assert_allclose(dp[0].relness.numpy(), ref["relness"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(dp[0].duration.numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(logits[0].numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
        spans = model.decode_spans(dp)[0]
        assert spans["span"].shape[0] == p
    dec = model.decode([pl], logits)[0]
    assert dec[0].shape[0] == (min(200, p * 20) if n > 1 else 0)


def test_baseline_forward_fused_preprocess_flag(tspn, device):
    """PREDICT.FUSE_PREPROCESS: raw features in, same logits as reference preprocess + forward (G1)."""
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    model = tspn.BaseModel(cases.baseline_cfg(**{"PREDICT.FUSE_PREPROCESS": True}))
    load(model, c["state_dict"])
    model.eval()
    _, _, logits = model([tspn.PairList(t(c["raw"]))], None)
    np.testing.assert_allclose(logits[0].numpy(), g["rel_logits"], rtol=0, atol=2e-6)


def test_span_restricted_reloipool(tspn, device):
    """RelOIPool over a pair's own span + predicate head == oracle.rel_oi_pool(feats, spans) ->
    predicate_head; whole-segment spans reproduce forward()'s logits; POOL_TOP_SPAN wires the top
    decoded span into forward()."""
    D, n, tt = 24, 7, 30
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D))
    load(model, sd)
    model.eval()
    v = tspn.synth.make_video(75, n, tt, D)
    pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), t(v["track_cls_logits"]))
    pp, dp, logits = model([pl], None)
    P = n * (n - 1)
    rng = np.random.RandomState(3)
    a = rng.randint(0, tt - 1, size=P)
    e = np.minimum(a + 1 + rng.randint(0, tt, size=P), tt)
    spans = np.stack([a, e], axis=1).astype(np.int64)
    spans[0] = (0, tt)
    spans[1] = (tt - 1, tt)
    got = model.classify_spans([pl], [spans])[0]
    w = oracle_weights(sd)
    pf, _ = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(n))
    ref = oracle.predicate_head(oracle.rel_oi_pool(pf, t(spans)), w["cls_w"], w["cls_b"])
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-5)
    whole = np.tile(np.array([[0, tt]], np.int64), (P, 1))
    np.testing.assert_allclose(model.classify_spans([pl], [whole])[0].numpy(), logits[0].numpy(), rtol=0, atol=1e-5)
    unused = np.tile(np.array([[-1, -1]], np.int64), (P, 1))     # "no span": whole segment
    np.testing.assert_allclose(model.classify_spans([pl], [unused])[0].numpy(), logits[0].numpy(), rtol=0, atol=1e-5)
    # forward() with POOL_TOP_SPAN: logits pooled over decode_spans' first span
    cfg = temporal_cfg(D)
    cfg.RELPN.DPN.POOL_TOP_SPAN = True
    m2 = tspn.BaseModel(cfg)
    load(m2, sd)
    m2.eval()
    _, dp2, lg2 = m2([pl], None)
    top = m2.decode_spans(dp2, top_k=1)[0]["span"][:, 0]
    ref_heads = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(n), w)
    ref_sp = oracle.decode_spans(ref_heads["relness"], ref_heads["duration"], m2.anchor_sizes(tt), top_k=1)
    np.testing.assert_array_equal(top.numpy(), ref_sp["span"][:, 0])
    ref2 = oracle.predicate_head(oracle.rel_oi_pool(pf, t(ref_sp["span"][:, 0])), w["cls_w"], w["cls_b"])
    np.testing.assert_allclose(lg2[0].numpy(), ref2.numpy(), rtol=0, atol=1e-5)


def test_predict_like_pipeline_forward_decode_associate(tspn, device):
    """The chain the reference's base.py:89-105 runs per video — forward on every 30-frame segment,
    top-k triplet decode (predict.py:59-117), greedy association across segments — on the GPU path +
    host association against the oracle chain on the same synthetic video."""
    D, n = 16, 5
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False))
    load(model, sd)
    model.eval()
    w = oracle_weights(sd)
    total = 75
    vid = tspn.synth.make_video(120, n, total, D)
    vid["track_cls_logits"] = 8.0 * vid["track_cls_logits"]
    segs = oracle.segment_video(0, total)                      # [(0,30), (15,45), (30,60), (45,75)]
    assert len(segs) == 4
    plists, trajs = [], {}
    for fs, fe in segs:
        plists.append(tspn.PairList.from_tracklets(t(vid["tracklet_feats"][:, fs:fe].copy()),
                                                   t(vid["tracklet_boxes"][:, fs:fe].copy()),
                                                   t(vid["track_cls_logits"])))
        trajs[("v0", fs, fe)] = vid["tracklet_boxes"][:, fs:fe].astype(np.float64)
    _, dp, logits = model(plists, None)
    dec = model.decode(plists, logits, topk_per_pair=3, topk_per_seg=12)
    rels_gpu, rels_ref = [], []
    for (fs, fe), lg, (sc, trip, tids) in zip(segs, logits, dec):
        ref = oracle.forward_dense(t(vid["tracklet_feats"][:, fs:fe].copy()), t(vid["tracklet_boxes"][:, fs:fe].copy()),
                                   oracle.pair_index(n), w)
        np.testing.assert_allclose(lg.numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
        cls = t(vid["track_cls_logits"])
        pairs = oracle.pair_index(n)
        # oracle decode on the GPU logits (index parity is defined on identical scores); tracklet samples
        # read the class of tracklet `tid` directly: rows = tracklets, "num_tracklets - 1" = 1
        rsc, rtrip, rtids = oracle.decode_topk(lg, torch.cat([cls, cls], dim=1), pairs, 2, 3, 12)
        np.testing.assert_array_equal(trip.numpy(), rtrip.numpy())
        np.testing.assert_array_equal(tids.numpy(), rtids.numpy())
        np.testing.assert_array_equal(sc.numpy(), rsc.numpy())
        preds = [(np.array(s), np.array(tr), np.array(td)) for s, tr, td in zip(sc.numpy(), trip.numpy(), tids.numpy())]
        rels_gpu.append((("v0", fs, fe), (preds, None, None)))
        rels_ref.append((("v0", fs, fe), (list(preds), None, None)))
    out = tspn.association.greedy_relational_association(None, rels_gpu, trajectories=trajs)
    ref_out = oracle.greedy_association(rels_ref, trajs)
    assert len(out) == len(ref_out) and len(out) >= 12
    assert max(len(r["sub_traj"]) for r in out) > 30           # something was associated across segments
    for a, b in zip(out, ref_out):
        assert a["triplet"] == b["triplet"] and a["duration"] == b["duration"] and a["score"] == b["score"]
        np.testing.assert_array_equal(np.asarray(a["sub_traj"]), b["sub_traj"])
        np.testing.assert_array_equal(np.asarray(a["obj_traj"]), b["obj_traj"])


def test_predict_loop_like_predict_py(tspn, device):
    """tspn.predict.predict_short_term_relations == the loop of reference predict.py:39-123 restated with
    the oracle's decode, on configs/baseline.yaml-style segments (2-D features, reference field set)."""
    c = cases.g6_inputs()
    n = c["n"]
    c["feature"] = np.concatenate([c["feat70"], tspn.hashrng.uniform(8, "rest", (c["feat70"].shape[0], 58))], axis=1)
    model = tspn.BaseModel(cases.baseline_cfg(**{"PREDICT.FEATURE_DIM": c["feature"].shape[1]}))
    model.eval()

    def segment(seed):
        feats = t(c["feature"]) + 0.001 * seed
        pl = tspn.PairList(feats)
        pl.add_field("tracklet_pairs", c["pairs"])                     # numpy, as vrdataset.py:75-81 leaves it
        pl.add_field("track_cls_logits", torch.zeros(n, 35))
        pl.add_field("num_tracklets", np.int64(n))
        pl.add_field("ious", np.eye(n, dtype=np.float32))
        pl.add_field("track_ids", -np.ones(n, dtype=np.int64))
        return pl

    lonely = tspn.PairList(torch.zeros((0, c["feature"].shape[1])))
    lonely.add_field("tracklet_pairs", np.zeros((0, 2), np.int64))
    lonely.add_field("track_cls_logits", torch.zeros(1, 35))
    lonely.add_field("num_tracklets", np.int64(1))
    loader = [([segment(0), segment(1)], None, [("v", 0, 30), ("v", 15, 45)]), ([lonely], None, [("v", 30, 60)])]
    seen = []
    rels = tspn.predict.predict_short_term_relations(model, loader, topk_per_pair=5, topk_per_seg=40,
                                                     on_segment=seen.append)
    assert seen == [("v", 0, 30), ("v", 15, 45), ("v", 30, 60)] and set(rels) == {("v", 0, 30), ("v", 15, 45)}
    for k, seed in ((("v", 0, 30), 0), (("v", 15, 45), 1)):
        preds, iou, tid = rels[k]
        feats = t(c["feature"]) + 0.001 * seed
        with torch.no_grad():
            lg = model([segment(seed)], None)[2][0]
        sc, trip, tids = oracle.decode_topk(lg, feats[:, :70], t(c["pairs"]), n, 5, 40)
        assert len(preds) == 40 and iou.shape == (n, n) and tid.shape == (n,)
        np.testing.assert_array_equal(np.array([p[0] for p in preds]), sc.numpy())
        np.testing.assert_array_equal(np.stack([p[1] for p in preds]), trip.numpy())
        np.testing.assert_array_equal(np.stack([p[2] for p in preds]), tids.numpy())
        assert preds[0][0].shape == () and preds[0][1].shape == (3,) and preds[0][2].shape == (2,)


def _oracle_train_reference(v, pairs, sd, gt_dur, gt_rel, targets):
    """The reference's intended DPN training step in plain torch autograd (CPU, float64): materialised
    pair features -> oracle.dpn_head (relpn/dpn.py:69-73) -> BCEWithLogits (dpn.py:44); RelOIPool over the
    segment -> RelationPredictor -> BCE (model.py:59-64).  Returns losses and parameter gradients."""
    w = {k: x.double().requires_grad_(True) for k, x in oracle_weights(sd).items()}
    pf, _ = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)
    rel, dur, _ = oracle.dpn_head(pf.double(), w["conv_w"], w["conv_b"], w["dur_w"], w["dur_b"], w["rel_w"], w["rel_b"])
    losses = {"loss_duration": torch.nn.functional.binary_cross_entropy_with_logits(dur, gt_dur.double())}
    if gt_rel is not None:
        losses["loss_relationness"] = torch.nn.functional.binary_cross_entropy_with_logits(rel, gt_rel.double())
    logit = oracle.predicate_head(pf.double().mean(dim=2), w["cls_w"], w["cls_b"])
    losses["loss_rel"] = torch.nn.functional.binary_cross_entropy(logit, targets.double())
    sum(losses.values()).backward()
    return {k: float(x) for k, x in losses.items()}, {k: x.grad for k, x in w.items()}


def test_dense_temporal_heads_input_gradient(tspn, device):
    """Round 4: the backward of the dense DPNHead runs on the library's HIP GEMM / conv kernels; the gradient with
    respect to the INPUT features (conv1d_input = the k=3 conv of dZ with reversed taps and swapped channel roles)
    and every parameter gradient against float64 autograd of relpn/dpn.py:69-73."""
    model_mod = __import__("importlib").import_module(tspn.BaseModel.__module__)
    P, C, T, H = 5, 24, 11, 12
    rs = np.random.RandomState(3)
    mk = lambda *shape: torch.from_numpy(rs.randn(*shape).astype(np.float32))   # noqa: E731
    x, cw, cb, hw, hb = mk(P, C, T), 0.3 * mk(C, C, 3), 0.1 * mk(C), 0.3 * mk(H, C), 0.1 * mk(H)
    gout = mk(P, H, T)
    dev_in = [v.clone().to(device).requires_grad_(True) for v in (x, cw, cb, hw, hb)]
    out = model_mod._TemporalHeadsDenseFn.apply(*dev_in)
    out.backward(gout.to(device))
    ref_in = [v.clone().double().requires_grad_(True) for v in (x, cw, cb, hw, hb)]
    act = torch.relu(torch.nn.functional.conv1d(ref_in[0], ref_in[1], ref_in[2], padding=1))
    ref = torch.nn.functional.conv1d(act, ref_in[3].unsqueeze(2), ref_in[4])
    ref.backward(gout.double())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-5)
    for name, a, b in zip(("x", "conv_w", "conv_b", "head_w", "head_b"), dev_in, ref_in):
        scale = max(1.0, float(b.grad.abs().max()))
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=2e-5 * scale, err_msg=name)


@pytest.mark.parametrize("form,with_relness", [("tracklets", True), ("tracklets", False), ("dense", True)])
def test_temporal_branch_training_losses_and_grads(tspn, device, form, with_relness):
    """USE_DPN=True in train mode (the reference's own path raises NameError, relpn/dpn.py:24-28): the loss
    dict of train.py:74-78 gains `loss_duration` (+ `loss_relationness`); losses and every parameter
    gradient equal the dense float64 autograd restatement — for tracklet samples (factorised encoder,
    gradients accumulated per tracklet) and for materialised [P,C,T] features."""
    D, N, T, A, K = 16, 6, 14, 4, 132
    sd = tspn.synth.make_weights(3, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False))
    load(model, sd)
    model.to(device).train()
    v = tspn.synth.make_video(90, N, T, D)
    pairs = oracle.pair_index(N)
    P = pairs.shape[0]
    gt_dur = t((tspn.hashrng.uniform(91, "gt_dur", (P, 2 * A, T)) < 0.3).astype(np.float32))
    gt_rel = t((tspn.hashrng.uniform(91, "gt_rel", (P, A, T)) < 0.3).astype(np.float32)) if with_relness else None
    targets = t((tspn.hashrng.uniform(91, "tg", (P, K)) < 0.05).astype(np.float32))
    if form == "tracklets":
        plist = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), t(v["track_cls_logits"]))
    else:
        plist = tspn.PairList(oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)[0])
    tlist = tspn.TargetList(targets)
    tlist.add_field("duration", gt_dur)
    if with_relness:
        tlist.add_field("relness", gt_rel)
    loss = model([plist.to(device)], [tlist.to(device)])
    want_keys = {"loss_duration", "loss_rel"} | ({"loss_relationness"} if with_relness else set())
    assert set(loss) == want_keys and all(x.dim() == 0 for x in loss.values())
    sum(loss.values()).backward()
    ref_loss, ref_grad = _oracle_train_reference(v, pairs, sd, gt_dur, gt_rel, targets)
    for k in want_keys:
        np.testing.assert_allclose(loss[k].item(), ref_loss[k], rtol=2e-5)
    h = model.relpn.duration_proposal_network.dpn_head
    got = {"conv_w": h.conv.weight.grad, "conv_b": h.conv.bias.grad,
           "dur_w": h.duration_pred.weight.grad, "dur_b": h.duration_pred.bias.grad,
           "cls_w": model.classifier.rel_predictor.weight.grad, "cls_b": model.classifier.rel_predictor.bias.grad}
    if with_relness:
        got.update({"rel_w": h.relness_pred.weight.grad, "rel_b": h.relness_pred.bias.grad})
    else:
        assert h.relness_pred.weight.grad is None or float(h.relness_pred.weight.grad.abs().max()) == 0.0
    for k, gq in got.items():
        r = ref_grad[k].numpy()
        np.testing.assert_allclose(gq.cpu().numpy(), r, rtol=0, atol=2e-4 * np.abs(r).max() + 1e-9, err_msg=k)
    # one optimiser step moves the loss down (the train.py loop: zero_grad / backward / step)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    opt.step()
    opt.zero_grad()
    loss2 = model([plist.to(device)], [tlist.to(device)])
    assert float(sum(loss2.values())) < float(sum(loss.values()))
    # eval after training uses the updated weights (device caches are refreshed by parameter version)
    model.eval()
    with torch.no_grad():
        _, dp, _ = model([plist], None)
    w2 = {k: x.detach().cpu() for k, x in (("conv_w", h.conv.weight), ("conv_b", h.conv.bias),
                                            ("dur_w", h.duration_pred.weight), ("dur_b", h.duration_pred.bias),
                                            ("rel_w", h.relness_pred.weight), ("rel_b", h.relness_pred.bias))}
    pf, _ = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)
    _, dur2, _ = oracle.dpn_head(pf, w2["conv_w"], w2["conv_b"], w2["dur_w"], w2["dur_b"], w2["rel_w"], w2["rel_b"])
    np.testing.assert_allclose(dp[0].duration.cpu().numpy(), dur2.numpy(), rtol=0, atol=1e-5)


def test_temporal_branch_training_errors(tspn, device):
    D, N, T = 16, 4, 8
    model = tspn.BaseModel(temporal_cfg(D, use_ppn=False)).to(device).train()
    v = tspn.synth.make_video(92, N, T, D)
    plist = tspn.PairList.from_tracklets(t(v["tracklet_feats"])).to(device)
    tl = tspn.TargetList(torch.zeros((12, 132)))
    with pytest.raises(KeyError):
        model([plist], [tl.to(device)])                       # no 'duration' field (dpn.py:30-32 reads it)
    tl.add_field("duration", torch.zeros((12, 8, T + 1)))
    with pytest.raises(ValueError):
        model([plist], [tl.to(device)])                       # wrong shape
    cpu_model = tspn.BaseModel(temporal_cfg(D, use_ppn=False)).train()
    with pytest.raises(RuntimeError):
        cpu_model([plist], [tl.to(device)])                   # parameters not on the HIP device


def test_forward_on_slices_of_one_batched_tensor_is_zero_copy_and_equal(tspn, device):
    """A loader that hands out `big[b*N:(b+1)*N]` slices of one device-resident [B*N,T,D] tensor (what
    bench.py --via-model does): the fused forward batches them back WITHOUT a copy (model._consecutive_view) and
    returns per-segment VIEWS of one batched output; results equal the forward on independent copies bit for
    bit, a second call reuses the module's workspace and pair table, and decode on the views equals decode on
    copies."""
    from tspn_mi355x import model as M
    D, N, T, B = 32, 5, 30, 3
    sd = tspn.synth.make_weights(4, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D))
    load(model, sd)
    model.eval()
    vids = [tspn.synth.make_video(300 + b, N, T, D) for b in range(B)]
    big = torch.cat([t(v["tracklet_feats"]) for v in vids]).to(device)
    boxes = torch.cat([t(v["tracklet_boxes"]) for v in vids]).to(device)
    cls = torch.stack([8.0 * t(v["track_cls_logits"]) for v in vids]).to(device)
    sl = [big[b * N:(b + 1) * N] for b in range(B)]
    view = M._consecutive_view(sl)
    assert view is not None and view.data_ptr() == big.data_ptr() and view.shape == big.shape
    assert M._consecutive_view([sl[0], sl[2]]) is None and M._consecutive_view([sl[1].clone(), sl[2]]) is None
    assert M._consecutive_view([big[1:N + 1], big[N + 1:2 * N + 1]]).data_ptr() == big[1:].data_ptr()
    plists = [tspn.PairList.from_tracklets(sl[b], boxes[b * N:(b + 1) * N], cls[b]) for b in range(B)]
    pp, dp, lg = model(plists, None)
    P = N * (N - 1)
    assert all(lg[b].data_ptr() == lg[0].data_ptr() + b * P * 132 * 4 for b in range(B))      # views of one output
    assert all(dp[b].heads.data_ptr() == dp[0].heads.data_ptr() + b * P * 12 * T * 4 for b in range(B))
    copies = [tspn.PairList.from_tracklets(sl[b].clone(), boxes[b * N:(b + 1) * N].clone(), cls[b].clone()) for b in range(B)]
    pp2, dp2, lg2 = model(copies, None)
    for b in range(B):
        assert torch.equal(lg[b], lg2[b]) and torch.equal(dp[b].heads, dp2[b].heads) and torch.equal(pp[b], pp2[b])
    assert len(model._workspaces) == 1 and len(model._pair_tables) == 1
    ws_ptr = next(iter(model._workspaces.values())).data_ptr()
    model(plists, None)
    assert next(iter(model._workspaces.values())).data_ptr() == ws_ptr
    d1 = model.decode(plists, lg)
    d2 = model.decode(copies, [x.clone() for x in lg2])
    for a, b in zip(d1, d2):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    w = oracle_weights(sd)
    ref = oracle.forward_dense(t(vids[1]["tracklet_feats"]), t(vids[1]["tracklet_boxes"]), oracle.pair_index(N), w)
    np.testing.assert_allclose(lg[1].cpu().numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(dp[1].duration.cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)



def test_overlapped_decode_with_separately_allocated_class_logits(tspn, device):
    """with per-video class logits that are NOT consecutive slices of one allocation (and not fp32),
    batching them is a cat / cast kernel; it has to run on the stream that decodes.  Sizes at which the encoder takes
    milliseconds, so that a cat queued behind it on the caller's stream would lose the race every time.  Also covers an
    explicit pair table with the geometry launch on the side stream (the table is built on the caller's stream)."""
    D, N, T, B = 512, 24, 150, 6
    sd = tspn.synth.make_weights(9, c=2 * D, bias_std=0.05)
    models = []
    for ov in (True, False):
        cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                    "PREDICT.FEATURE_DIM": 2 * D, "RELPN.OVERLAP_TAIL": ov,
                                    "RELPN.DPN.PAIR_GEOMETRY": True})
        m = tspn.BaseModel(cfg)
        load(m, sd)
        m.eval()
        models.append(m)
    for step in range(3):
        vids = [tspn.synth.make_video(700 + 10 * step + b, N, T, D) for b in range(B)]
        big = torch.cat([t(v["tracklet_feats"]) for v in vids]).to(device)
        boxes = torch.cat([t(v["tracklet_boxes"]) for v in vids]).to(device)
        pad = []
        plists = []
        for b, v in enumerate(vids):
            c = (8.0 * t(v["track_cls_logits"])).double().to(device)      # own allocation, needs a cast
            pad.append(torch.empty(1000 + 37 * b, device=device))            # keeps the allocations apart
            plists.append(tspn.PairList.from_tracklets(big[b * N:(b + 1) * N], boxes[b * N:(b + 1) * N], c))
        torch.cuda.synchronize()
        outs = []
        for m in models:
            pp, dp, lg = m(plists, None)
            dec = m.decode(plists, lg)
            torch.cuda.synchronize()
            outs.append((pp, dp, lg, dec))
        (pp0, dp0, lg0, d0), (pp1, dp1, lg1, d1) = outs
        for b in range(B):
            assert torch.equal(pp0[b], pp1[b]) and torch.equal(lg0[b], lg1[b])
            assert torch.equal(dp0[b].geom, dp1[b].geom)
            assert all(torch.equal(x, y) for x, y in zip(d0[b], d1[b]))
    # one segment with its own pair table: first use of that table is the geometry launch on the side stream
    v = tspn.synth.make_video(990, N, T, D)
    tab = torch.from_numpy(np.array([(i, j) for i in range(N) for j in range(N) if i != j][::3], dtype=np.int64))
    res = []
    for m in models:
        pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]).to(device), t(v["tracklet_boxes"]).to(device),
                                          (8.0 * t(v["track_cls_logits"])).to(device))
        pl.add_field("tracklet_pairs", tab)
        _, dp, lg = m([pl], None)
        torch.cuda.synchronize()
        res.append((dp[0].geom, lg[0]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("bf16", [False, True])
def test_overlapped_tail_equals_serial(tspn, device, bf16):
    """RELPN.OVERLAP_TAIL (default on): PPN and the top-k decode run on a second HIP stream under the encoder of the
    same forward, behind the logits-ready event, and join the caller's stream -- results equal the single-stream
    model bit for bit, over repeated steps on rotating inputs, also when the caller works on a non-default stream,
    when decode is called twice, and when it is handed logits that did not come from the last forward."""
    D, N, T, B = 32, 6, 30, 3
    sd = tspn.synth.make_weights(6, c=2 * D, bias_std=0.05)
    models = []
    for ov in (True, False):
        cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                    "PREDICT.FEATURE_DIM": 2 * D, "RELPN.OVERLAP_TAIL": ov})
        m = tspn.BaseModel(cfg)
        load(m, sd)
        m.eval()
        assert m.overlap_tail is ov
        models.append(m)
    st = torch.cuda.Stream(device=device)
    for step in range(4):
        vids = [tspn.synth.make_video(400 + 10 * step + b, N, T, D) for b in range(B)]
        big = torch.cat([t(v["tracklet_feats"]) for v in vids]).to(device)
        if bf16:
            big = big.to(torch.bfloat16)
        cls = torch.stack([8.0 * t(v["track_cls_logits"]) for v in vids]).to(device)
        boxes = torch.cat([t(v["tracklet_boxes"]) for v in vids]).to(device)
        plists = [tspn.PairList.from_tracklets(big[b * N:(b + 1) * N], boxes[b * N:(b + 1) * N], cls[b]) for b in range(B)]
        torch.cuda.synchronize()
        outs = []
        for m in models:
            ctx = torch.cuda.stream(st) if step % 2 else torch.cuda.stream(torch.cuda.current_stream(device))
            with ctx:
                pp, dp, lg = m(plists, None)
                dec = m.decode(plists, lg)
                dec2 = m.decode(plists, lg)                       # the event was consumed: plain path, same answer
                other = m.decode(plists, [x.clone() for x in lg])  # not the forward's tensor: plain path
            torch.cuda.synchronize()
            outs.append((pp, dp, lg, dec))
            for a, b in zip(dec, dec2):
                assert all(torch.equal(x, y) for x, y in zip(a, b))
            for a, b in zip(dec, other):
                assert all(torch.equal(x, y) for x, y in zip(a, b))
        (pp0, dp0, lg0, d0), (pp1, dp1, lg1, d1) = outs
        for b in range(B):
            assert torch.equal(pp0[b], pp1[b]) and torch.equal(lg0[b], lg1[b]) and torch.equal(dp0[b].heads, dp1[b].heads)
            assert all(torch.equal(x, y) for x, y in zip(d0[b], d1[b]))


def test_overlapped_geometry_equals_serial(tspn, device):
    """RELPN.DPN.PAIR_GEOMETRY with and without RELPN.OVERLAP_TAIL (geometry on the second stream), with and without PPN."""
    D, N, T, B = 32, 5, 30, 2
    sd = tspn.synth.make_weights(7, c=2 * D, bias_std=0.05)
    vids = [tspn.synth.make_video(500 + b, N, T, D) for b in range(B)]
    plists = [tspn.PairList.from_tracklets(t(v["tracklet_feats"]).to(device), t(v["tracklet_boxes"]).to(device),
                                           t(v["track_cls_logits"]).to(device)) for v in vids]
    for ppn in (True, False):
        res = []
        for ov in (True, False):
            m = tspn.BaseModel(cases.baseline_cfg(**{"RELPN.USE_PPN": ppn, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                                     "PREDICT.FEATURE_DIM": 2 * D, "RELPN.OVERLAP_TAIL": ov,
                                                     "RELPN.DPN.PAIR_GEOMETRY": True}))
            load(m, sd)
            m.eval()
            _, dp, lg = m(plists, None)
            torch.cuda.synchronize()
            res.append((dp, lg))
        for b in range(B):
            assert torch.equal(res[0][0][b].geom, res[1][0][b].geom) and torch.equal(res[0][1][b], res[1][1][b])
            ref = oracle.pair_geometry(t(vids[b]["tracklet_boxes"]), oracle.pair_index(N))
            np.testing.assert_allclose(res[0][0][b].geom.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("pinned", [False, True])
def test_host_inputs_pipeline_equals_resident(tspn, device, bf16, pinned):
    """predict.py:50-57 hands CPU PairLists.  The chunked upload / encoder / download pipeline (model._HostPipeline;
    pageable sources staged through pinned memory, pinned ones DMA'd in place) gives the resident path's results bit
    for bit, returns HOST tensors, survives buffer reuse over consecutive forwards with other data, and `decode` on the
    returned host logits (device copy reused) equals decode on the resident logits -- also after an in-place edit of
    the host logits, which must be seen."""
    D, N, T, B = 64, 7, 30, 7
    sd = tspn.synth.make_weights(12, c=2 * D, bias_std=0.05)
    cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                "PREDICT.FEATURE_DIM": 2 * D, "RELPN.DPN.PAIR_GEOMETRY": True,
                                "RELPN.DPN.HOST_CHUNK_VIDEOS": 2})
    m = tspn.BaseModel(cfg)
    load(m, sd)
    m.eval()
    assert m.host_chunk_videos == 2
    for step in range(3):
        vids = [tspn.synth.make_video(800 + 10 * step + b, N, T, D) for b in range(B)]
        host, res = [], []
        for v in vids:
            f = t(v["tracklet_feats"])
            if bf16:
                f = f.to(torch.bfloat16)
            c = 8.0 * t(v["track_cls_logits"])
            pl = tspn.PairList.from_tracklets(f, t(v["tracklet_boxes"]), c)
            host.append(pl.pin_memory() if pinned else pl)
            res.append(tspn.PairList.from_tracklets(f.to(device), t(v["tracklet_boxes"]).to(device), c.to(device)))
        assert host[0].get_field("tracklet_feats").is_pinned() is pinned
        pp_h, dp_h, lg_h = m(host, None)
        dec_h = m.decode(host, lg_h)
        pp_r, dp_r, lg_r = m(res, None)
        dec_r = m.decode(res, lg_r)
        torch.cuda.synchronize()
        for b in range(B):
            assert lg_h[b].device.type == "cpu" and dp_h[b].heads.device.type == "cpu" and dp_h[b].geom.device.type == "cpu"
            assert torch.equal(lg_h[b], lg_r[b].cpu()) and torch.equal(dp_h[b].heads, dp_r[b].heads.cpu())
            assert torch.equal(dp_h[b].geom, dp_r[b].geom.cpu()) and torch.equal(pp_h[b].cpu(), pp_r[b].cpu())
            assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(dec_h[b], dec_r[b]))
    # an edit of the returned host logits must reach decode (the cached device copy no longer applies)
    pp_h, dp_h, lg_h = m(host, None)
    lg_h[0].mul_(0.5)
    dec_e = m.decode(host, lg_h)
    lg_c = [x.clone() for x in lg_h]
    dec_c = m.decode(host, lg_c)
    for b in range(B):
        assert all(torch.equal(x, y) for x, y in zip(dec_e[b], dec_c[b]))
    assert not torch.equal(dec_e[0][0], dec_h[0][0])


def test_pair_list_pin_memory_and_non_blocking_to(tspn, device):
    """PairList.pin_memory() (what DataLoader(pin_memory=True) calls on custom batch types) and .to(device,
    non_blocking=True): tensors move, numpy fields stay on the host as in the reference (list_pair.py:28-31)."""
    pl = tspn.PairList(torch.arange(12.0).view(3, 4))
    pl.add_field("track_cls_logits", torch.ones(2, 35))
    pl.add_field("tracklet_pairs", np.array([[0, 1], [1, 0], [0, 1]]))
    pl.add_field("num_tracklets", 2)
    pin = pl.pin_memory()
    assert pin.features.is_pinned() and pin.get_field("track_cls_logits").is_pinned()
    assert isinstance(pin.get_field("tracklet_pairs"), np.ndarray) and pin.get_field("num_tracklets") == 2
    dv = pin.to(device, non_blocking=True)
    torch.cuda.synchronize()
    assert dv.features.is_cuda and torch.equal(dv.features.cpu(), pl.features)
    assert isinstance(dv.get_field("tracklet_pairs"), np.ndarray)
    assert torch.equal(pl.to(device).features, dv.features)


@pytest.mark.parametrize("pinned", [False, True])
def test_device_prefetcher_yields_the_loaders_batches_on_the_device(tspn, device, pinned):
    """dataset.DevicePrefetcher around a loader of host batches `(pair_list, target_list, indexs)` (the reference's
    collate, lib/dataset/build.py:84-93): same batches, tensors in HBM, numpy / int fields untouched, model results
    equal those on the host batches; works for an empty loader and for a single batch."""
    D, N, T, B = 32, 5, 30, 3
    sd = tspn.synth.make_weights(14, c=2 * D, bias_std=0.05)
    cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                "PREDICT.FEATURE_DIM": 2 * D})
    m = tspn.BaseModel(cfg)
    load(m, sd)
    m.eval()

    def batches(count):
        for step in range(count):
            pls = []
            for b in range(B):
                v = tspn.synth.make_video(900 + 10 * step + b, N, T, D)
                pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]).repeat(1, 40, 1)[:, :T * 40].contiguous()[:, :T],
                                                  t(v["tracklet_boxes"]), 8.0 * t(v["track_cls_logits"]))
                pl.add_field("ious", np.zeros((N, N), dtype=np.float32))
                pls.append(pl.pin_memory() if pinned else pl)
            yield pls, None, [("vid", 15 * step, 15 * step + 30)] * B

    assert list(tspn.dataset.DevicePrefetcher(batches(0), device)) == []
    for count in (1, 4):
        host = list(batches(count))
        got = list(tspn.dataset.DevicePrefetcher(batches(count), device))
        assert len(got) == count
        for (hp, _, hidx), (dp, dt, didx) in zip(host, got):
            assert dt is None and didx == hidx
            for a, b in zip(hp, dp):
                f = b.get_field("tracklet_feats")
                assert f.is_cuda and torch.equal(f.cpu(), a.get_field("tracklet_feats"))
                assert isinstance(b.get_field("ious"), np.ndarray) and b.get_field("num_tracklets") == N
            _, _, lg_h = m(hp, None)
            _, _, lg_d = m(dp, None)
            dec_h, dec_d = m.decode(hp, lg_h), m.decode(dp, lg_d)
            torch.cuda.synchronize()
            for b in range(B):
                assert lg_d[b].is_cuda and torch.equal(lg_d[b].cpu(), lg_h[b])
                assert all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(dec_h[b], dec_d[b]))


def test_in_place_data_edits_need_invalidate_or_verify_weights(tspn, device, monkeypatch):
    """`_DeviceCache` sees optimiser steps, load_state_dict and .to(); an in-place edit through `.data` bumps no version
    counter: the documented contract is `invalidate_caches()` -- or TSPN_VERIFY_WEIGHTS=1, which adds a content
    fingerprint to the cache signature (checked here by switching the module flag the environment variable sets)."""
    model_mod = __import__("importlib").import_module(tspn.BaseModel.__module__)
    c = cases.g1_inputs()
    m = tspn.BaseModel(cases.baseline_cfg())
    load(m, c["state_dict"])
    m.eval()
    plist = tspn.PairList(oracle.feature_preprocess(t(c["raw"])).to(device))
    base = m([plist], None)[2][0].clone()
    m.classifier.rel_predictor.weight.data.mul_(0.5)              # no version bump, same storage
    stale = m([plist], None)[2][0]
    assert torch.equal(stale, base)                                 # the cached packed copy is still in use (documented)
    m.invalidate_caches()
    fresh = m([plist], None)[2][0]
    assert not torch.equal(fresh, base)
    monkeypatch.setattr(model_mod, "_VERIFY_WEIGHTS", True)
    m.classifier.rel_predictor.weight.data.mul_(2.0)              # back to the original values, again in place
    again = m([plist], None)[2][0]
    assert torch.equal(again, base)                                 # seen without invalidate_caches()
