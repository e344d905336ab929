"""Winograd F(6,3) temporal conv (tspn_wino63.hip) through the C ABI: against the float64 conv for ragged shapes
(sextets masked at tracklet ends, tiles that straddle tracklets, 1 .. many super-stages, partial weight / sextet
tiles), its packed weights against the closed-form U = G g, its fp32 error next to the other algorithms at the
contraction depth of the headline config, and the fused path / BaseModel on it against the dense oracle."""
import numpy as np
import pytest
import torch

import cases
import oracle

pytestmark = pytest.mark.gpu

DPN_PRE = "relpn.duration_proposal_network.dpn_head."


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def conv_ref(x_tc, w, b, relu):
    """x [B,T,Cin], w [M,Cin,3] -> [B,M,T] float64."""
    y = torch.nn.functional.conv1d(t(x_tc).double().transpose(1, 2), t(w).double(),
                                   None if b is None else t(b).double(), padding=1)
    return (torch.relu(y) if relu else y).numpy()


@pytest.mark.parametrize("B,Cin,T,M", [(1, 32, 1, 32), (2, 32, 5, 32), (3, 64, 30, 128), (5, 96, 33, 160),
                                       (7, 64, 150, 64), (2, 160, 257, 288), (40, 32, 30, 96), (3, 32, 7, 32),
                                       (33, 128, 13, 128), (1, 32, 6, 32), (9, 256, 150, 256), (70, 32, 6, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv3_winograd63_vs_fp64(tspn, device, B, Cin, T, M, relu):
    x = tspn.hashrng.uniform(61, "x", (B, T, Cin), -1, 1)
    w = tspn.hashrng.normal(61, "w", (M, Cin, 3), std=0.1)
    b = tspn.hashrng.normal(61, "b", (M,), std=0.1)
    fr = tspn.ops.pack_conv3_wino63(t(w).to(device))
    assert tuple(fr.shape) == (M // 32, Cin // 8, 8, 64, 4) and tspn.ops.wino63_frag_dims(fr) == (Cin, M)
    # frag[mb][c][j][32 kh + li][e] = U_j[8 c + 4 kh + e][32 mb + li], U = G g in double, rounded once
    g = w.astype(np.float64).transpose(2, 1, 0)   # [3][Cin][M]
    u = np.stack([g[0], -2 / 9 * (g[0] + g[1] + g[2]), -2 / 9 * (g[0] - g[1] + g[2]),
                  g[0] / 90 + g[1] / 45 + 2 * g[2] / 45, g[0] / 90 - g[1] / 45 + 2 * g[2] / 45,
                  (32 * g[0] + 16 * g[1] + 8 * g[2]) / 45, (32 * g[0] - 16 * g[1] + 8 * g[2]) / 45, g[2]]).astype(np.float32)
    want = u.reshape(8, Cin // 8, 2, 4, M // 32, 32).transpose(4, 1, 0, 2, 5, 3)
    np.testing.assert_array_equal(fr.cpu().numpy().reshape(M // 32, Cin // 8, 8, 2, 32, 4), want)
    for bias in (b, None):
        y = tspn.ops.conv3_tc_wino63(t(x).to(device), fr, None if bias is None else t(bias).to(device), relu=relu)
        assert y.shape == (B, M, T)
        np.testing.assert_allclose(y.cpu().numpy(), conv_ref(x, w, bias, relu), rtol=0, atol=6e-5)
    y2 = tspn.ops.conv3_tc_wino63(t(x).to(device), fr, t(b).to(device), relu=relu)
    assert torch.equal(y, y2) if bias is not None else True     # deterministic


def test_conv3_winograd63_split_packing_and_errors(tspn, device):
    """split = D packs the subject / object halves of the pair encoder as 2M rows over D channels."""
    M, D = 32, 32
    w = tspn.hashrng.normal(62, "w", (M, 2 * D, 3), std=0.1)
    fr = tspn.ops.pack_conv3_wino63(t(w).to(device), split=D)
    stacked = np.concatenate([w[:, :D], w[:, D:]], axis=0)               # [2M, D, 3]
    fr2 = tspn.ops.pack_conv3_wino63(t(stacked).to(device))
    assert torch.equal(fr, fr2) and tuple(fr.shape) == (2, 4, 8, 64, 4)
    with pytest.raises(ValueError):
        tspn.ops.pack_conv3_wino63(torch.zeros((36, 32, 3), device=device))        # M % 32
    with pytest.raises(tspn._abi.TspnError) as e:                                   # Cin % 32 != 0: refused
        tspn.ops.conv3_tc_wino63(torch.zeros((2, 9, 24), device=device),
                                 tspn.ops.pack_conv3_wino63(torch.zeros((32, 24, 3), device=device)))
    assert e.value.code == tspn._abi.TSPN_EUNSUPPORTED
    y = tspn.ops.conv3_tc_wino63(torch.zeros((0, 9, 32), device=device), fr2)       # empty batch
    assert y.shape == (0, 64, 9)
    assert tspn._abi.lib().tspn_conv3_tc_wino63_workspace_bytes(16 * 32, 150, 2048) == 2048 // 4 * 8 * 12800 * 16


def test_conv3_winograd63_error_at_headline_depth(tspn, device):
    """fp32 error against float64 at K = 3 x 2048 channels, inputs in [0,1), weights N(0, 0.01^2): F(6,3) stays in
    the class of F(4,3) (the accumulation over the channels dominates, not the transforms) and an order of
    magnitude inside the path's 1e-4 bound."""
    B, T, Cin, M = 3, 150, 2048, 128
    x = tspn.hashrng.uniform(48, "x", (B, T, Cin))
    w = tspn.hashrng.normal(48, "w", (M, Cin, 3), std=0.01)
    ref = conv_ref(x, w, None, False)
    xd, wd = t(x).to(device), t(w).to(device)
    err = {"direct": np.abs(tspn.ops.conv3_tc(xd, tspn.ops.pack_conv3(wd)).cpu().numpy() - ref).max(),
           "F(4,3)": np.abs(tspn.ops.conv3_tc_wino43(xd, tspn.ops.pack_conv3_wino43(wd)).cpu().numpy() - ref).max(),
           "F(6,3)": np.abs(tspn.ops.conv3_tc_wino63(xd, tspn.ops.pack_conv3_wino63(wd)).cpu().numpy() - ref).max()}
    print("conv3 max abs error vs float64 (|y| max %.3f):" % np.abs(ref).max(), {k: "%.2e" % v for k, v in err.items()})
    assert err["F(6,3)"] <= 2.0 * err["F(4,3)"] and err["F(6,3)"] <= 3e-5


@pytest.mark.parametrize("B,N,T,D", [(2, 5, 30, 32), (1, 9, 33, 32), (3, 4, 150, 64)])
def test_fused_winograd63_vs_dense_oracle(tspn, device, B, N, T, D):
    """tspn_forward_fused_f32 with conv_algo 4 against the dense oracle (1e-5) and against conv_algo 3 (F(4,3))."""
    sd = tspn.synth.make_weights(50, c=2 * D, bias_std=0.05)
    w = {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
         "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
         "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
         "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}
    vids = [tspn.synth.make_video(60 + b, N, T, D) for b in range(B)]
    feats = torch.cat([t(v["tracklet_feats"]) for v in vids])
    pairs = torch.cat([oracle.pair_index(N) + b * N for b in range(B)])
    d = lambda v: v.to(device).contiguous()   # noqa: E731
    hw = d(torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]))
    hb = d(torch.cat([w["rel_b"], w["dur_b"]]))
    outs = []
    for packed in (tspn.ops.pack_conv3_wino63(d(w["conv_w"]), split=D),
                   tspn.ops.repack_wino43_frag(tspn.ops.pack_conv3_wino43(d(w["conv_w"]), split=D))):
        outs.append(tspn.ops.forward_fused(d(feats), d(pairs), B, N, packed, d(w["conv_b"]), hw, hb,
                                           d(w["cls_w"]), d(w["cls_b"]), canonical_pairs=True))
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=0, atol=1e-5)
    assert torch.equal(outs[0][1], outs[1][1])           # the logits do not depend on the conv algorithm
    for b in range(B):
        ref = oracle.forward_dense(t(vids[b]["tracklet_feats"]), t(vids[b]["tracklet_boxes"]), oracle.pair_index(N), w)
        sl = slice(b * N * (N - 1), (b + 1) * N * (N - 1))
        np.testing.assert_allclose(outs[0][0][sl, :4].cpu().numpy(), ref["relness"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(outs[0][0][sl, 4:].cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)


def test_basemodel_winograd6_config(tspn, device):
    """RELPN.DPN.CONV_ALGO = "winograd6" selects the F(6,3) weights (and falls back to F(4,3) when D % 32 != 0)."""
    for D, want in ((32, (8, 64, 4)), (16, (6, 64, 4))):
        cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": False, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                    "PREDICT.FEATURE_DIM": 2 * D, "RELPN.DPN.CONV_ALGO": "winograd6"})
        model = tspn.BaseModel(cfg)
        sd = tspn.synth.make_weights(3, c=2 * D, bias_std=0.05)
        own = model.state_dict()
        model.load_state_dict({k: t(v) for k, v in sd.items() if k in own})
        model.eval()
        v = tspn.synth.make_video(77, 6, 30, D)
        pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), t(v["track_cls_logits"]))
        _, dp, logits = model([pl], None)
        caches = model.relpn.duration_proposal_network._cache._store
        packed = [val[1][0] for key, val in caches.items() if key.startswith("conv_split")]
        assert len(packed) == 1 and tuple(packed[0].shape[2:]) == want
        w = {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
             "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
             "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
             "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}
        ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(6), w)
        np.testing.assert_allclose(dp[0].duration.numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(logits[0].numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)
