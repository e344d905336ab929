"""bf16-operand path (BASELINE config 3: N=64, T=900, D=1024, bf16) on the GPU against the oracle's
bf16 restatement (pinned by golden g8) and float64 references of the single kernels.

Tolerances: products of bf16 values are exact in fp32, so the kernels differ from a float64
evaluation of the same bf16 operands only by fp32 accumulation order (~1e-6 relative to the sum of
magnitudes) and — downstream of a rounding point — by rare one-ulp flips of a bf16 activation."""
import numpy as np
import pytest
import torch

import cases
import oracle

pytestmark = pytest.mark.gpu

DPN_PRE = "relpn.duration_proposal_network.dpn_head."


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def r16(x):
    """bf16 rounding on the host (torch's cast), kept in fp32."""
    return t(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).float()


def test_cast_bf16_bit_exact(tspn, device):
    x = tspn.hashrng.normal(80, "x", (70001,), std=3.0)
    x[:8] = [0.0, -0.0, 1.0, 1.00390625, 1.001953125, 3.3895314e38, 1e-40, -2.5]  # ties, overflow edge, denormal
    got = tspn.ops.cast_bf16(t(x).to(device)).cpu()
    assert got.dtype == torch.bfloat16
    assert torch.equal(got.view(torch.int16), t(x).to(torch.bfloat16).view(torch.int16))


def test_pack_layouts_bf16(tspn, device):
    M, Cin = 12, 32
    w = tspn.hashrng.normal(81, "w", (M, Cin, 3), std=0.1)
    p = tspn.ops.pack_conv3_bf16(t(w).to(device)).cpu().float().numpy()        # [3, Cin/8, M, 8]
    exp = r16(w).numpy().transpose(2, 1, 0).reshape(3, Cin // 8, 8, M).transpose(0, 1, 3, 2)
    np.testing.assert_array_equal(p, exp)
    half = Cin // 2
    ps = tspn.ops.pack_conv3_bf16(t(w).to(device), split=half).cpu().float().numpy()   # [3, half/8, 2M, 8]
    ws = np.concatenate([w[:, :half], w[:, half:]], axis=0)                     # [2M, half, 3]
    exps = r16(ws).numpy().transpose(2, 1, 0).reshape(3, half // 8, 8, 2 * M).transpose(0, 1, 3, 2)
    np.testing.assert_array_equal(ps, exps)
    hw = tspn.hashrng.normal(81, "hw", (12, 64), std=0.1)
    hp = tspn.ops.pack_heads_bf16(t(hw).to(device)).cpu().float().numpy()       # [C/8, 16, 8]
    exph = np.zeros((8, 16, 8), np.float32)
    exph[:, :12] = r16(hw).numpy().reshape(12, 8, 8).transpose(1, 0, 2)
    np.testing.assert_array_equal(hp, exph)


def conv_ref64(xb, wb, b):
    """x [B,T,Cin], w [M,Cin,3] (bf16-valued) -> [B,T,M] float64."""
    y = torch.nn.functional.conv1d(xb.double().transpose(1, 2), wb.double(), None if b is None else b.double(),
                                   padding=1)
    return y.transpose(1, 2).contiguous()


@pytest.mark.parametrize("B,T,Cin,M", [(1, 1, 16, 4), (2, 5, 32, 8), (3, 30, 64, 128), (5, 33, 48, 132),
                                       (2, 257, 128, 260), (7, 150, 64, 64), (40, 30, 32, 36)])
def test_conv3_bf16_vs_fp64(tspn, device, B, T, Cin, M):
    x = r16(tspn.hashrng.uniform(82, "x", (B, T, Cin), -1, 1))
    w = tspn.hashrng.normal(82, "w", (M, Cin, 3), std=0.1)
    b = tspn.hashrng.normal(82, "b", (M,), std=0.1)
    packed = tspn.ops.pack_conv3_bf16(t(w).to(device))
    y = tspn.ops.conv3_tc_bf16(x.to(torch.bfloat16).to(device), packed, t(b).to(device))
    ref = conv_ref64(x, r16(w), t(b))
    assert y.shape == (B, T, M)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    y0 = tspn.ops.conv3_tc_bf16(x.to(torch.bfloat16).to(device), packed, None)
    np.testing.assert_allclose(y0.cpu().numpy(), conv_ref64(x, r16(w), None).numpy(), rtol=0, atol=2e-5)


def test_conv3_bf16_exact_integers_and_limits(tspn, device):
    """Small-integer operands are bf16-exact and every partial sum is fp32-exact: bit-identical to
    the float64 conv, which checks the MFMA fragment maps and the sequence-end masks exactly."""
    B, C, T, M = 3, 64, 131, 132
    x = ((np.arange(B * T * C, dtype=np.float32).reshape(B, T, C) * 7) % 23) - 11.0
    w = ((np.arange(M * C * 3, dtype=np.float32).reshape(M, C, 3) * 5) % 9) - 4.0
    y = tspn.ops.conv3_tc_bf16(t(x).to(torch.bfloat16).to(device), tspn.ops.pack_conv3_bf16(t(w).to(device)))
    np.testing.assert_array_equal(y.cpu().numpy(), conv_ref64(t(x), t(w), None).float().numpy())
    with pytest.raises(tspn._abi.TspnError) as e:
        tspn.ops.conv3_tc_bf16(torch.zeros(2, 30, 8, dtype=torch.bfloat16, device=device),
                               torch.zeros(3, 1, 8, 8, dtype=torch.bfloat16, device=device))
    assert e.value.code == tspn._abi.TSPN_EUNSUPPORTED
    with pytest.raises(RuntimeError):
        tspn.ops.conv3_tc_bf16(torch.zeros(2, 30, 32, dtype=torch.bfloat16), torch.zeros(3, 4, 8, 8, dtype=torch.bfloat16))


def heads_ref64(y, B, N, hw, hb):
    """y [B*N,T,2C] fp32; -> [B*N*(N-1), H, T] float64 with the kernel's rounding point."""
    C = y.shape[2] // 2
    out = []
    for b in range(B):
        yy = y[b * N:(b + 1) * N]
        pairs = oracle.pair_index(N)
        a = torch.relu(yy[pairs[:, 0], :, :C] + yy[pairs[:, 1], :, C:])        # fp32 add, as on the GPU
        a = a.to(torch.bfloat16).double()                                       # [P,T,C]
        out.append(torch.einsum("ptc,hc->pht", a, hw.double()) + hb.double().view(1, -1, 1))
    return torch.cat(out)


@pytest.mark.parametrize("B,N,T,C", [(1, 2, 1, 32), (2, 5, 30, 64), (1, 8, 16, 32), (1, 11, 37, 96), (3, 9, 150, 64),
                                     (1, 17, 20, 32)])
def test_heads_pairgrid_bf16_vs_fp64(tspn, device, B, N, T, C):
    y = t(tspn.hashrng.normal(83, "y", (B * N, T, 2 * C), std=1.0))
    hw = r16(tspn.hashrng.normal(83, "hw", (12, C), std=0.1))
    hb = t(tspn.hashrng.normal(83, "hb", (12,), std=0.1))
    out = tspn.ops.heads_pairgrid_bf16(y.to(device), B, N, tspn.ops.pack_heads_bf16(hw.to(device)), hb.to(device), 12)
    ref = heads_ref64(y, B, N, hw, hb)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=3e-5)


def test_temporal_mean_bf16(tspn, device):
    x = r16(tspn.hashrng.uniform(84, "x", (37, 53, 40)))
    got = tspn.ops.temporal_mean_bf16(x.to(torch.bfloat16).to(device)).cpu()
    ref = x.double().mean(dim=1)
    np.testing.assert_array_equal(got.numpy(), r16(got.numpy()).numpy())        # values are bf16
    same = (got == ref.float().to(torch.bfloat16).float()).float().mean()
    assert same > 0.995
    assert float((got.double() - ref).abs().max()) <= 2.0 ** -8                 # <= 1 ulp below 1.0


def oracle_weights(sd):
    return {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
            "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
            "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
            "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}


def temporal_cfg(D):
    return cases.baseline_cfg(**{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True,
                                 "RELPN.DPN.IN_CHANNELS": 2 * D, "PREDICT.FEATURE_DIM": 2 * D})


def check_against_oracle(out, ref, what):
    scale = max(float(ref.abs().max()), 1e-3)
    err = float((out - ref).abs().max())
    # one flipped bf16 activation moves a head output by ~2^-8 |a| |w|; allow a few of them
    assert err <= 2e-3 * scale, (what, err, scale)


def test_model_forward_bf16_tracklets_vs_oracle(tspn, device):
    """BaseModel.forward on bf16 tracklet features (ragged batch) == oracle.forward_bf16."""
    D = 32
    sd = tspn.synth.make_weights(0, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(temporal_cfg(D))
    own = model.state_dict()
    model.load_state_dict({k: t(v) for k, v in sd.items() if k in own})
    model.eval()
    shapes = [(6, 30), (9, 17), (6, 30), (3, 1)]
    vids = [tspn.synth.make_video(90 + i, n, tt, D) for i, (n, tt) in enumerate(shapes)]
    plists = [tspn.PairList.from_tracklets(t(v["tracklet_feats"]).to(torch.bfloat16), t(v["tracklet_boxes"]),
                                           t(v["track_cls_logits"])) for v in vids]
    pp, dp, logits = model(plists, None)
    w = oracle_weights(sd)
    for i, v in enumerate(vids):
        n = shapes[i][0]
        ref = oracle.forward_bf16(t(v["tracklet_feats"]), oracle.pair_index(n), w)
        assert dp[i].relness.dtype == torch.float32 and dp[i].relness.shape == ref["relness"].shape
        check_against_oracle(dp[i].relness, ref["relness"], "relness")
        check_against_oracle(dp[i].duration, ref["duration"], "duration")
        check_against_oracle(logits[i], ref["rel_logits"], "rel_logits")
        assert logits[i].device.type == "cpu" and pp[i].shape == (min(256, n * n),)


def test_forward_fused_bf16_matches_fp32_path_on_rounded_operands(tspn, device):
    """Mid-size property check (N=16, T=150, D=128): the bf16 kernels agree with the fp32 HIP path
    run on the same bf16-rounded operands up to the one activation rounding (2^-8 relative)."""
    N, T, D, A, K = 16, 150, 128, 4, 132
    C = 2 * D
    v = tspn.synth.make_video(95, N, T, D)
    sd = tspn.synth.make_weights(0, c=C, bias_std=0.05)
    w = {k: r16(x.numpy()).to(device) for k, x in oracle_weights(sd).items()}
    feats = r16(v["tracklet_feats"]).to(device)
    pairs = tspn.ops.pair_index(N, device)
    hw = torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]).contiguous()
    hb = torch.cat([w["rel_b"], w["dur_b"]]).contiguous()
    h16, l16 = tspn.ops.forward_fused_bf16(feats.to(torch.bfloat16), pairs, 1, N,
                                           tspn.ops.pack_conv3_bf16(w["conv_w"], split=D), w["conv_b"],
                                           tspn.ops.pack_heads_bf16(hw), hb, w["cls_w"], w["cls_b"])
    h32, l32 = tspn.ops.forward_fused(feats, pairs, 1, N, tspn.ops.pack_conv3(w["conv_w"], split=D), w["conv_b"],
                                      hw, hb, w["cls_w"], w["cls_b"], canonical_pairs=True)
    scale = float(h32.abs().max())
    assert float((h16 - h32).abs().max()) <= 2.0 ** -8 * scale
    assert float((l16 - l32).abs().max()) <= 2.0 ** -8
    # and a checksum of checksums is reproducible run to run (no races)
    h16b, l16b = tspn.ops.forward_fused_bf16(feats.to(torch.bfloat16), pairs, 1, N,
                                             tspn.ops.pack_conv3_bf16(w["conv_w"], split=D), w["conv_b"],
                                             tspn.ops.pack_heads_bf16(hw), hb, w["cls_w"], w["cls_b"])
    assert torch.equal(h16, h16b) and torch.equal(l16, l16b)


def test_bf16_path_rejects_what_it_cannot_do(tspn, device):
    D = 8
    model = tspn.BaseModel(temporal_cfg(D))
    model.eval()
    v = tspn.synth.make_video(96, 4, 10, D)
    pl = tspn.PairList.from_tracklets(t(v["tracklet_feats"]).to(torch.bfloat16), t(v["tracklet_boxes"]),
                                      t(v["track_cls_logits"]))
    with pytest.raises(ValueError, match="D % 16"):
        model([pl], None)


def test_cfg3_full_size_bf16_vs_fp32_path(tspn, device):
    """BASELINE config 3 at full size (N=64, T=900, D=1024 -> P=4032, C=2048): the bf16 kernels against
    the fp32 HIP path (itself oracle-checked) on the same bf16-rounded operands: they differ only by the
    one rounding of the encoder activation / pooled feature (<= 2^-8 relative to the output range), and
    the result is bit-reproducible run to run.  Also exercises 3600-workgroup grids, T % 16 != 0 and the
    1-video launch (912 pair-stage workgroups on 256 CUs)."""
    N, T, D = 64, 900, 1024
    C = 2 * D
    v = tspn.synth.make_video(97, N, T, D)
    sd = tspn.synth.make_weights(0, c=C, bias_std=0.05)
    w = {k: r16(x.numpy()).to(device) for k, x in oracle_weights(sd).items()}
    feats = r16(v["tracklet_feats"]).to(device)
    pairs = tspn.ops.pair_index(N, device)
    hw = torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]).contiguous()
    hb = torch.cat([w["rel_b"], w["dur_b"]]).contiguous()
    args16 = (feats.to(torch.bfloat16), pairs, 1, N, tspn.ops.pack_conv3_bf16(w["conv_w"], split=D), w["conv_b"],
              tspn.ops.pack_heads_bf16(hw), hb, w["cls_w"], w["cls_b"])
    h16, l16 = tspn.ops.forward_fused_bf16(*args16)
    h32, l32 = tspn.ops.forward_fused(feats, pairs, 1, N, tspn.ops.pack_conv3_wino63(w["conv_w"], split=D),
                                      w["conv_b"], hw, hb, w["cls_w"], w["cls_b"], canonical_pairs=True)
    assert h16.shape == (N * (N - 1), 12, T) and l16.shape == (N * (N - 1), 132)
    scale = float(h32.abs().max())
    assert float((h16 - h32).abs().max()) <= 2.0 ** -8 * scale
    assert float((l16 - l32).abs().max()) <= 2.0 ** -8
    assert torch.isfinite(h16).all() and torch.isfinite(l16).all()
    h16b, l16b = tspn.ops.forward_fused_bf16(*args16)
    assert torch.equal(h16, h16b) and torch.equal(l16, l16b)
    # span decode of the bf16 heads runs at T=900 (A*T = 3600 candidates per pair)
    sizes = [(a + 1) * T / 4 for a in range(4)]
    res = tspn.ops.decode_spans(h16[:64].contiguous(), sizes, top_k=16)
    assert res["span"].shape == (64, 16, 2) and int(res["count"].min()) >= 1
    ok = res["span"][..., 0] >= 0
    assert bool((res["span"][..., 1][ok] <= T).all()) and bool((res["span"][..., 0][ok] < res["span"][..., 1][ok]).all())


def test_bf16_and_span_entry_points_edge_cases(tspn, device):
    """Empty and degenerate shapes of the newer entry points: zero videos / pairs return empty outputs,
    T = 1 and one-frame spans work, bad arguments are reported through the ABI's error convention."""
    ops = tspn.ops
    D, C = 16, 32
    w = torch.zeros((C, C, 3), device=device)
    packed = ops.pack_conv3_bf16(w, split=D)
    hp = ops.pack_heads_bf16(torch.zeros((12, C), device=device))
    z = lambda *s: torch.zeros(s, device=device)     # noqa: E731
    # N = 1: no pairs
    h, l = ops.forward_fused_bf16(torch.zeros((1, 5, D), dtype=torch.bfloat16, device=device),
                                  torch.zeros((0, 2), dtype=torch.int64, device=device), 1, 1, packed, z(C), hp,
                                  z(12), z(132, C), z(132))
    assert h.shape == (0, 12, 5) and l.shape == (0, 132)
    # T = 1 (both sequence-end masks on the same column)
    x = r16(tspn.hashrng.uniform(85, "x", (3, 1, D), -1, 1))
    wc = tspn.hashrng.normal(85, "w", (8, D, 3), std=0.1)
    y = ops.conv3_tc_bf16(x.to(torch.bfloat16).to(device), ops.pack_conv3_bf16(t(wc).to(device)))
    ref = torch.einsum("btc,mc->btm", x.double(), r16(wc)[:, :, 1].double())
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)
    # span pooling: empty pair list, one-frame spans, error code for a short workspace via the raw ABI
    feats = t(tspn.hashrng.uniform(86, "f", (4, 6, D))).to(device)
    cw, cb = t(tspn.hashrng.normal(86, "w", (132, 2 * D), std=0.1)).to(device), z(132)
    e = ops.span_predicate(feats, torch.zeros((0, 2), dtype=torch.int64, device=device),
                           torch.zeros((0, 2), dtype=torch.int64, device=device), cw, cb)
    assert e.shape == (0, 132)
    pairs = ops.pair_index(4, device)
    one = torch.tensor([[3, 4]] * 12, dtype=torch.int64, device=device)
    got = ops.span_predicate(feats, pairs, one, cw, cb).cpu()
    f3 = feats.cpu()[:, 3]
    ref = torch.sigmoid(torch.cat([f3[pairs.cpu()[:, 0]], f3[pairs.cpu()[:, 1]]], 1).double() @ cw.cpu().double().t())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-6)
    with pytest.raises(IndexError):
        ops.span_predicate(feats, pairs + 4, one, cw, cb)
    lib = tspn._abi.lib()
    import ctypes
    rc = lib.tspn_span_predicate_f32(ctypes.c_void_p(feats.data_ptr()), 4, 6, D, ctypes.c_void_p(pairs.data_ptr()),
                                     ctypes.c_void_p(one.data_ptr()), 12, ctypes.c_void_p(cw.data_ptr()),
                                     ctypes.c_void_p(cb.data_ptr()), 132, ctypes.c_void_p(got.data_ptr()),
                                     ctypes.c_void_p(feats.data_ptr()), 16, None)
    assert rc == tspn._abi.TSPN_EWORKSPACE and b"workspace" in lib.tspn_last_error()


@pytest.mark.parametrize("case", range(10))
def test_random_shapes_fp32_and_bf16_paths_vs_oracle(tspn, device, case):
    """Seeded random (videos, N, T, D) draws: ragged tiles in every dimension for both operand types."""
    rs = np.random.RandomState(1000 + case)
    B, N, T, D = int(rs.randint(1, 4)), int(rs.randint(2, 21)), int(rs.randint(1, 71)), int(rs.choice([16, 32, 48]))
    C = 2 * D
    sd = tspn.synth.make_weights(case, c=C, bias_std=0.05)
    w = oracle_weights(sd)
    vids = [tspn.synth.make_video(500 + 10 * case + b, N, T, D) for b in range(B)]
    model = tspn.BaseModel(temporal_cfg(D))
    own = model.state_dict()
    model.load_state_dict({k: t(v) for k, v in sd.items() if k in own})
    model.eval()
    for dtype in (torch.float32, torch.bfloat16):
        plists = [tspn.PairList.from_tracklets(t(v["tracklet_feats"]).to(dtype), t(v["tracklet_boxes"]),
                                               t(v["track_cls_logits"])) for v in vids]
        _, dp, logits = model(plists, None)
        for b, v in enumerate(vids):
            if dtype == torch.float32:
                ref = oracle.forward_dense(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), oracle.pair_index(N), w)
                for got, key in ((dp[b].relness, "relness"), (dp[b].duration, "duration"), (logits[b], "rel_logits")):
                    np.testing.assert_allclose(got.numpy(), ref[key].numpy(), rtol=0, atol=1e-5, err_msg=f"{key} {B,N,T,D}")
            else:
                ref = oracle.forward_bf16(t(v["tracklet_feats"]), oracle.pair_index(N), w)
                check_against_oracle(dp[b].relness, ref["relness"], f"relness {B,N,T,D}")
                check_against_oracle(dp[b].duration, ref["duration"], f"duration {B,N,T,D}")
                check_against_oracle(logits[b], ref["rel_logits"], f"rel_logits {B,N,T,D}")


def test_fused_passes_are_graph_capturable(tspn, device):
    """The ABI neither allocates nor synchronises and takes the stream explicitly: after
    one warm-up call (which sets kernel attributes) a whole fused pass -- fp32 and bf16 -- is captured
    into a HIP graph and replayed on new inputs with the results of a direct call."""
    N, T, D, B = 9, 30, 32, 2
    C = 2 * D
    sd = tspn.synth.make_weights(3, c=C, bias_std=0.05)
    w = {k: x.to(device) for k, x in oracle_weights(sd).items()}
    hw = torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]).contiguous()
    hb = torch.cat([w["rel_b"], w["dur_b"]]).contiguous()
    pairs = torch.cat([tspn.ops.pair_index(N, device, base=b * N) for b in range(B)])
    P = pairs.shape[0]
    f_a = t(np.concatenate([tspn.synth.make_video(130 + b, N, T, D)["tracklet_feats"] for b in range(B)])).to(device)
    f_b = t(np.concatenate([tspn.synth.make_video(140 + b, N, T, D)["tracklet_feats"] for b in range(B)])).to(device)
    # ---- fp32
    packed = tspn.ops.pack_conv3_wino63(w["conv_w"], split=D)
    ws = torch.empty(tspn.ops.fused_workspace_bytes(B, N, T, D, 4, 132, P), dtype=torch.uint8, device=device)
    oh, ol = torch.empty((P, 12, T), device=device), torch.empty((P, 132), device=device)
    feats = f_a.clone()
    run = lambda: tspn.ops.forward_fused(feats, pairs, B, N, packed, w["conv_b"], hw, hb, w["cls_w"], w["cls_b"],   # noqa: E731
                                         workspace=ws, out_heads=oh, out_logits=ol, check_pairs=False,
                                         canonical_pairs=True)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    feats.copy_(f_b)
    g.replay()
    torch.cuda.synchronize()
    got_h, got_l = oh.clone(), ol.clone()
    ref_h, ref_l = tspn.ops.forward_fused(f_b, pairs, B, N, packed, w["conv_b"], hw, hb, w["cls_w"], w["cls_b"],
                                          canonical_pairs=True)
    assert torch.equal(got_h, ref_h) and torch.equal(got_l, ref_l)
    # ---- bf16
    p16, h16 = tspn.ops.pack_conv3_bf16(w["conv_w"], split=D), tspn.ops.pack_heads_bf16(hw)
    d16 = tspn._abi.FusedBf16Desc()
    d16.B, d16.N, d16.T, d16.D, d16.A, d16.K, d16.P = B, N, T, D, 4, 132, P
    ws16 = torch.empty(tspn._abi.lib().tspn_forward_fused_bf16_workspace_bytes(d16), dtype=torch.uint8, device=device)
    x16 = f_a.to(torch.bfloat16)
    out = {}

    def run16():
        out["h"], out["l"] = tspn.ops.forward_fused_bf16(x16, pairs, B, N, p16, w["conv_b"], h16, hb, w["cls_w"],
                                                         w["cls_b"], workspace=ws16)
    run16()
    torch.cuda.synchronize()
    g16 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g16):
        run16()
    x16.copy_(f_b.to(torch.bfloat16))
    g16.replay()
    torch.cuda.synchronize()
    rh, rl = tspn.ops.forward_fused_bf16(f_b.to(torch.bfloat16), pairs, B, N, p16, w["conv_b"], h16, hb, w["cls_w"], w["cls_b"])
    assert torch.equal(out["h"], rh) and torch.equal(out["l"], rl)


@pytest.mark.parametrize("tag", list(cases.G3_SHAPES))
def test_dense_bf16_encoder_heads_vs_reference_bf16_modules_g8(tspn, device, tag):
    """tspn_temporal_encoder_heads_bf16 (the reference-faithful dense form on a materialised [P,C,T]) against
    golden g8 = the reference's own DPNHead cast with .bfloat16() (dpn.py:55-73, torch CPU bf16 kernels) on the G3
    inputs: rounding the HIP output to bf16 reproduces the reference's output on (nearly) every element and never
    differs by more than two bf16 ulps; against the oracle's unrounded restatement within fp32 accumulation."""
    g = cases.load("g8_bf16.npz")
    c = cases.g3_inputs(tag)
    sd = {k: t(v) for k, v in c["state_dict"].items()}
    x = t(c["x"])
    P, C, T = x.shape
    d = lambda v: v.to(device).contiguous()   # noqa: E731
    hw = torch.cat([sd[DPN_PRE + "relness_pred.weight"][:, :, 0], sd[DPN_PRE + "duration_pred.weight"][:, :, 0]])
    hb = torch.cat([sd[DPN_PRE + "relness_pred.bias"], sd[DPN_PRE + "duration_pred.bias"]])
    A = sd[DPN_PRE + "relness_pred.bias"].numel()
    out = tspn.ops.temporal_encoder_heads_bf16(
        d(x), tspn.ops.pack_conv3_bf16(d(sd[DPN_PRE + "conv.weight"])), d(r16(sd[DPN_PRE + "conv.bias"])),
        tspn.ops.pack_heads_bf16(d(hw)), d(r16(hb)), 3 * A).cpu()
    assert out.shape == (P, 3 * A, T)
    g11 = cases.load("g11_relness_head.npz")   # dpn_anchor.py:82-108's DPNHead.bfloat16(): the relationness leg
    for got, ref in ((out[:, A:], g[f"{tag}_duration"]), (out[:, :A], g11[f"{tag}_relness_bf16"])):
        same = (oracle.bf16_round(got).numpy() == ref).mean()
        assert same > 0.999, same
        ulp = np.maximum(np.abs(ref), 2.0 ** -126) * 2.0 ** -7
        assert np.all(np.abs(got.numpy() - ref) <= 2 * ulp)
    rel_o, dur_o, _ = oracle.dpn_head_bf16(x, sd[DPN_PRE + "conv.weight"], sd[DPN_PRE + "conv.bias"],
                                           sd[DPN_PRE + "duration_pred.weight"], sd[DPN_PRE + "duration_pred.bias"],
                                           sd[DPN_PRE + "relness_pred.weight"], sd[DPN_PRE + "relness_pred.bias"])
    exp = torch.cat([rel_o, dur_o], dim=1)
    scale = float(exp.abs().max())
    # a bf16 activation that sits on a rounding boundary may flip by one ulp (2^-8 relative) between fp32 and
    # float64 accumulation of the conv: a few elements move by up to |w_head| * ulp
    assert float((out - exp).abs().max()) <= 2e-3 * scale
    assert float((out - exp).abs().mean()) <= 2e-6 * scale


def test_dense_bf16_pieces_ragged_and_errors(tspn, device):
    """transpose_cast_bf16 is bit-exact; heads_dense on ragged T (not a multiple of 16) and H < 16 against float64;
    channels-last bf16 input is accepted as is; bad shapes raise."""
    P, C, T, H = 5, 96, 37, 12
    x = tspn.hashrng.uniform(91, "x", (P, C, T), -1, 1)
    xt = tspn.ops.transpose_cast_bf16(t(x).to(device))
    assert xt.shape == (P, T, C) and xt.dtype == torch.bfloat16
    assert torch.equal(xt.cpu().view(torch.int16), t(x).transpose(1, 2).contiguous().to(torch.bfloat16).view(torch.int16))
    cw = tspn.hashrng.normal(91, "cw", (C, C, 3), std=0.08)
    cb = r16(tspn.hashrng.normal(91, "cb", (C,), std=0.1))
    hw = tspn.hashrng.normal(91, "hw", (H, C), std=0.1)
    hb = r16(tspn.hashrng.normal(91, "hb", (H,), std=0.1))
    d = lambda v: v.to(device).contiguous()   # noqa: E731
    packed, hp = tspn.ops.pack_conv3_bf16(d(t(cw))), tspn.ops.pack_heads_bf16(d(t(hw)))
    out = tspn.ops.temporal_encoder_heads_bf16(d(t(x)), packed, d(cb), hp, d(hb), H).cpu()
    out2 = tspn.ops.temporal_encoder_heads_bf16(xt, packed, d(cb), hp, d(hb), H).cpu()
    assert torch.equal(out, out2)
    h = torch.nn.functional.conv1d(r16(x).double(), r16(cw).double(), cb.double(), padding=1)
    h = oracle.bf16_round(torch.relu(h).float()).double()
    exp = torch.einsum("hc,pct->pht", r16(hw).double(), h) + hb.double()[None, :, None]
    scale = float(exp.abs().max())
    assert float((out.double() - exp).abs().max()) <= 2e-3 * scale
    assert float((out.double() - exp).abs().mean()) <= 2e-6 * scale
    with pytest.raises(ValueError):
        tspn.ops.temporal_encoder_heads_bf16(d(t(x)), packed, d(cb), hp, d(hb), H - 1)
    with pytest.raises(RuntimeError):   # C % 32 != 0 is refused by the library
        x2 = t(tspn.hashrng.uniform(92, "x", (2, 48, 8), -1, 1))
        tspn.ops.temporal_encoder_heads_bf16(d(x2), tspn.ops.pack_conv3_bf16(d(t(tspn.hashrng.normal(92, "w", (48, 48, 3))))),
                                             None, tspn.ops.pack_heads_bf16(d(t(tspn.hashrng.normal(92, "h", (H, 48))))),
                                             d(hb), H)
    assert tspn.ops.temporal_encoder_heads_bf16(d(t(x[:0])), packed, d(cb), hp, d(hb), H).shape == (0, H, T)


def test_bf16_buffer_offset_limits_are_refused_not_wrapped(tspn, device):
    """The operand pieces of the bf16 conv and pair stage are buffer loads with 32-bit offsets: a video whose projections
    reach 2 GB, or packed conv weights of 2 GB, are refused with TSPN_EUNSUPPORTED before any launch (the sizes are only
    declared here: nothing that large is allocated)."""
    lib = tspn._abi.lib()
    small = torch.zeros(4096, dtype=torch.float32, device=device)
    s16 = torch.zeros(4096, dtype=torch.bfloat16, device=device)
    p = lambda z: z.data_ptr()
    # pair stage: N * T * ldm * 4 = 64 * 4200 * 2048 * 4 = 2.2 GB per video
    rc = lib.tspn_heads_pairgrid_bf16(p(small), 2048, 1, 64, 1024, 4200, p(s16), p(small), 12, p(small), 0)
    assert rc == tspn._abi.TSPN_EUNSUPPORTED and "2 GB" in tspn._abi.lib().tspn_last_error().decode()
    # conv: 3 * Cin * M * 2 = 3 * 16384 * 32768 * 2 = 3.2 GB of packed weights
    rc = lib.tspn_conv3_tc_bf16(p(s16), 1, 8, 16384, p(s16), 32768, None, p(small), 32768, 0)
    assert rc == tspn._abi.TSPN_EUNSUPPORTED
    assert 64 * 4000 * 2048 * 4 < 2 ** 31 <= 64 * 4200 * 2048 * 4
    # 2-D conv: one 32768 x 32768 x 64 bf16 image is 128 GB -- every tap past 2 GB would read as padding
    rc = lib.tspn_conv2d_nhwc_bf16(p(s16), 1, 32768, 32768, 64, p(s16), 64, 1, 1, 1, 0, None, None, 0, p(s16), 0)
    assert rc == tspn._abi.TSPN_EUNSUPPORTED and "2 GB" in tspn._abi.lib().tspn_last_error().decode()


def test_conv3_bf16_on_features_beyond_two_gigabytes(tspn, device):
    """x pieces of conv3_bf16_big_kernel are buffer loads based at the tile's first column: on 2.2 GB of features every
    tracklet still gives what it gives alone (first, middle, last)."""
    B, T, Cin, M = 1200, 900, 1024, 256
    assert B * T * Cin * 2 > 2 ** 31
    g = torch.Generator(device=device).manual_seed(6)
    x = torch.empty((B, T, Cin), dtype=torch.bfloat16, device=device)
    for lo in range(0, B, 200):
        x[lo:lo + 200] = (torch.rand((200, T, Cin), device=device, generator=g) - 0.5).to(torch.bfloat16)
    w = (torch.rand((M, Cin, 3), device=device, generator=g) - 0.5) * 0.05
    b = torch.rand(M, device=device, generator=g) - 0.5
    packed = tspn.ops.pack_conv3_bf16(w)
    y = tspn.ops.conv3_tc_bf16(x, packed, b)
    assert tuple(y.shape) == (B, T, M)
    for i in (0, B // 2, B - 1):
        assert torch.equal(y[i:i + 1], tspn.ops.conv3_tc_bf16(x[i:i + 1].contiguous(), packed, b)), i
