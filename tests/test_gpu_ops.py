"""Parity of every C-ABI entry point (called through ops.py -> ctypes -> HIP) against the CPU
oracle and the reference-generated golden vectors.  Tolerances (BASELINE.json north_star):
indices bit-exact; fp32 values within 1e-4 absolute (most checks are far tighter)."""
import numpy as np
import pytest
import torch

import cases
import oracle

pytestmark = pytest.mark.gpu

ATOL = 1e-4  # north_star: fp32 logits within 1e-4
PPN_PRE = "relpn.pair_proposal_network.ppn_head."
DPN_PRE = "relpn.duration_proposal_network.dpn_head."


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def dev_sd(sd, device):
    return {k: t(v).to(device) for k, v in sd.items()}


def test_library_is_the_hip_build(tspn, device):
    import ctypes
    lib = tspn._abi.lib()
    assert isinstance(lib, ctypes.CDLL) and lib.tspn_version() == tspn._abi.ABI_VERSION


# ------------------------------------------------------------------ predicate head
def test_predicate_head_golden_cfg1(tspn, device):
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    feats = t(c["raw"]).to(device)
    tspn.ops.feature_preprocess_(feats)
    np.testing.assert_allclose(feats[:4, :1200].cpu().numpy(), g["preprocessed_rows"], rtol=2e-6, atol=1e-9)
    assert float(feats[3, 70:1070].abs().max()) == 0.0
    sd = dev_sd(c["state_dict"], device)
    out = tspn.ops.predicate_head(feats, sd["classifier.rel_predictor.weight"],
                                  sd["classifier.rel_predictor.bias"])
    assert out.shape == (56, 132)
    np.testing.assert_allclose(out.cpu().numpy(), g["rel_logits"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("P,F,K", [(1, 3, 1), (5, 31, 7), (64, 64, 144), (65, 257, 145), (200, 1001, 132),
                                   (992, 4096, 132)])
def test_predicate_head_shapes(tspn, device, P, F, K):
    x = tspn.hashrng.uniform(21, "x", (P, F), -1.0, 1.0)
    w = tspn.hashrng.normal(21, "w", (K, F), std=0.05)
    b = tspn.hashrng.normal(21, "b", (K,), std=0.1)
    ref = oracle.predicate_head(t(x).double(), t(w).double(), t(b).double()).float().numpy()
    out = tspn.ops.predicate_head(t(x).to(device), t(w).to(device), t(b).to(device))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=5e-6)
    raw = tspn.ops.predicate_head(t(x).to(device), t(w).to(device), None, apply_sigmoid=False)
    ref_raw = (t(x).double() @ t(w).double().t()).float().numpy()
    np.testing.assert_allclose(raw.cpu().numpy(), ref_raw, rtol=0, atol=2e-5)


def test_predicate_head_fused_preprocess_golden_cfg1(tspn, device):
    """f2: raw 11070-d features in, block-L1 normalisation folded into the GEMM == reference
    `_feature_preprocess` + `RelationPredictor` (golden G1), incl. the zero-norm block."""
    g = cases.load("g1_baseline_cfg1.npz")
    c = cases.g1_inputs()
    sd = dev_sd(c["state_dict"], device)
    raw = t(c["raw"]).to(device)
    out = tspn.ops.predicate_head(raw, sd["classifier.rel_predictor.weight"],
                                  sd["classifier.rel_predictor.bias"], norm=(70, 1000, 8))
    np.testing.assert_allclose(out.cpu().numpy(), g["rel_logits"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(raw.cpu().numpy(), c["raw"])  # input untouched


@pytest.mark.parametrize("P,F,K,norm", [(3, 40, 5, (4, 6, 5)), (70, 300, 132, (0, 100, 3)), (200, 1001, 17, (1, 250, 4)),
                                        (992, 11070, 132, (70, 1000, 8)), (9, 64, 4, (0, 64, 0))])
def test_predicate_head_fused_preprocess_vs_oracle(tspn, device, P, F, K, norm):
    x = tspn.hashrng.uniform(25, "x", (P, F), -1.0, 1.0)
    x[min(2, P - 1), norm[0]:norm[0] + norm[1]] = 0
    w = tspn.hashrng.normal(25, "w", (K, F), std=0.05)
    b = tspn.hashrng.normal(25, "b", (K,), std=0.1)
    ref = oracle.predicate_head(oracle.feature_preprocess(t(x).double(), *norm), t(w).double(), t(b).double())
    out = tspn.ops.predicate_head(t(x).to(device), t(w).to(device), t(b).to(device), norm=norm)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), rtol=0, atol=5e-6)


def test_predicate_head_empty_and_errors(tspn, device):
    w = torch.zeros(4, 8, device=device)
    out = tspn.ops.predicate_head(torch.zeros(0, 8, device=device), w, None)
    assert out.shape == (0, 4)
    with pytest.raises(RuntimeError):
        tspn.ops.predicate_head(torch.zeros(2, 8), w, None)  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        tspn.ops.predicate_head(torch.zeros(2, 9, device=device), w, None)


def test_feature_preprocess_matches_oracle(tspn, device):
    x = tspn.hashrng.uniform(22, "pre", (37, 11070), -1.0, 1.0)
    x[5, 2070:3070] = 0
    ref = oracle.feature_preprocess(t(x)).numpy()
    got = tspn.ops.feature_preprocess_(t(x).to(device)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-6, atol=1e-9)
    # idempotence (size-independent property): an L1-normalised block has norm 1
    again = tspn.ops.feature_preprocess_(t(got).to(device)).cpu().numpy()
    np.testing.assert_allclose(again, got, rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(got[:, :70], x[:, :70])
    np.testing.assert_array_equal(got[:, 8070:], x[:, 8070:])


# ------------------------------------------------------------------------------ PPN
def test_ppn_golden(tspn, device):
    g = cases.load("g2_ppn_n32.npz")
    c = cases.g2_inputs(int(g["input_seed"]))
    sd = dev_sd(c["state_dict"], device)
    w = {k[len(PPN_PRE):]: v for k, v in sd.items() if k.startswith(PPN_PRE)}
    mat, idx = tspn.ops.ppn_pair_matrix_topk(t(c["cls"]).to(device), w, 256)
    np.testing.assert_allclose(mat.cpu().numpy(), g["pair_matrix"], rtol=0, atol=2e-6)
    assert idx.dtype == torch.int64
    np.testing.assert_array_equal(idx.cpu().numpy(), g["topk"])  # pair indices bit-exact


@pytest.mark.parametrize("B,N,topk", [(1, 2, 256), (3, 8, 256), (2, 33, 100), (1, 64, 256)])
def test_ppn_batched_and_ties(tspn, device, B, N, topk):
    sd = tspn.synth.make_weights(3, c=8, bias_std=0.1)
    w_np = {k[len(PPN_PRE):]: v for k, v in sd.items() if k.startswith(PPN_PRE)}
    cls = 6.0 * tspn.hashrng.uniform(23, "cls", (B, N, 35))
    cls[:, N // 2] = cls[:, 0]  # duplicate tracklet -> exact ties in the matrix
    w = {k: t(v).to(device) for k, v in w_np.items()}
    mat, idx = tspn.ops.ppn_pair_matrix_topk(t(cls).to(device), w, topk)
    k = min(topk, N * N)
    assert mat.shape == (B, N, N) and idx.shape == (B, k)
    for b in range(B):
        ref = oracle.ppn_pair_matrix(t(cls[b]), {k_: t(v) for k_, v in w_np.items()})
        np.testing.assert_allclose(mat[b].cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)
        # index order must be the stable descending order OF THE DEVICE'S OWN matrix (tie rule)
        mine = oracle.ppn_topk(mat[b].cpu(), k)
        np.testing.assert_array_equal(idx[b].cpu().numpy(), mine.numpy())


def test_ppn_limits(tspn, device):
    sd = tspn.synth.make_weights(3, c=8)
    w = {k[len(PPN_PRE):]: t(v).to(device) for k, v in sd.items() if k.startswith(PPN_PRE)}
    with pytest.raises(tspn._abi.TspnError) as e:
        tspn.ops.ppn_pair_matrix_topk(torch.zeros(1, 129, 35, device=device), w, 10)
    assert e.value.code == tspn._abi.TSPN_EUNSUPPORTED


# ------------------------------------------------------------------------- traj IoU
def test_traj_iou_golden_bit_exact(tspn, device):
    g = cases.load("g4_cubic_iou.npz")
    b, b2 = cases.g4_inputs()
    iou = tspn.ops.traj_iou(t(b).to(device))
    np.testing.assert_array_equal(iou.cpu().numpy(), g["iou"])
    cross = tspn.ops.traj_iou(t(b).to(device), t(b2).to(device))
    np.testing.assert_array_equal(cross.cpu().numpy(), g["iou_cross"])


def test_traj_iou_batched_noninteger(tspn, device):
    xy = tspn.hashrng.uniform(24, "xy", (3, 9, 17, 2), 0, 500)
    wh = tspn.hashrng.uniform(24, "wh", (3, 9, 17, 2), 1, 200)
    boxes = np.concatenate([xy, xy + wh], -1).astype(np.float32)
    got = tspn.ops.traj_iou(t(boxes).to(device)).cpu().numpy()
    for b in range(3):
        np.testing.assert_allclose(got[b], oracle.cubic_iou(boxes[b]), rtol=2e-6, atol=1e-7)


# ----------------------------------------------------------------------- pair builder
@pytest.mark.parametrize("N", [0, 1, 2, 5, 32])
def test_pair_index(tspn, device, N):
    got = tspn.ops.pair_index(N, device, base=7).cpu()
    ref = oracle.pair_index(N) + 7
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.numpy(), ref.numpy())


# (5, 30, 64), (4, 150, 128), (3, 160, 192), (3, 1, 64), (4, 7, 64): the whole-tracklet form of round 6 (D % 64 == 0, T <= 160,
# odd T included); (3, 161, 64) and the others: the 32 x 32 form
@pytest.mark.parametrize("N,T,D", [(2, 1, 1), (3, 7, 5), (5, 30, 64), (4, 33, 70), (6, 150, 96), (4, 150, 128), (3, 160, 192),
                                   (3, 1, 64), (4, 7, 64), (3, 161, 64)])
def test_pair_gather(tspn, device, N, T, D):
    v = tspn.synth.make_video(31, N, T, D)
    pairs = oracle.pair_index(N)
    ref_f, ref_g = oracle.pair_gather(t(v["tracklet_feats"]), t(v["tracklet_boxes"]), pairs)
    f, g = tspn.ops.pair_gather(t(v["tracklet_feats"]).to(device), t(v["tracklet_boxes"]).to(device),
                                pairs.to(device))
    np.testing.assert_array_equal(f.cpu().numpy(), ref_f.numpy())  # pure data movement: bit-exact
    np.testing.assert_allclose(g.cpu().numpy(), ref_g.numpy(), rtol=2e-6, atol=2e-6)
    # arbitrary (filtered / repeated) pair tables, cf. VRDataset._get_proposal_idx
    sub = pairs[torch.tensor([0, len(pairs) - 1, 0])]
    f2, _ = tspn.ops.pair_gather(t(v["tracklet_feats"]).to(device), None, sub.to(device), want_geom=False)
    np.testing.assert_array_equal(f2.cpu().numpy(), ref_f[[0, len(pairs) - 1, 0]].numpy())
    with pytest.raises(IndexError):
        tspn.ops.pair_gather(t(v["tracklet_feats"]).to(device), None,
                             torch.tensor([[0, N]], device=device), want_geom=False)


def test_transpose_mean_rows(tspn, device):
    x = tspn.hashrng.uniform(32, "x", (5, 37, 41), -1, 1)
    xd = t(x).to(device)
    np.testing.assert_array_equal(tspn.ops.transpose_td(xd).cpu().numpy(), x.transpose(0, 2, 1))
    np.testing.assert_allclose(tspn.ops.temporal_mean(xd, True).cpu().numpy(), x.mean(1), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(tspn.ops.temporal_mean(xd, False).cpu().numpy(), x.mean(2), rtol=1e-6, atol=1e-7)
    src = tspn.hashrng.uniform(32, "src", (6, 13))
    pairs = oracle.pair_index(6)
    got = tspn.ops.pair_rows(t(src).to(device), pairs.to(device)).cpu().numpy()
    np.testing.assert_array_equal(got, np.concatenate([src[pairs[:, 0]], src[pairs[:, 1]]], 1))


# ------------------------------------------------------ temporal encoder (conv3 MFMA)
def conv_ref(x, w, b, relu):
    y = torch.nn.functional.conv1d(t(x).double(), t(w).double(), None if b is None else t(b).double(), padding=1)
    return (torch.relu(y) if relu else y).float().numpy()


@pytest.mark.parametrize("B,Cin,T,M", [(1, 1, 1, 1), (2, 3, 5, 7), (3, 16, 30, 128), (5, 20, 33, 130),
                                       (7, 64, 150, 64), (2, 130, 257, 260), (40, 32, 30, 36)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv3_vs_fp64(tspn, device, B, Cin, T, M, relu):
    x = tspn.hashrng.uniform(41, "x", (B, Cin, T), -1, 1)
    w = tspn.hashrng.normal(41, "w", (M, Cin, 3), std=0.1)
    b = tspn.hashrng.normal(41, "b", (M,), std=0.1)
    packed = tspn.ops.pack_conv3(t(w).to(device))
    np.testing.assert_array_equal(packed.cpu().numpy(), w.transpose(2, 1, 0))
    y = tspn.ops.conv3(t(x).to(device), packed, t(b).to(device), relu=relu)
    np.testing.assert_allclose(y.cpu().numpy(), conv_ref(x, w, b, relu), rtol=0, atol=2e-5)
    y0 = tspn.ops.conv3(t(x).to(device), packed, None, relu=relu)
    np.testing.assert_allclose(y0.cpu().numpy(), conv_ref(x, w, None, relu), rtol=0, atol=2e-5)


@pytest.mark.parametrize("B,Cin,T,M", [(1, 16, 1, 4), (3, 16, 30, 128), (5, 32, 33, 132), (7, 64, 150, 64),
                                       (2, 144, 257, 260), (40, 32, 30, 36)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv3_channels_last_vs_fp64(tspn, device, B, Cin, T, M, relu):
    """tspn_conv3_tc_f32: x in the tracklet layout [B,T,Cin]; same operator, same output layout."""
    x = tspn.hashrng.uniform(45, "x", (B, T, Cin), -1, 1)
    w = tspn.hashrng.normal(45, "w", (M, Cin, 3), std=0.1)
    b = tspn.hashrng.normal(45, "b", (M,), std=0.1)
    packed = tspn.ops.pack_conv3(t(w).to(device))
    y = tspn.ops.conv3_tc(t(x).to(device), packed, t(b).to(device), relu=relu)
    ref = conv_ref(np.ascontiguousarray(x.transpose(0, 2, 1)), w, b, relu)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=2e-5)
    # and it agrees with the channels-first kernel on the transposed input
    y2 = tspn.ops.conv3(tspn.ops.transpose_td(t(x).to(device)), packed, t(b).to(device), relu=relu)
    np.testing.assert_allclose(y.cpu().numpy(), y2.cpu().numpy(), rtol=0, atol=1e-5)


def test_conv3_channels_last_rejects_ragged(tspn, device):
    with pytest.raises(tspn._abi.TspnError) as e:
        tspn.ops.conv3_tc(torch.zeros(2, 5, 20, device=device), torch.zeros(3, 20, 8, device=device))
    assert e.value.code == tspn._abi.TSPN_EUNSUPPORTED


def test_conv3_asymmetric_identity(tspn, device):
    """Exact-integer check of the MFMA operand / accumulator lane maps: W = shifted identity,
    asymmetric x (a transposed C-write or a swapped tap would show)."""
    B, C, T = 2, 96, 70
    x = (np.arange(B * C * T, dtype=np.float32).reshape(B, C, T) % 251) - 100.0
    w = np.zeros((C, C, 3), np.float32)
    for m in range(C):
        w[m, (m + 1) % C, 0] = 1.0   # tap -1 of channel m+1
        w[m, m, 1] = 2.0             # centre tap
        w[m, (m + 5) % C, 2] = -3.0  # tap +1 of channel m+5
    y = tspn.ops.conv3(t(x).to(device), tspn.ops.pack_conv3(t(w).to(device)), None).cpu().numpy()
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
    ref = (np.roll(xp[:, :, :-2], -1, 1) + 2 * xp[:, :, 1:-1] - 3 * np.roll(xp[:, :, 2:], -5, 1))
    np.testing.assert_array_equal(y, ref)
    xt = t(np.ascontiguousarray(x.transpose(0, 2, 1))).to(device)
    y_tc = tspn.ops.conv3_tc(xt, tspn.ops.pack_conv3(t(w).to(device)), None).cpu().numpy()
    np.testing.assert_array_equal(y_tc, ref)


def test_conv3_split_pack(tspn, device):
    C, D = 24, 12
    w = tspn.hashrng.normal(42, "w", (C, C, 3), std=0.1)
    packed = tspn.ops.pack_conv3(t(w).to(device), split=D).cpu().numpy()
    assert packed.shape == (3, D, 2 * C)
    np.testing.assert_array_equal(packed[:, :, :C], w[:, :D, :].transpose(2, 1, 0))
    np.testing.assert_array_equal(packed[:, :, C:], w[:, D:, :].transpose(2, 1, 0))


def test_dpn_head_golden(tspn, device):
    """DPNHead.forward of the reference through the dense HIP path: duration against golden G3 (relpn/dpn.py:55-73),
    relationness against golden G11 (relpn/dpn_anchor.py:82-108, the reference's own two-headed DPNHead)."""
    g = cases.load("g3_dpn_head.npz")
    g11 = cases.load("g11_relness_head.npz")
    for tag in cases.G3_SHAPES:
        c = cases.g3_inputs(tag)
        sd = dev_sd(c["state_dict"], device)
        packed = tspn.ops.pack_conv3(sd[DPN_PRE + "conv.weight"])
        hw = torch.cat([sd[DPN_PRE + "relness_pred.weight"][:, :, 0],
                        sd[DPN_PRE + "duration_pred.weight"][:, :, 0]]).contiguous()
        hb = torch.cat([sd[DPN_PRE + "relness_pred.bias"], sd[DPN_PRE + "duration_pred.bias"]]).contiguous()
        out = tspn.ops.temporal_encoder_heads(t(c["x"]).to(device), packed, sd[DPN_PRE + "conv.bias"], hw, hb)
        assert out.shape == (c["x"].shape[0], 12, c["x"].shape[2])
        np.testing.assert_allclose(out[:, 4:].cpu().numpy(), g[f"{tag}_duration"], rtol=0, atol=5e-6)
        cpu = {k: v.cpu() for k, v in sd.items()}
        rel, _, _ = oracle.dpn_head(t(c["x"]), cpu[DPN_PRE + "conv.weight"], cpu[DPN_PRE + "conv.bias"],
                                    cpu[DPN_PRE + "duration_pred.weight"], cpu[DPN_PRE + "duration_pred.bias"],
                                    cpu[DPN_PRE + "relness_pred.weight"], cpu[DPN_PRE + "relness_pred.bias"])
        np.testing.assert_allclose(out[:, :4].cpu().numpy(), rel.numpy(), rtol=0, atol=5e-6)
        np.testing.assert_allclose(out[:, :4].cpu().numpy(), g11[f"{tag}_relness"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("P,C,T,H", [(1, 1, 1, 1), (3, 5, 7, 12), (9, 64, 30, 12), (6, 130, 33, 16), (11, 96, 150, 3)])
def test_heads_dense_and_factorised(tspn, device, P, C, T, H):
    a = tspn.hashrng.uniform(43, "a", (P, C, T), -1, 1)
    b = tspn.hashrng.uniform(43, "b", (P, C, T), -1, 1)
    wh = tspn.hashrng.normal(43, "wh", (H, C), std=0.1)
    bh = tspn.hashrng.normal(43, "bh", (H,), std=0.1)
    bias = tspn.hashrng.normal(43, "bias", (C,), std=0.3)
    ia = tspn.hashrng.integers(43, "ia", (2 * P,), 0, P)
    ib = tspn.hashrng.integers(43, "ib", (2 * P,), 0, P)
    d = lambda v: t(v).to(device)
    ref0 = torch.einsum("hc,pct->pht", t(wh).double(), t(a).double()) + t(bh).double().view(1, -1, 1)
    out0 = tspn.ops.heads(d(a), d(wh), d(bh))
    np.testing.assert_allclose(out0.cpu().numpy(), ref0.float().numpy(), rtol=0, atol=2e-5)
    h = torch.relu(t(a).double()[ia] + t(b).double()[ib] + t(bias).double().view(1, -1, 1))
    ref1 = torch.einsum("hc,pct->pht", t(wh).double(), h) + t(bh).double().view(1, -1, 1)
    out1 = tspn.ops.heads(d(a), d(wh), d(bh), b=d(b), ia=d(ia), ib=d(ib), bias=d(bias))
    assert out1.shape == (2 * P, H, T)
    np.testing.assert_allclose(out1.cpu().numpy(), ref1.float().numpy(), rtol=0, atol=2e-5)


# -------------------------------------------------------------- whole fused pass
def make_w(tspn, seed, D, A=4, K=132, bias_std=0.05):
    sd = tspn.synth.make_weights(seed, c=2 * D, a=A, k=K, bias_std=bias_std)
    return sd, {"conv_w": t(sd[DPN_PRE + "conv.weight"]), "conv_b": t(sd[DPN_PRE + "conv.bias"]),
                "dur_w": t(sd[DPN_PRE + "duration_pred.weight"]), "dur_b": t(sd[DPN_PRE + "duration_pred.bias"]),
                "rel_w": t(sd[DPN_PRE + "relness_pred.weight"]), "rel_b": t(sd[DPN_PRE + "relness_pred.bias"]),
                "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}


def run_fused(tspn, device, feats, pairs, B, N, w, canonical=False):
    D = feats.shape[2]
    d = lambda v: v.to(device).contiguous()
    packed = tspn.ops.pack_conv3(d(w["conv_w"]), split=D)
    hw = d(torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]))
    hb = d(torch.cat([w["rel_b"], w["dur_b"]]))
    return tspn.ops.forward_fused(d(feats), d(pairs), B, N, packed, d(w["conv_b"]), hw, hb,
                                  d(w["cls_w"]), d(w["cls_b"]), canonical_pairs=canonical)


@pytest.mark.parametrize("B,N,C,T,H", [(1, 2, 4, 2, 12), (2, 9, 20, 30, 12), (1, 17, 64, 33, 5), (3, 8, 48, 150, 16)])
def test_heads_pairgrid_matches_generic(tspn, device, B, N, C, T, H):
    """Blocked pair stage (canonical pair table) == generic indexed pair stage, and == fp64."""
    y = tspn.hashrng.uniform(44, "y", (B * N, 2 * C, T), -1, 1)
    wh = tspn.hashrng.normal(44, "wh", (H, C), std=0.1)
    bh = tspn.hashrng.normal(44, "bh", (H,), std=0.1)
    d = lambda v: t(v).to(device)
    pairs = torch.cat([oracle.pair_index(N) + b * N for b in range(B)])
    blocked = tspn.ops.heads_pairgrid(d(y), B, N, d(wh), d(bh)).cpu()
    yd = d(y)
    generic = tspn.ops.heads(yd[:, :C].contiguous(), d(wh), d(bh), b=yd[:, C:].contiguous(),
                             ia=pairs[:, 0].contiguous().to(device), ib=pairs[:, 1].contiguous().to(device)).cpu()
    assert blocked.shape == (B * N * (N - 1), H, T)
    np.testing.assert_allclose(blocked.numpy(), generic.numpy(), rtol=0, atol=1e-5)
    yy = t(y).double()
    h = torch.relu(yy[pairs[:, 0], :C] + yy[pairs[:, 1], C:])
    ref = torch.einsum("hc,pct->pht", t(wh).double(), h) + t(bh).double().view(1, -1, 1)
    np.testing.assert_allclose(blocked.numpy(), ref.float().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("canonical", [False, True])
@pytest.mark.parametrize("B,N,T,D", [(1, 2, 1, 2), (1, 8, 30, 32), (3, 5, 33, 18), (2, 6, 150, 64), (1, 11, 30, 16)])
def test_forward_fused_vs_dense_oracle(tspn, device, B, N, T, D, canonical):
    _, w = make_w(tspn, 0, D)
    vids = [tspn.synth.make_video(50 + b, N, T, D) for b in range(B)]
    feats = torch.cat([t(v["tracklet_feats"]) for v in vids])
    pairs = torch.cat([oracle.pair_index(N) + b * N for b in range(B)])
    heads, logits = run_fused(tspn, device, feats, pairs, B, N, w, canonical)
    A = 4
    for b in range(B):
        ref = oracle.forward_dense(t(vids[b]["tracklet_feats"]), t(vids[b]["tracklet_boxes"]),
                                   oracle.pair_index(N), w)
        sl = slice(b * N * (N - 1), (b + 1) * N * (N - 1))
        np.testing.assert_allclose(heads[sl, :A].cpu().numpy(), ref["relness"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(heads[sl, A:].cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(logits[sl].cpu().numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-5)


def test_fused_equals_dense_hip_path(tspn, device):
    """The two product paths agree: materialising builder + dense encoder vs the fused form."""
    N, T, D = 7, 30, 48
    _, w = make_w(tspn, 1, D)
    v = tspn.synth.make_video(60, N, T, D)
    pairs = oracle.pair_index(N)
    heads_f, _ = run_fused(tspn, device, t(v["tracklet_feats"]), pairs, 1, N, w)
    pf, _ = tspn.ops.pair_gather(t(v["tracklet_feats"]).to(device), t(v["tracklet_boxes"]).to(device),
                                 pairs.to(device))
    d = lambda x: x.to(device).contiguous()
    hw = d(torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]))
    hb = d(torch.cat([w["rel_b"], w["dur_b"]]))
    heads_d = tspn.ops.temporal_encoder_heads(pf, tspn.ops.pack_conv3(d(w["conv_w"])), d(w["conv_b"]), hw, hb)
    np.testing.assert_allclose(heads_f.cpu().numpy(), heads_d.cpu().numpy(), rtol=0, atol=1e-5)


def test_forward_fused_full_size_sampled(tspn, device):
    """BASELINE cfg2 (N=32, T=150, D=2048, C=4096) at full size.  The dense oracle for all 992
    pairs costs ~15 TFLOP on the CPU, so the full-size run is checked on a sample of pairs (the
    oracle scores just those pairs) plus a linearity property of the span heads."""
    N, T, D, B = 32, 150, 2048, 1
    _, w = make_w(tspn, 0, D, bias_std=0.0)
    v = tspn.synth.make_video(1, N, T, D)
    feats = t(v["tracklet_feats"])
    pairs = oracle.pair_index(N)
    heads, logits = run_fused(tspn, device, feats, pairs, B, N, w, canonical=True)
    assert heads.shape == (992, 12, 150) and logits.shape == (992, 132)
    assert bool(torch.isfinite(heads).all()) and bool(torch.isfinite(logits).all())
    sample = torch.tensor([0, 31, 500, 991])
    ref = oracle.forward_dense(feats, t(v["tracklet_boxes"]), pairs[sample], w)
    np.testing.assert_allclose(heads[sample, :4].cpu().numpy(), ref["relness"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(heads[sample, 4:].cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(logits[sample].cpu().numpy(), ref["rel_logits"].numpy(), rtol=0, atol=ATOL)
    err = float(np.abs(heads[sample, 4:].cpu().numpy() - ref["duration"].numpy()).max())
    print(f"cfg2 full-size sampled max |err| (span heads) = {err:.3e}")
    # size-independent property: doubling the head weights and zeroing their bias doubles the output
    w2 = dict(w)
    w2["rel_w"], w2["dur_w"] = 2 * w["rel_w"], 2 * w["dur_w"]
    heads2, _ = run_fused(tspn, device, feats, pairs, B, N, w2, canonical=True)
    np.testing.assert_allclose(heads2.cpu().numpy(), 2 * heads.cpu().numpy(), rtol=0, atol=1e-6)
    # the generic (indexed) pair stage agrees with the blocked one at full size
    heads3, _ = run_fused(tspn, device, feats, pairs, B, N, w, canonical=False)
    np.testing.assert_allclose(heads3.cpu().numpy(), heads.cpu().numpy(), rtol=0, atol=2e-5)


def test_fused_descriptor_errors(tspn, device):
    _, w = make_w(tspn, 0, 8)
    feats = torch.zeros(4, 5, 8)
    pairs = oracle.pair_index(4)
    with pytest.raises(IndexError):
        d = lambda v: v.to(device).contiguous()
        packed = tspn.ops.pack_conv3(d(w["conv_w"]), split=8)
        hw = d(torch.cat([w["rel_w"][:, :, 0], w["dur_w"][:, :, 0]]))
        hb = d(torch.cat([w["rel_b"], w["dur_b"]]))
        tspn.ops.forward_fused(d(feats), d(pairs + 1), 1, 4, packed, d(w["conv_b"]), hw, hb,
                               d(w["cls_w"]), d(w["cls_b"]))
    with pytest.raises(ValueError):
        run_fused(tspn, device, feats, pairs, 2, 4, w)  # B*N mismatch


# ----------------------------------------------------------------- f1: top-k triplet decode
def test_decode_topk_golden(tspn, device):
    """Golden G6 = the reference's predict.py:66-106 restated line by line (tie-free input)."""
    g = cases.load("g6_decode.npz")
    c = cases.g6_inputs()
    sc, trip, tids = tspn.ops.decode_topk(t(c["rel_logit"]).to(device), t(c["pairs"]).to(device),
                                          t(c["feat70"]).to(device), row_mul=c["n"] - 1)
    np.testing.assert_array_equal(sc.cpu().numpy(), g["scores"])
    np.testing.assert_array_equal(trip.cpu().numpy(), g["triplets"])
    np.testing.assert_array_equal(tids.cpu().numpy(), g["pair_tids"])


@pytest.mark.parametrize("S,N,K,kp,ks", [(1, 2, 5, 20, 200), (3, 8, 132, 20, 200), (2, 32, 132, 20, 200),
                                         (1, 12, 70, 7, 33), (1, 40, 132, 20, 1024)])
def test_decode_topk_vs_oracle_with_ties(tspn, device, S, N, K, kp, ks):
    """Quantised scores produce many exact ties at both levels: indices must still be bit-exact
    (lower index first), incl. saturated sigmoids (1.0)."""
    P = N * (N - 1)
    logit = np.round(tspn.hashrng.uniform(61, "dec", (S, P, K)) * 16) / 16
    logit[:, :, 3] = 1.0
    feat = tspn.hashrng.uniform(61, "feat", (S, P, 75))
    feat[:, :, 2] = feat[:, :, 9]  # ties in the class argmax too
    pairs = np.stack([cases.ref_pairs(N)] * S)
    sc, trip, tids = tspn.ops.decode_topk(t(logit).to(device), t(pairs).to(device), t(feat).to(device),
                                          row_mul=N - 1, topk_per_pair=kp, topk_per_seg=ks)
    for s in range(S):
        rs, rt, ri = oracle.decode_topk(t(logit[s]), t(feat[s, :, :70]), t(pairs[s]), N, kp, ks)
        np.testing.assert_array_equal(sc[s].cpu().numpy(), rs.numpy())
        np.testing.assert_array_equal(trip[s].cpu().numpy(), rt.numpy())
        np.testing.assert_array_equal(tids[s].cpu().numpy(), ri.numpy())
    # per-tracklet class logits (row_mul = 1)
    cls = tspn.hashrng.uniform(62, "cls", (S, N, 35))
    _, trip2, tids2 = tspn.ops.decode_topk(t(logit).to(device), t(pairs).to(device), t(cls).to(device),
                                           topk_per_pair=kp, topk_per_seg=ks)
    lab = cls.argmax(-1)
    for s in range(S):
        np.testing.assert_array_equal(trip2[s, :, 0].cpu().numpy(), lab[s][tids2[s, :, 0].cpu().numpy()])
        np.testing.assert_array_equal(trip2[s, :, 2].cpu().numpy(), lab[s][tids2[s, :, 1].cpu().numpy()])


# ------------------------------------------------------- f3: span decode + temporal NMS
@pytest.mark.parametrize("P,A,T,top_k,thr", [(1, 1, 1, 4, 0.5), (5, 4, 30, 64, 0.5), (7, 4, 150, 64, 0.5),
                                             (3, 3, 33, 10, 0.7), (2, 4, 900, 64, 0.5), (4, 2, 150, 1024, 0.3)])
def test_decode_spans_bit_exact(tspn, device, P, A, T, top_k, thr):
    """Span indices (anchor ids, integer frames, counts) bit-exact vs the oracle, incl. tied logits."""
    sizes = [4.0, 8.0, 16.0, 32.0, 64.0][:A]
    heads = tspn.hashrng.normal(71, "spans", (P, 3 * A, T), std=1.0)
    heads[:, :A] = np.round(heads[:, :A] * 8) / 8          # quantised logits -> exact ties
    heads[:, A:] *= 0.4
    if T > 4:
        heads[0, A + 1, 3] = 9.0                            # d_w above the clamp
    hd = t(heads).to(device)
    got = tspn.ops.decode_spans(hd, sizes, top_k=top_k, nms_threshold=thr)
    ref = oracle.decode_spans(t(heads[:, :A]), t(heads[:, A:]), sizes, top_k=top_k, nms_threshold=thr)
    for k in ("count", "anchor", "span"):
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k].numpy(), err_msg=k)
    np.testing.assert_array_equal(got["span_f"].cpu().numpy(), ref["span_f"].numpy())
    np.testing.assert_allclose(got["score"].cpu().numpy(), ref["score"].numpy(), rtol=0, atol=1e-7)
    sp, cnt = got["span"].cpu().numpy(), got["count"].cpu().numpy()
    for p in range(P):                                      # invariants
        v = sp[p, :cnt[p]]
        assert np.all(v[:, 0] >= 0) and np.all(v[:, 1] <= T) and np.all(v[:, 1] > v[:, 0])
        assert np.all(sp[p, cnt[p]:] == -1)


def test_decode_spans_limits(tspn, device):
    with pytest.raises(tspn._abi.TspnError) as e:
        tspn.ops.decode_spans(torch.zeros(1, 12, 2000, device=device), [1.0, 2.0, 3.0, 4.0])
    assert e.value.code == tspn._abi.TSPN_EUNSUPPORTED
    with pytest.raises(ValueError):
        tspn.ops.decode_spans(torch.zeros(1, 11, 20, device=device), [1.0, 2.0, 3.0, 4.0])


# --------------------------------------------- other BASELINE configs as parity cases
def test_cfg3_shape_fp32_sampled(tspn, device):
    """BASELINE cfg3 shape (VidOR long clip: N=64, T=900, D=1024 -> C=2048, P=4032) through the
    fp32 path (the bf16 path is a later round): sampled pairs vs the dense oracle."""
    N, T, D = 64, 900, 1024
    _, w = make_w(tspn, 0, D, bias_std=0.0)
    v = tspn.synth.make_video(3, N, T, D)
    feats = t(v["tracklet_feats"])
    pairs = oracle.pair_index(N)
    heads, logits = run_fused(tspn, device, feats, pairs, 1, N, w, canonical=True)
    assert heads.shape == (4032, 12, 900) and logits.shape == (4032, 132)
    sample = torch.tensor([0, 63, 2017, 4031])
    ref = oracle.forward_dense(feats, t(v["tracklet_boxes"]), pairs[sample], w)
    np.testing.assert_allclose(heads[sample, :4].cpu().numpy(), ref["relness"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(heads[sample, 4:].cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(logits[sample].cpu().numpy(), ref["rel_logits"].numpy(), rtol=0, atol=ATOL)


def test_cfg4_batch_independence(tspn, device):
    """BASELINE cfg4 (many videos sharded / batched): a video's results do not depend on what it is
    batched with — bitwise, since each video's tiles see identical operands in the same order...
    except that tile boundaries move with the batch offset, so compare to a tight tolerance and
    check bitwise equality for the slot-aligned case (same position in the batch)."""
    N, T, D, B = 12, 30, 32, 5
    _, w = make_w(tspn, 2, D)
    vids = [tspn.synth.make_video(200 + b, N, T, D) for b in range(B)]
    feats = torch.cat([t(v["tracklet_feats"]) for v in vids])
    pairs = torch.cat([oracle.pair_index(N) + b * N for b in range(B)])
    hb, lb = run_fused(tspn, device, feats, pairs, B, N, w, canonical=True)
    P = N * (N - 1)
    for b in (0, 3):
        h1, l1 = run_fused(tspn, device, t(vids[b]["tracklet_feats"]), oracle.pair_index(N), 1, N, w, canonical=True)
        np.testing.assert_allclose(hb[b * P:(b + 1) * P].cpu().numpy(), h1.cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(lb[b * P:(b + 1) * P].cpu().numpy(), l1.cpu().numpy(), rtol=0, atol=2e-6)
    h0, l0 = run_fused(tspn, device, t(vids[0]["tracklet_feats"]), oracle.pair_index(N), 1, N, w, canonical=True)
    np.testing.assert_array_equal(hb[:P].cpu().numpy(), h0.cpu().numpy())  # video 0: same tiles -> bitwise
    # determinism: two runs of the same launch are bitwise identical (no atomics anywhere)
    hb2, lb2 = run_fused(tspn, device, feats, pairs, B, N, w, canonical=True)
    assert torch.equal(hb, hb2) and torch.equal(lb, lb2)


@pytest.mark.parametrize("R,T,D", [(5, 150, 64), (3, 7, 8), (2, 31, 2048), (4, 10, 4), (1, 900, 128)])
def test_temporal_mean_vector_form_bit_identical_to_scalar(tspn, device, R, T, D):
    """Round 4: x [R,T,D] with D % 4 == 0 on 16-byte aligned rows takes the four-channels-per-thread kernel (ten frames
    in flight); a copy of the same values at a 4-byte offset takes the scalar kernel.  Same frame order per channel:
    the two results are equal bit for bit, and equal the float64 mean within fp32 rounding."""
    x = tspn.hashrng.uniform(61, "tm", (R, T, D), -1, 3)
    xa = t(x).to(device)
    buf = torch.zeros(R * T * D + 1, dtype=torch.float32, device=device)
    buf[1:] = xa.flatten()
    xb = buf[1:].view(R, T, D)                       # same values, rows at a 4-byte offset: not 16-byte aligned
    assert xa.data_ptr() % 16 == 0 and xb.data_ptr() % 16 == 4
    a, b = tspn.ops.temporal_mean(xa, layout_tc=True), tspn.ops.temporal_mean(xb, layout_tc=True)
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), x.astype(np.float64).mean(axis=1), rtol=0, atol=2e-6)
