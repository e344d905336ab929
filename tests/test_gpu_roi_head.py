"""RoI feature head (first slice) on the GPU against its CPU restatement
(oracle/roi_head_oracle.py: detectron2's published ROIAlign / BottleneckBlock / FrozenBN algorithms; the
reference itself has no runnable counterpart — parity unpinned by the reference)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import roi_head_oracle as ro

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize("NB,H,W,Cin,Cout,k,stride,pad", [
    (2, 7, 7, 16, 32, 1, 1, 0), (3, 14, 14, 32, 64, 1, 2, 0), (2, 7, 7, 32, 48, 3, 1, 1),
    (1, 9, 11, 16, 132, 3, 2, 1), (5, 7, 7, 64, 256, 3, 1, 1), (40, 7, 7, 48, 96, 1, 1, 0),
    (1, 1, 1, 16, 4, 1, 1, 0), (2, 5, 6, 16, 20, 3, 1, 0)])
@pytest.mark.parametrize("fused", [False, True])
def test_conv2d_nhwc_vs_torch(tspn, device, NB, H, W, Cin, Cout, k, stride, pad, fused):
    """Implicit-GEMM conv2d on channels-last tensors == F.conv2d (float64) for 1x1 / 3x3, stride 1 / 2, with
    and without padding, partial weight and pixel tiles; fused bias + residual + ReLU epilogue."""
    x = tspn.hashrng.uniform(71, "x", (NB, H, W, Cin), -1, 1)
    w = tspn.hashrng.normal(71, "w", (Cout, Cin, k, k), std=0.1)
    b = tspn.hashrng.normal(71, "b", (Cout,), std=0.1)
    ref = torch.nn.functional.conv2d(t(x).double().permute(0, 3, 1, 2), t(w).double(), t(b).double() if fused else None,
                                     stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = tspn.hashrng.uniform(71, "r", tuple(ref.shape), -1, 1)
    if fused:
        ref = torch.relu(ref + t(res).double())
    packed = tspn.ops.pack_conv2d(t(w).to(device))
    np.testing.assert_array_equal(packed.cpu().numpy(), w.transpose(2, 3, 1, 0).reshape(k * k, Cin, Cout))
    y = tspn.ops.conv2d_nhwc(t(x).to(device), packed, (k, k), stride, pad,
                             bias=t(b).to(device) if fused else None,
                             residual=t(res).to(device) if fused else None, relu=fused)
    assert tuple(y.shape) == tuple(ref.shape)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    if Cout % 32 == 0:   # registers-direct kernel on fragment-major weights: same contraction order, same bits
        fr = tspn.ops.pack_conv2d_frag(t(w).to(device))
        want = w.reshape(Cout // 32, 32, Cin // 16, 4, 2, 2, k * k).transpose(0, 6, 2, 4, 1, 3, 5)
        np.testing.assert_array_equal(fr.cpu().numpy().reshape(Cout // 32, k * k, Cin // 16, 2, 32, 4, 2), want)
        y2 = tspn.ops.conv2d_nhwc(t(x).to(device), fr, (k, k), stride, pad,
                                  bias=t(b).to(device) if fused else None,
                                  residual=t(res).to(device) if fused else None, relu=fused)
        assert torch.equal(y, y2)


def test_conv2d_nhwc_errors(tspn, device):
    x = torch.zeros((1, 4, 4, 8), device=device)
    with pytest.raises(tspn._abi.TspnError):          # Cin % 16 != 0
        tspn.ops.conv2d_nhwc(x, torch.zeros((1, 8, 16), device=device), (1, 1))
    with pytest.raises(ValueError):                   # packed weights for another Cin
        tspn.ops.conv2d_nhwc(torch.zeros((1, 4, 4, 16), device=device), torch.zeros((1, 32, 16), device=device), (1, 1))
    with pytest.raises(RuntimeError):                 # CPU tensor: no fallback
        tspn.ops.conv2d_nhwc(torch.zeros((1, 4, 4, 16)), torch.zeros((1, 16, 16)), (1, 1))
    y = tspn.ops.conv2d_nhwc(torch.zeros((0, 4, 4, 16), device=device), torch.zeros((1, 16, 16), device=device), (1, 1))
    assert y.shape == (0, 4, 4, 16)
    with pytest.raises(ValueError):                   # fragment-major layout needs Cout % 32 == 0
        tspn.ops.pack_conv2d_frag(torch.zeros((48, 16, 1, 1), device=device))


@pytest.mark.parametrize("sampling_ratio,aligned", [(0, True), (2, True), (0, False), (3, False)])
def test_roi_align_nhwc_vs_oracle(tspn, device, sampling_ratio, aligned):
    """ROIAlign == the restated detectron2 CPU kernel: boxes inside, touching and beyond the map, tiny and
    large boxes (adaptive grid 1..4 samples), several maps."""
    NF, H, W, C, P = 3, 9, 12, 8, 5
    feat = tspn.hashrng.uniform(72, "f", (NF, H, W, C), -1, 1)
    rois = np.array([[0, 10, 20, 100, 120], [1, 0, 0, 191, 143], [2, -30, -10, 60, 50], [0, 150, 100, 260, 200],
                     [1, 40.5, 33.25, 41.0, 34.0], [2, 5, 5, 180, 20], [1, 300, 300, 400, 400]], dtype=np.float32)
    ref = ro.roi_align_nhwc(feat, rois, P, 1.0 / 16, sampling_ratio, aligned)
    got = tspn.ops.roi_align_nhwc(t(feat).to(device), t(rois).to(device), P, 1.0 / 16, sampling_ratio, aligned)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)
    if aligned and sampling_ratio == 0:   # bf16 map read directly == the fp32 kernel on the same (exact) values, rounded
        f16 = t(feat).to(torch.bfloat16)
        a = tspn.ops.roi_align_nhwc(f16.to(device), t(rois).to(device), P, 1.0 / 16, sampling_ratio, aligned)
        b = tspn.ops.roi_align_nhwc(f16.float().to(device), t(rois).to(device), P, 1.0 / 16, sampling_ratio, aligned,
                                    out_bf16=True)
        assert a.dtype == torch.bfloat16 and torch.equal(a, b)
    const = tspn.ops.roi_align_nhwc(torch.full((1, H, W, C), 3.0, device=device), t(rois[:2] * [0, 1, 1, 1, 1]).float().to(device),
                                    P, 1.0 / 16, sampling_ratio, aligned)
    np.testing.assert_allclose(const.cpu().numpy(), 3.0, rtol=1e-6)   # a constant map pools to the constant


def _head_and_weights(tspn, device, cin, mid, cout, **kw):
    u = lambda name, shape, lo, hi: tspn.hashrng.uniform(73, name, shape, lo, hi)
    nrm = lambda name, shape, std: tspn.hashrng.normal(73, name, shape, std=std)
    p = ro.make_res5_weights(u, nrm, cin, mid, cout)
    head = tspn.Res5RoIHead(cin, mid, cout, **kw)
    missing, unexpected = head.load_state_dict(p, strict=True)
    assert not missing and not unexpected            # detectron2 key names: res5.{b}.{conv}.weight / .norm.*
    return head.to(device), p


def test_res5_roi_head_matches_oracle(tspn, device):
    """Tracklet boxes + res4 maps -> [N,T,D] features == ROIAlign + 3 bottleneck blocks (FrozenBN unfused,
    float64 convs) + spatial mean; then the features go straight into the fused scorer."""
    N, T, cin, mid, cout = 3, 4, 64, 32, 128
    head, p = _head_and_weights(tspn, device, cin, mid, cout, roi_chunk=5)      # 12 RoIs in chunks of 5
    fm = tspn.hashrng.uniform(74, "fm", (T, 10, 12, cin), 0, 1)
    v = tspn.synth.make_video(75, N, T, 16)
    boxes = (v["tracklet_boxes"] * np.float32(0.15)).astype(np.float32)          # inside a 192 x 160 image
    got = head(t(fm), t(boxes))                                                   # CPU in -> CPU out
    assert got.device.type == "cpu" and tuple(got.shape) == (N, T, cout)
    ref = ro.res5_roi_head(t(fm), t(boxes), p, dtype=torch.float64)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5 * max(scale, 1.0))
    # NCHW maps (what a detectron2 backbone returns) through the helper
    got2 = head(tspn.Res5RoIHead.from_nchw(t(fm).permute(0, 3, 1, 2).contiguous()).to(device), t(boxes).to(device))
    assert got2.is_cuda and torch.equal(got2.cpu(), got)
    # hand-over to the pair builder: the features are the `tracklet_feats` of PairList.from_tracklets
    plist = tspn.PairList.from_tracklets(got2, t(boxes).to(device), t(v["track_cls_logits"]).to(device))
    assert plist.get_field("tracklet_feats").shape == (N, T, cout)


def test_res5_roi_head_full_width(tspn, device):
    """The real widths (1024 -> 512 -> 2048, 14x14 -> 7x7) on a handful of boxes."""
    N, T = 2, 2
    head, p = _head_and_weights(tspn, device, 1024, 512, 2048)
    fm = tspn.hashrng.uniform(76, "fm", (T, 8, 10, 1024), 0, 1)
    boxes = np.array([[[8, 8, 100, 90], [20, 10, 140, 120]], [[0, 0, 159, 127], [60, 40, 90, 70]]], dtype=np.float32)
    got = head(t(fm).to(device), t(boxes).to(device)).cpu()
    ref = ro.res5_roi_head(t(fm), t(boxes), p, dtype=torch.float64)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=5e-5 * max(scale, 1.0))


def test_res5_roi_head_errors(tspn, device):
    head = tspn.Res5RoIHead(32, 16, 64).to(device)
    with pytest.raises(ValueError):
        head(torch.zeros((2, 5, 5, 16), device=device), torch.zeros((1, 2, 4), device=device))   # wrong C
    with pytest.raises(ValueError):
        head(torch.zeros((2, 5, 5, 32), device=device), torch.zeros((1, 3, 4), device=device))   # T mismatch


@pytest.mark.parametrize("NB,H,W,Cin,Cout,k,stride,pad", [
    (2, 7, 7, 64, 32, 1, 1, 0), (3, 14, 14, 128, 64, 1, 2, 0), (2, 7, 7, 64, 96, 3, 1, 1),
    (1, 9, 11, 64, 160, 3, 2, 1), (5, 7, 7, 192, 256, 3, 1, 1), (40, 7, 7, 64, 128, 1, 1, 0), (1, 1, 1, 64, 32, 1, 1, 0)])
@pytest.mark.parametrize("fused", [False, True])
def test_conv2d_nhwc_bf16_vs_fp64(tspn, device, NB, H, W, Cin, Cout, k, stride, pad, fused):
    """bf16-operand conv2d: exact products, fp32 accumulation (vs float64 on the same bf16 operands), one
    rounding of the result to bf16; the fragment-major bf16 weight layout is the documented permutation."""
    r16 = lambda a: t(a).to(torch.bfloat16)
    x = tspn.hashrng.uniform(77, "x", (NB, H, W, Cin), -1, 1)
    w = tspn.hashrng.normal(77, "w", (Cout, Cin, k, k), std=0.1)
    b = tspn.hashrng.normal(77, "b", (Cout,), std=0.1)
    xb, wb = r16(x), r16(w)
    ref = torch.nn.functional.conv2d(xb.double().permute(0, 3, 1, 2), wb.double(), t(b).double() if fused else None,
                                     stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = r16(tspn.hashrng.uniform(77, "r", tuple(ref.shape), -1, 1))
    if fused:
        ref = torch.relu(ref + res.double())
    frag = tspn.ops.pack_conv2d_frag_bf16(t(w).to(device))
    want = wb.view(torch.int16).numpy().reshape(Cout // 32, 32, Cin // 64, 4, 2, 8, k * k).transpose(0, 2, 6, 3, 4, 1, 5)
    np.testing.assert_array_equal(frag.cpu().view(torch.int16).numpy().reshape(Cout // 32, Cin // 64, k * k, 4, 2, 32, 8), want)
    y = tspn.ops.conv2d_nhwc_bf16(xb.to(device), frag, (k, k), stride, pad, bias=t(b).to(device) if fused else None,
                                  residual=res.to(device) if fused else None, relu=fused)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == tuple(ref.shape)
    got = y.cpu().double()
    # the result is the bf16 rounding of an fp32 sum that is within 2e-5 of the float64 one: half a bf16 ulp
    # of the value plus that slack
    tol = ref.abs() * 2.0 ** -8 + 3e-5
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())
    # and on the vast majority of elements it IS the correctly rounded float64 result
    exact = (got == ref.float().to(torch.bfloat16).double()).double().mean()
    assert float(exact) > 0.99


def test_res5_roi_head_bf16_matches_oracle(tspn, device):
    """bf16 feature maps select the bf16 kernels; output = bf16 features within a few bf16 ulps of the
    float64 restatement with the same rounding points, and close to the fp32 head."""
    N, T, cin, mid, cout = 3, 4, 128, 64, 256
    head, p = _head_and_weights(tspn, device, cin, mid, cout, roi_chunk=7)
    fm = t(tspn.hashrng.uniform(78, "fm", (T, 10, 12, cin), 0, 1)).to(torch.bfloat16)
    v = tspn.synth.make_video(79, N, T, 16)
    boxes = (v["tracklet_boxes"] * np.float32(0.15)).astype(np.float32)
    got = head(fm.to(device), t(boxes).to(device))
    assert got.dtype == torch.bfloat16 and got.is_cuda and tuple(got.shape) == (N, T, cout)
    ref = ro.res5_roi_head_bf16(fm, t(boxes), p)
    scale = float(ref.abs().max())
    err = (got.cpu().float() - ref).abs()
    assert float(err.max()) <= 4 * 2.0 ** -8 * scale, (float(err.max()), scale)   # a few bf16 ulps of the range
    assert float((err <= 2.0 ** -8 * ref.abs() + 1e-6).double().mean()) > 0.97     # mostly within one ulp
    f32 = head(fm.float().to(device), t(boxes).to(device)).cpu()
    assert float((got.cpu().float() - f32).abs().max()) <= 0.05 * float(f32.abs().max())
    # the bf16 features feed the bf16 scorer path unchanged
    plist = tspn.PairList.from_tracklets(got, t(boxes).to(device), t(v["track_cls_logits"]).to(device))
    assert plist.get_field("tracklet_feats").dtype == torch.bfloat16


def test_end_to_end_maps_to_relations(tspn, device):
    """res4 maps + tracklet boxes -> RoI head -> pair builder / temporal encoder / heads -> triplet decode, all on
    the GPU, against the oracle chain (RoI-head restatement -> oracle.forward_dense -> oracle.decode_topk)."""
    import cases
    N, T, cin, mid, D = 4, 6, 64, 32, 32
    head, p = _head_and_weights(tspn, device, cin, mid, D)
    cfg = cases.baseline_cfg(**{"RELPN.USE_PPN": False, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                                "PREDICT.FEATURE_DIM": 2 * D})
    sd = tspn.synth.make_weights(5, c=2 * D, bias_std=0.05)
    model = tspn.BaseModel(cfg)
    own = model.state_dict()
    model.load_state_dict({k: t(v) for k, v in sd.items() if k in own})
    model.eval()
    fm = tspn.hashrng.uniform(81, "fm", (T, 10, 12, cin), 0, 1)
    v = tspn.synth.make_video(82, N, T, 16)
    boxes = (v["tracklet_boxes"] * np.float32(0.15)).astype(np.float32)
    # --- GPU pipeline
    feats = head(t(fm).to(device), t(boxes).to(device))
    plist = tspn.PairList.from_tracklets(feats, t(boxes).to(device), t(v["track_cls_logits"]).to(device))
    with torch.no_grad():
        _, dur, logits = model([plist], None)
    trip = model.decode([plist], logits, topk_per_pair=5, topk_per_seg=20)[0]
    # --- oracle chain
    pre = "relpn.duration_proposal_network.dpn_head."
    w = {"conv_w": t(sd[pre + "conv.weight"]), "conv_b": t(sd[pre + "conv.bias"]),
         "dur_w": t(sd[pre + "duration_pred.weight"]), "dur_b": t(sd[pre + "duration_pred.bias"]),
         "rel_w": t(sd[pre + "relness_pred.weight"]), "rel_b": t(sd[pre + "relness_pred.bias"]),
         "cls_w": t(sd["classifier.rel_predictor.weight"]), "cls_b": t(sd["classifier.rel_predictor.bias"])}
    ofeats = ro.res5_roi_head(t(fm), t(boxes), p, dtype=torch.float64)
    ref = oracle.forward_dense(ofeats, t(boxes), oracle.pair_index(N), w)
    scale = max(1.0, float(ofeats.abs().max()))
    np.testing.assert_allclose(feats.cpu().numpy(), ofeats.numpy(), rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(dur[0].duration.cpu().numpy(), ref["duration"].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(dur[0].relness.cpu().numpy(), ref["relness"].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(logits[0].cpu().numpy(), ref["rel_logits"].numpy(), rtol=0, atol=1e-4)
    # decode of the GPU logits == predict.py:66-106 restated on the same logits (tracklet segments take the
    # classes from track_cls_logits): indices bit-exact
    lg = logits[0].cpu()
    sc, idx = torch.sort(lg, descending=True, dim=-1, stable=True)
    sc, idx = sc[:, :5], idx[:, :5]
    order = torch.sort(sc.flatten(), descending=True, stable=True)[1][:20]
    pi, ki = order // 5, order % 5
    tids = oracle.pair_index(N)[pi]
    cls = t(v["track_cls_logits"])
    want = torch.stack([cls[tids[:, 0]].argmax(1), idx[pi, ki], cls[tids[:, 1]].argmax(1)]).t()
    assert torch.equal(trip[1].cpu(), want) and torch.equal(trip[2].cpu(), tids)
    assert torch.equal(trip[0].cpu(), sc[pi, ki])


def test_max_pool_nhwc(tspn, device):
    x = tspn.hashrng.uniform(83, "x", (2, 9, 12, 8), -1, 1)
    ref = torch.nn.functional.max_pool2d(t(x).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = tspn.ops.max_pool_nhwc(t(x).to(device), 3, 2, 1)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
    got16 = tspn.ops.max_pool_nhwc(t(x).to(device), 3, 2, 1, out_bf16=True)
    assert got16.dtype == torch.bfloat16 and torch.equal(got16.cpu(), ref.to(torch.bfloat16))


def _backbone_and_weights(tspn, device, stem_out, res2_out, blocks):
    u = lambda name, shape, lo, hi: tspn.hashrng.uniform(84, name, shape, lo, hi)
    nrm = lambda name, shape, std: tspn.hashrng.normal(84, name, shape, std=std)
    p = ro.make_backbone_weights(u, nrm, stem_out, res2_out, blocks)
    net = tspn.ResNetC4(stem_out=stem_out, res2_out=res2_out, blocks=blocks, frame_chunk=2)
    missing, unexpected = net.load_state_dict(p, strict=True)     # detectron2 key names
    assert not missing and not unexpected
    return net.to(device), p


def test_resnet_c4_backbone_matches_oracle(tspn, device):
    """Frames -> stem (7x7/2 on zero-padded RGB, FrozenBN, ReLU, max pool) -> res2 -> res3 -> res4 == the float64
    restatement of detectron2's ResNet; odd image sizes, three frames in chunks of two."""
    blocks = (2, 2, 3)
    net, p = _backbone_and_weights(tspn, device, 16, 64, blocks)
    img = tspn.hashrng.uniform(85, "img", (3, 70, 100, 3), -1, 1)
    got = net(t(img).to(device))
    ref = ro.resnet_c4(t(img), p, blocks)
    assert tuple(got.shape) == tuple(ref.shape) == (3, 5, 7, 256)
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-5 * scale)


def test_resnet_c4_backbone_bf16_and_pixels_to_features(tspn, device):
    """bf16=True: stem in fp32, res2-res4 on the bf16 MFMA kernels, within a few bf16 ulps of the restatement
    with the same rounding points; its maps feed the RoI head (frames -> per-tracklet RoI features)."""
    blocks = (1, 1, 2)
    net, p = _backbone_and_weights(tspn, device, 64, 256, blocks)
    img = tspn.hashrng.uniform(86, "img", (2, 64, 96, 3), -1, 1)
    got = net(t(img).to(device), bf16=True)
    ref = ro.resnet_c4_bf16(t(img), p, blocks)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == tuple(ref.shape) == (2, 4, 6, 1024)
    scale = float(ref.abs().max())
    err = (got.cpu().float() - ref).abs()
    assert float(err.max()) <= 6 * 2.0 ** -8 * scale, (float(err.max()), scale)
    assert float((err <= 2.0 ** -7 * ref.abs() + 1e-3 * scale).double().mean()) > 0.97
    head = tspn.Res5RoIHead(1024, 64, 128).to(device)
    boxes = torch.tensor([[[4.0, 4, 60, 50], [10, 8, 90, 60]], [[0, 0, 95, 63], [30, 20, 50, 40]]], device=device)
    feats = head(got, boxes)
    assert feats.dtype == torch.bfloat16 and tuple(feats.shape) == (2, 2, 128) and bool(torch.isfinite(feats.float()).all())


@pytest.mark.parametrize("NB,H,W,Cin,Cout,k,stride,pad", [(2, 30, 41, 3, 64, 7, 2, 3), (1, 9, 9, 3, 32, 3, 1, 1),
                                                           (3, 16, 20, 4, 96, 5, 2, 2), (1, 7, 7, 1, 32, 7, 1, 3)])
def test_conv2d_stem_form_vs_torch(tspn, device, NB, H, W, Cin, Cout, k, stride, pad):
    """Stem form (Cin <= 4: one K chunk = four taps x 4 channels) == F.conv2d in float64, including tap counts
    that are not a multiple of 4 (49, 9, 25) and fewer than 4 input channels."""
    x = tspn.hashrng.uniform(87, "x", (NB, H, W, Cin), -1, 1)
    w = tspn.hashrng.normal(87, "w", (Cout, Cin, k, k), std=0.1)
    b = tspn.hashrng.normal(87, "b", (Cout,), std=0.1)
    ref = torch.relu(torch.nn.functional.conv2d(t(x).double().permute(0, 3, 1, 2), t(w).double(), t(b).double(),
                                                stride=stride, padding=pad)).permute(0, 2, 3, 1)
    frag = tspn.ops.pack_conv2d_frag_cin4(t(w).to(device))
    x4 = torch.nn.functional.pad(t(x), (0, 4 - Cin)).contiguous().to(device)
    y = tspn.ops.conv2d_nhwc_cin4(x4, frag, (k, k), stride, pad, bias=t(b).to(device), relu=True)
    assert tuple(y.shape) == tuple(ref.shape)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("NB,H,W,Cout", [(2, 30, 41, 64), (1, 7, 7, 32), (3, 64, 300, 64), (1, 33, 515, 32), (2, 16, 258, 64)])
def test_stem_bf16_vs_oracle(tspn, device, NB, H, W, Cout):
    """bf16-operand stem conv (space-to-depth + persistent 4x4 conv kernel, tspn_stem_bf16.hip) == conv2d_bf16 of
    the restatement (bf16 image and weights, exact products, one rounding), odd sizes, several 128-pixel tiles per
    row with a ragged last one, 32 and 64 output channels; then the bf16 max pool, bit-exact."""
    x = tspn.hashrng.uniform(88, "x", (NB, H, W, 3), -2, 2)
    w = tspn.hashrng.normal(88, "w", (Cout, 3, 7, 7), std=0.1)
    b = tspn.hashrng.normal(88, "b", (Cout,), std=0.1)
    ref = ro.conv2d_bf16(t(x).permute(0, 3, 1, 2), t(w), t(b), stride=2, padding=3, relu=True).permute(0, 2, 3, 1)
    frag = tspn.ops.pack_stem_bf16(t(w).to(device))
    y = tspn.ops.stem_conv_bf16(t(x).to(device), frag, t(b).to(device))
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == tuple(ref.shape)
    err = (y.cpu().double() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) <= 2.0 ** -8 * scale and float((err > 0).double().mean()) < 0.01   # rare one-ulp flips
    pooled = tspn.ops.max_pool_nhwc_bf16(y, 3, 2, 1)
    want = torch.nn.functional.max_pool2d(y.cpu().float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(pooled.cpu().float(), want)
    assert torch.equal(tspn.ops.stem_conv_bf16(t(x).to(device), frag, t(b).to(device)), y)      # deterministic


@pytest.mark.parametrize("NB,H,W,Cout", [(2, 30, 41, 64), (1, 7, 7, 32), (3, 64, 300, 64), (1, 33, 515, 32), (2, 16, 258, 64),
                                         (1, 1, 1, 64), (1, 2, 513, 64), (5, 440, 70, 64), (1, 720, 1280, 64),
                                         (2, 257, 1027, 32)])
def test_stem_pool_fused_bit_identical_to_conv_then_pool(tspn, device, NB, H, W, Cout):
    """tspn_stem_pool_bf16 (the 3x3/2 max pool inside the stem conv kernel: three conv rows per pooled row reduced in
    the accumulator layout, columns through an LDS slab that carries column -1 from tile to tile) == the conv launch
    followed by the pool launch, BIT FOR BIT: odd sizes, one-pixel images, rows of several 128-column tiles with a
    ragged last one, more pooled rows than workgroup slots (several rows per persistent workgroup), 720p."""
    x = t(tspn.hashrng.uniform(89, "x", (NB, H, W, 3), -2, 2)).to(device)
    w = tspn.hashrng.normal(89, "w", (Cout, 3, 7, 7), std=0.1)
    b = t(tspn.hashrng.normal(89, "b", (Cout,), std=0.1)).to(device)
    frag = tspn.ops.pack_stem_bf16(t(w).to(device))
    want = tspn.ops.max_pool_nhwc_bf16(tspn.ops.stem_conv_bf16(x, frag, b), 3, 2, 1)
    got = tspn.ops.stem_pool_bf16(x, frag, b)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == tuple(want.shape)
    assert torch.equal(got, want)
    assert torch.equal(tspn.ops.stem_pool_bf16(x, frag, b), got)       # deterministic


def test_stem_bf16_errors(tspn, device):
    with pytest.raises(ValueError):
        tspn.ops.pack_stem_bf16(torch.zeros((48, 3, 7, 7), device=device))
    with pytest.raises(ValueError):
        tspn.ops.pack_stem_bf16(torch.zeros((64, 3, 3, 3), device=device))
    frag = tspn.ops.pack_stem_bf16(torch.zeros((64, 3, 7, 7), device=device))
    y = tspn.ops.stem_conv_bf16(torch.zeros((0, 8, 8, 3), device=device), frag, torch.zeros(64, device=device))
    assert y.shape == (0, 4, 4, 64)
    with pytest.raises(ValueError):
        tspn.ops.stem_conv_bf16(torch.zeros((1, 8, 8, 4), device=device), frag, torch.zeros(64, device=device))
    assert tspn.ops.stem_pool_bf16(torch.zeros((0, 8, 8, 3), device=device), frag, torch.zeros(64, device=device)).shape == (0, 2, 2, 64)
    with pytest.raises(ValueError):
        tspn.ops.stem_pool_bf16(torch.zeros((1, 8, 8, 4), device=device), frag, torch.zeros(64, device=device))
    net = tspn.ResNetC4(stem_out=16, res2_out=64, blocks=(1, 1, 1)).to(device)
    with pytest.raises(ValueError):          # the bf16 backbone needs a 32- or 64-channel detectron2 stem
        net(torch.zeros((1, 32, 32, 3), device=device), bf16=True)


@pytest.mark.parametrize("CM,NB,H,W", [(64, 2, 9, 13), (128, 1, 16, 16), (256, 3, 7, 11), (64, 1, 1, 1), (256, 1, 45, 80),
                                      (128, 2, 30, 17)])
def test_bottleneck_tail_fused_bit_identical_to_two_convs(tspn, device, CM, NB, H, W):
    """tspn_bottleneck_tail_bf16 (3x3 conv + 1x1 expand + residual + ReLU in one launch, h2 in LDS, 32 contiguous bytes
    per lane in the epilogue) == tspn_conv2d_nhwc_bf16 applied twice, BIT FOR BIT (same contraction order, same rounding
    points), and == the float64 restatement within bf16 rounding: pixel counts that are not a multiple of the 128-pixel
    tile, images smaller than the 3x3 halo, all three channel widths (wave tilings 1x4 / 2x2 / 4x1)."""
    h1 = tspn.hashrng.uniform(90, "h1", (NB, H, W, CM), 0, 1)
    res = tspn.hashrng.uniform(90, "res", (NB, H, W, 4 * CM), -1, 1)
    w2 = tspn.hashrng.normal(90, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(90, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    b2 = tspn.hashrng.normal(90, "b2", (CM,), std=0.1)
    b3 = tspn.hashrng.normal(90, "b3", (4 * CM,), std=0.1)
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(d(w2)), tspn.ops.pack_conv2d_frag_bf16(d(w3))
    h1d, resd = d(h1, torch.bfloat16), d(res, torch.bfloat16)
    h2 = tspn.ops.conv2d_nhwc_bf16(h1d, f2, (3, 3), 1, 1, bias=d(b2), relu=True)
    want = tspn.ops.conv2d_nhwc_bf16(h2, f3, (1, 1), 1, 0, bias=d(b3), residual=resd, relu=True)
    got = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (NB, H, W, 4 * CM)
    assert torch.equal(got, want), f"max diff {float((got.float() - want.float()).abs().max())}"
    big = torch.zeros((NB + 2, H, W, 4 * CM), dtype=torch.bfloat16, device=device)      # `out`: a slice of a larger result
    ret = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, out=big[1:NB + 1])
    assert ret.data_ptr() == big[1:].data_ptr() and torch.equal(big[1:NB + 1], want)
    assert not bool(big[0].any()) and not bool(big[NB + 1].any())
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, out=big)
    r2 = ro.conv2d_bf16(h1d.cpu().float().permute(0, 3, 1, 2), t(w2), t(b2), padding=1, relu=True)
    ref = ro.conv2d_bf16(r2, t(w3), t(b3), residual=resd.cpu().float().permute(0, 3, 1, 2), relu=True).permute(0, 2, 3, 1)
    err = (got.cpu().double() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) <= 4 * 2.0 ** -8 * scale and float((err <= 2.0 ** -8 * ref.abs() + 1e-6).double().mean()) > 0.97


@pytest.mark.parametrize("CM,NB,H,W", [(64, 2, 9, 13), (128, 1, 16, 16), (64, 1, 1, 1), (128, 2, 30, 17), (64, 3, 4, 30),
                                      (64, 1, 5, 31), (128, 1, 3, 61), (64, 1, 45, 80), (128, 2, 23, 40)])
def test_bottleneck_block_one_launch_bit_identical_to_conv1_plus_fused_tail(tspn, device, CM, NB, H, W):
    """tspn_bottleneck_block_bf16 (round 5): a whole identity-shortcut bottleneck block -- conv1, 3x3, expand, residual,
    ReLU -- in ONE launch on 4 x 30 pixel tiles with the halo recomputed, h1 / h2 in LDS, the 4 CM-channel map read once.
    BIT FOR BIT the chain it replaces (tspn_conv2d_nhwc_bf16 for conv1, then tspn_bottleneck_tail_bf16): same contraction
    order and rounding points; images smaller than a tile, exactly a tile, one pixel more than a tile in either direction,
    several images; and == the float64 restatement within bf16 rounding.  Repeated launches agree."""
    x = tspn.hashrng.uniform(95, "x", (NB, H, W, 4 * CM), -1, 1)
    w1 = tspn.hashrng.normal(95, "w1", (CM, 4 * CM, 1, 1), std=float(np.sqrt(2.0 / (4 * CM))))
    w2 = tspn.hashrng.normal(95, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(95, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    b1 = tspn.hashrng.normal(95, "b1", (CM,), std=0.1)
    b2 = tspn.hashrng.normal(95, "b2", (CM,), std=0.1)
    b3 = tspn.hashrng.normal(95, "b3", (4 * CM,), std=0.1)
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f1, f2, f3 = (tspn.ops.pack_conv2d_frag_bf16(d(w)) for w in (w1, w2, w3))
    xd = d(x, torch.bfloat16)
    h1 = tspn.ops.conv2d_nhwc_bf16(xd, f1, (1, 1), 1, 0, bias=d(b1), relu=True)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, d(b2), f3, d(b3), xd)
    got = tspn.ops.bottleneck_block_bf16(xd, f1, d(b1), f2, d(b2), f3, d(b3))
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (NB, H, W, 4 * CM)
    assert torch.equal(got, want), f"max diff {float((got.float() - want.float()).abs().max())}"
    assert float(got.float().abs().max()) > 0.1
    big = torch.zeros((NB + 2, H, W, 4 * CM), dtype=torch.bfloat16, device=device)      # `out`: a slice of a larger result
    ret = tspn.ops.bottleneck_block_bf16(xd, f1, d(b1), f2, d(b2), f3, d(b3), out=big[1:NB + 1])
    assert ret.data_ptr() == big[1:].data_ptr() and torch.equal(big[1:NB + 1], want)
    assert not bool(big[0].any()) and not bool(big[NB + 1].any())
    for _ in range(2):
        assert torch.equal(tspn.ops.bottleneck_block_bf16(xd, f1, d(b1), f2, d(b2), f3, d(b3)), want)
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_block_bf16(xd, f2, d(b1), f2, d(b2), f3, d(b3))
    with pytest.raises(RuntimeError):                                                    # in place: refused by the entry
        tspn.ops.bottleneck_block_bf16(xd, f1, d(b1), f2, d(b2), f3, d(b3), out=xd)
    xf = xd.cpu().float().permute(0, 3, 1, 2)
    r1 = ro.conv2d_bf16(xf, t(w1), t(b1), relu=True)
    r2 = ro.conv2d_bf16(r1, t(w2), t(b2), padding=1, relu=True)
    ref = ro.conv2d_bf16(r2, t(w3), t(b3), residual=xf, relu=True).permute(0, 2, 3, 1)
    err = (got.cpu().double() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) <= 4 * 2.0 ** -8 * scale
    if err.numel() >= 4096:      # three bf16 roundings in a row: most values within one ulp (a statistic: not for tiny maps)
        assert float((err <= 2.0 ** -8 * ref.abs() + 1e-6).double().mean()) > 0.95


@pytest.mark.parametrize("CIN,CM,stride,NB,H,W", [(64, 64, 1, 2, 9, 13), (64, 64, 1, 1, 1, 1), (64, 64, 1, 1, 10, 30),
                                                 (64, 64, 1, 1, 11, 31), (64, 64, 1, 1, 45, 80), (64, 64, 1, 3, 23, 40)])
def test_first_block_of_a_stage_one_launch_bit_identical_to_its_four_launches(tspn, device, CIN, CM, stride, NB, H, W):
    """tspn_bottleneck_block_proj_bf16 (round 5): the first block of a stage -- conv1 and the PROJECTION shortcut (1x1 convs
    of stride s on the same input), 3x3, expand, residual, ReLU -- in one launch; the shortcut is computed on the tile's
    own input pixels and rounded to bf16 like the separate launch's output, so the result equals conv1 + shortcut + fused
    tail BIT FOR BIT.  H, W here are the INPUT sizes."""
    x = tspn.hashrng.uniform(96, "x", (NB, H, W, CIN), -1, 1)
    w1 = tspn.hashrng.normal(96, "w1", (CM, CIN, 1, 1), std=float(np.sqrt(2.0 / CIN)))
    w2 = tspn.hashrng.normal(96, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(96, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    ws = tspn.hashrng.normal(96, "ws", (4 * CM, CIN, 1, 1), std=float(np.sqrt(1.0 / CIN)))
    b1, b2 = (tspn.hashrng.normal(96, n, (CM,), std=0.1) for n in ("b1", "b2"))
    b3, bs = (tspn.hashrng.normal(96, n, (4 * CM,), std=0.1) for n in ("b3", "bs"))
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f1, f2, f3, fs = (tspn.ops.pack_conv2d_frag_bf16(d(w)) for w in (w1, w2, w3, ws))
    xd = d(x, torch.bfloat16)
    h1 = tspn.ops.conv2d_nhwc_bf16(xd, f1, (1, 1), stride, 0, bias=d(b1), relu=True)
    sc = tspn.ops.conv2d_nhwc_bf16(xd, fs, (1, 1), stride, 0, bias=d(bs), relu=False)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, d(b2), f3, d(b3), sc)
    got = tspn.ops.bottleneck_block_proj_bf16(xd, stride, f1, d(b1), f2, d(b2), f3, d(b3), fs, d(bs))
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == tuple(want.shape)
    assert torch.equal(got, want), f"max diff {float((got.float() - want.float()).abs().max())}"
    assert float(got.float().abs().max()) > 0.1
    for _ in range(2):
        assert torch.equal(tspn.ops.bottleneck_block_proj_bf16(xd, stride, f1, d(b1), f2, d(b2), f3, d(b3), fs, d(bs)), want)
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_block_proj_bf16(xd, stride + 1, f1, d(b1), f2, d(b2), f3, d(b3), fs, d(bs))


@pytest.mark.parametrize("NB,H,W", [(2, 9, 13), (1, 1, 1), (1, 12, 60), (1, 13, 61), (1, 90, 160), (2, 47, 81), (3, 180, 320)])
def test_res3_first_block_conv1_3x3_expand_one_launch_bit_identical(tspn, device, NB, H, W):
    """tspn_bottleneck_block_res_bf16 (round 5): res3.0 -- conv1 as a 1x1 of stride 2 on the 256-channel input, 3x3, expand,
    + the separately launched projection shortcut as the residual, ReLU -- equals conv1 + fused tail BIT FOR BIT (H, W are
    the INPUT sizes; odd sizes, single pixels, 720p's res2 map, several images; three launches each into a poisoned output)."""
    CIN, CM, stride = 256, 128, 2
    x = tspn.hashrng.uniform(97, "x", (NB, H, W, CIN), -1, 1)
    w1 = tspn.hashrng.normal(97, "w1", (CM, CIN, 1, 1), std=float(np.sqrt(2.0 / CIN)))
    w2 = tspn.hashrng.normal(97, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(97, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    ws = tspn.hashrng.normal(97, "ws", (4 * CM, CIN, 1, 1), std=float(np.sqrt(1.0 / CIN)))
    b1, b2 = (tspn.hashrng.normal(97, n, (CM,), std=0.1) for n in ("b1", "b2"))
    b3, bs = (tspn.hashrng.normal(97, n, (4 * CM,), std=0.1) for n in ("b3", "bs"))
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f1, f2, f3, fs = (tspn.ops.pack_conv2d_frag_bf16(d(w)) for w in (w1, w2, w3, ws))
    xd = d(x, torch.bfloat16)
    h1 = tspn.ops.conv2d_nhwc_bf16(xd, f1, (1, 1), stride, 0, bias=d(b1), relu=True)
    sc = tspn.ops.conv2d_nhwc_bf16(xd, fs, (1, 1), stride, 0, bias=d(bs), relu=False)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, d(b2), f3, d(b3), sc)
    for trial in range(3):
        out = torch.full_like(want, 777.0)
        ret = tspn.ops.bottleneck_block_res_bf16(xd, stride, f1, d(b1), f2, d(b2), f3, d(b3), sc, out=out)
        assert ret.data_ptr() == out.data_ptr()
        assert torch.equal(out, want), f"trial {trial}: max diff {float((out.float() - want.float()).abs().max())}"
    assert float(want.float().abs().max()) > 0.1
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_block_res_bf16(xd, 1, f1, d(b1), f2, d(b2), f3, d(b3), sc)
    with pytest.raises(RuntimeError):
        tspn.ops.bottleneck_block_res_bf16(xd, stride, f1, d(b1), f2, d(b2), f3, d(b3), sc, out=sc)


@pytest.mark.parametrize("CM,H,W", [(64, 180, 320), (128, 90, 160)])
def test_one_launch_blocks_at_720p_scale_with_every_cu_busy(tspn, device, CM, H, W):
    """The one-launch blocks at the backbone's own sizes (4 frames of 720p at res2 / res3: 800 / 360 tiles, several
    workgroups per CU) into a poisoned output, six launches each: every launch equals conv1 (+ shortcut) + fused tail bit for
    bit.  This is the test that catches TIMING-dependent faults -- the store-data hazard of round 5 showed up only with more
    than one workgroup per CU, in 0.02 % of the outputs, differently in every launch (profiles/r5/bottleneck_block_study.md §3)."""
    NB = 4
    g = torch.Generator(device=device).manual_seed(5)
    x = (torch.rand((NB, H, W, 4 * CM), device=device, generator=g) - 0.5).to(torch.bfloat16)
    w1 = (torch.rand((CM, 4 * CM, 1, 1), device=device, generator=g) - 0.5) * 0.1
    w2 = (torch.rand((CM, CM, 3, 3), device=device, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=device, generator=g) - 0.5) * 0.1
    b1, b2 = torch.rand(CM, device=device, generator=g) - 0.5, torch.rand(CM, device=device, generator=g) - 0.5
    b3 = torch.rand(4 * CM, device=device, generator=g) - 0.5
    f1, f2, f3 = (tspn.ops.pack_conv2d_frag_bf16(w) for w in (w1, w2, w3))
    h1 = tspn.ops.conv2d_nhwc_bf16(x, f1, (1, 1), 1, 0, bias=b1, relu=True)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, x)
    for trial in range(6):
        out = torch.full_like(x, 777.0)
        tspn.ops.bottleneck_block_bf16(x, f1, b1, f2, b2, f3, b3, out=out)
        bad = int((out != want).sum())
        assert bad == 0, f"trial {trial}: {bad} of {out.numel()} outputs differ"
    if CM == 64:                                   # res2.0: the stem's 64-channel map in, projection shortcut inside
        xs = (torch.rand((NB, H, W, 64), device=device, generator=g) - 0.5).to(torch.bfloat16)
        w1s = (torch.rand((64, 64, 1, 1), device=device, generator=g) - 0.5) * 0.2
        wsc = (torch.rand((256, 64, 1, 1), device=device, generator=g) - 0.5) * 0.2
        bsc = torch.rand(256, device=device, generator=g) - 0.5
        f1s, fsc = tspn.ops.pack_conv2d_frag_bf16(w1s), tspn.ops.pack_conv2d_frag_bf16(wsc)
        h1s = tspn.ops.conv2d_nhwc_bf16(xs, f1s, (1, 1), 1, 0, bias=b1, relu=True)
        sc = tspn.ops.conv2d_nhwc_bf16(xs, fsc, (1, 1), 1, 0, bias=bsc, relu=False)
        want_p = tspn.ops.bottleneck_tail_bf16(h1s, f2, b2, f3, b3, sc)
        for trial in range(6):
            out = torch.full_like(want_p, 777.0)
            tspn.ops.bottleneck_block_proj_bf16(xs, 1, f1s, b1, f2, b2, f3, b3, fsc, bsc, out=out)
            bad = int((out != want_p).sum())
            assert bad == 0, f"res2.0 trial {trial}: {bad} of {out.numel()} outputs differ"


def test_backbone_with_one_launch_blocks_equals_the_chain(tspn, device):
    """ResNetC4 on bf16 maps: `fuse_block` on (identity blocks of res2 / res3 as one launch each) and off (conv1 + fused
    tail) give the same res4 maps bit for bit, on one stream and on two; the one-launch form really ran."""
    net, _ = _backbone_and_weights(tspn, device, 64, 256, (3, 4, 2))
    img = t(tspn.hashrng.uniform(99, "img", (5, 96, 128, 3), -1, 1)).to(device)
    net.frame_chunk = 2
    calls = []
    real = tspn.ops.bottleneck_block_bf16
    outs = []
    try:
        def spy(*a, **k):
            calls.append(a[0].shape[3])
            return real(*a, **k)
        tspn.ops.bottleneck_block_bf16 = spy
        tspn.roi_head.ops.bottleneck_block_bf16 = spy
        real_p = tspn.ops.bottleneck_block_proj_bf16
        pcalls = []

        def spy_p(*a, **k):
            pcalls.append(tuple(a[0].shape[1:]))
            return real_p(*a, **k)
        tspn.ops.bottleneck_block_proj_bf16 = spy_p
        tspn.roi_head.ops.bottleneck_block_proj_bf16 = spy_p
        real_r = tspn.ops.bottleneck_block_res_bf16
        rcalls = []

        def spy_r(*a, **k):
            rcalls.append(tuple(a[0].shape[1:]))
            return real_r(*a, **k)
        tspn.ops.bottleneck_block_res_bf16 = spy_r
        tspn.roi_head.ops.bottleneck_block_res_bf16 = spy_r
        for streams in (1, 2):
            for on in (True, False):
                net.streams = streams
                net.fuse_blocks = on
                n0 = len(calls)
                outs.append(net(img, bf16=True))
                torch.cuda.synchronize(device)
                assert (len(calls) - n0) == (3 * (2 + 3) if on else 0)        # 3 frame chunks x (2 res2 + 3 res3 blocks)
    finally:
        tspn.ops.bottleneck_block_bf16 = real
        tspn.roi_head.ops.bottleneck_block_bf16 = real
        tspn.ops.bottleneck_block_proj_bf16 = real_p
        tspn.roi_head.ops.bottleneck_block_proj_bf16 = real_p
        tspn.ops.bottleneck_block_res_bf16 = real_r
        tspn.roi_head.ops.bottleneck_block_res_bf16 = real_r
    assert set(calls) == {256, 512}
    assert len(pcalls) == 2 * 3 and all(c[2] == 64 for c in pcalls)          # res2.0 of every chunk, with fuse_blocks on
    assert len(rcalls) == 2 * 3 and all(c[2] == 256 for c in rcalls)         # res3.0 likewise (shortcut launched separately)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("NB,H,W", [(3, 7, 11), (1, 45, 80), (1, 1, 1), (2, 16, 8), (1, 13, 129)])
def test_bottleneck_tail_with_next_conv1_bit_identical(tspn, device, NB, H, W):
    """tspn_bottleneck_tail_next_bf16 (round 4): the tail launch that also computes conv1 of the FOLLOWING block on the
    tile it has just produced.  Its `out` equals tspn_bottleneck_tail_bf16's and its `h1_next` equals
    tspn_conv2d_nhwc_bf16(out, W1n, relu) BIT FOR BIT (same contraction order: channels 0..1023 in k-steps of 16 on
    one accumulator chain), on pixel counts that are not multiples of the 128-pixel tile; repeated launches agree."""
    CM = 256
    h1 = tspn.hashrng.uniform(93, "h1", (NB, H, W, CM), 0, 1)
    res = tspn.hashrng.uniform(93, "res", (NB, H, W, 4 * CM), -1, 1)
    w2 = tspn.hashrng.normal(93, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(93, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    w1n = tspn.hashrng.normal(93, "w1n", (CM, 4 * CM, 1, 1), std=float(np.sqrt(2.0 / (4 * CM))))
    b2 = tspn.hashrng.normal(93, "b2", (CM,), std=0.1)
    b3 = tspn.hashrng.normal(93, "b3", (4 * CM,), std=0.1)
    b1n = tspn.hashrng.normal(93, "b1n", (CM,), std=0.1)
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f2, f3, f1n = (tspn.ops.pack_conv2d_frag_bf16(d(w)) for w in (w2, w3, w1n))
    h1d, resd = d(h1, torch.bfloat16), d(res, torch.bfloat16)
    want = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd)
    want_h = tspn.ops.conv2d_nhwc_bf16(want, f1n, (1, 1), 1, 0, bias=d(b1n), relu=True)
    got, got_h = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, next_frag1=f1n, next_bias1=d(b1n))
    assert tuple(got_h.shape) == (NB, H, W, CM) and got_h.dtype == torch.bfloat16
    assert torch.equal(got, want), f"out: max diff {float((got.float() - want.float()).abs().max())}"
    assert torch.equal(got_h, want_h), f"h1_next: max diff {float((got_h.float() - want_h.float()).abs().max())}"
    assert float(got_h.float().abs().max()) > 0.1                 # not a vacuous comparison
    for _ in range(3):
        g2, h2 = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, next_frag1=f1n, next_bias1=d(b1n))
        assert torch.equal(g2, got) and torch.equal(h2, got_h)
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, next_frag1=f3, next_bias1=d(b1n))


@pytest.mark.parametrize("NB,H,W,wgs", [(3, 7, 11, 0), (1, 45, 80, 0), (1, 1, 1, 0), (2, 16, 8, 1), (1, 13, 129, 3),
                                         (9, 45, 80, 0), (40, 45, 80, 0), (5, 31, 33, 7)])
def test_bottleneck_tail_persistent_pipelined_bit_identical(tspn, device, NB, H, W, wgs):
    """tspn_bottleneck_tail_pipe_bf16 (round 4): the persistent eight-wave kernel -- phase 2 of tile t on four waves
    beside phase 3 of tile t - 1 on the other four, 14 workgroup barriers per tile on both sides -- equals
    tspn_bottleneck_tail_bf16 BIT FOR BIT: fewer tiles than CUs, one tile, several tiles per workgroup (a grid limited
    to 1 / 3 / 7 workgroups, and 40 frames of the res4 shape = 1125 tiles on 256 CUs), pixel counts that are not a
    multiple of the tile; repeated launches agree."""
    CM = 256
    h1 = tspn.hashrng.uniform(94, "h1", (NB, H, W, CM), 0, 1)
    res = tspn.hashrng.uniform(94, "res", (NB, H, W, 4 * CM), -1, 1)
    w2 = tspn.hashrng.normal(94, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(94, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    b2 = tspn.hashrng.normal(94, "b2", (CM,), std=0.1)
    b3 = tspn.hashrng.normal(94, "b3", (4 * CM,), std=0.1)
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(d(w2)), tspn.ops.pack_conv2d_frag_bf16(d(w3))
    h1d, resd = d(h1, torch.bfloat16), d(res, torch.bfloat16)
    want = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd)
    for rep in range(3):
        got = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, persistent=True, max_workgroups=wgs)
        assert torch.equal(got, want), f"rep {rep}: max diff {float((got.float() - want.float()).abs().max())}"
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_tail_bf16(h1d[..., :128].contiguous(), f2, d(b2), f3, d(b3), resd, persistent=True)


@pytest.mark.parametrize("NB,H,W", [(3, 7, 11), (1, 45, 80), (1, 1, 1), (2, 16, 8), (1, 13, 129), (9, 45, 80), (5, 31, 33),
                                     (1, 8, 16), (1, 2, 65)])
def test_bottleneck_tail_role_split_bit_identical(tspn, device, NB, H, W):
    """tspn_bottleneck_tail_io_bf16 (round 5): four MFMA waves + four io waves per workgroup, the expand sums handed over
    through LDS under two counters per wave pair, the epilogue line-major.  Equals tspn_bottleneck_tail_bf16 BIT FOR BIT
    into a poisoned output: one pixel, one tile exactly (8 x 16), tiles that end inside a 64-pixel half (1 x 2 x 65: the
    second half of a sub-pass has ONE pixel), fewer tiles than CUs, one per CU (9 frames of res4), image widths that are
    no multiple of anything; repeated launches agree."""
    CM = 256
    h1 = tspn.hashrng.uniform(95, "h1", (NB, H, W, CM), 0, 1)
    res = tspn.hashrng.uniform(95, "res", (NB, H, W, 4 * CM), -1, 1)
    w2 = tspn.hashrng.normal(95, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))
    w3 = tspn.hashrng.normal(95, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))
    b2 = tspn.hashrng.normal(95, "b2", (CM,), std=0.1)
    b3 = tspn.hashrng.normal(95, "b3", (4 * CM,), std=0.1)
    d = lambda a, dt=None: (t(a).to(device) if dt is None else t(a).to(device).to(dt))   # noqa: E731
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(d(w2)), tspn.ops.pack_conv2d_frag_bf16(d(w3))
    h1d, resd = d(h1, torch.bfloat16), d(res, torch.bfloat16)
    want = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd)
    assert float(want.float().abs().max()) > 0.1                 # not a vacuous comparison
    for rep in range(3):
        out = torch.full_like(want, 777.0)
        got = tspn.ops.bottleneck_tail_bf16(h1d, f2, d(b2), f3, d(b3), resd, out=out, io_waves=True)
        assert got.data_ptr() == out.data_ptr()
        bad = int((got != want).sum())
        assert bad == 0, f"rep {rep}: {bad} of {got.numel()} outputs differ, max diff {float((got.float() - want.float()).abs().max())}"
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_tail_bf16(h1d[..., :128].contiguous(), f2, d(b2), f3, d(b3), resd, io_waves=True)


def test_bottleneck_tail_role_split_at_backbone_scale(tspn, device):
    """The role-split tail on 40 frames of the res4 shape (1125 tiles: four to five rounds of one workgroup per CU) and on
    two HIP streams at once, eight launches into poisoned outputs: every one equals the one-role kernel bit for bit.  (Timing-
    dependent faults -- a counter hand-over that loses a sub-pass, the store-data hazard -- show up at this scale, not on
    a handful of tiles.)"""
    CM, NB, H, W = 256, 40, 45, 80
    g = torch.Generator(device=device).manual_seed(6)
    h1 = torch.rand((NB, H, W, CM), device=device, generator=g).to(torch.bfloat16)
    res = (torch.rand((NB, H, W, 4 * CM), device=device, generator=g) - 0.5).to(torch.bfloat16)
    w2 = (torch.rand((CM, CM, 3, 3), device=device, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=device, generator=g) - 0.5) * 0.1
    b2, b3 = torch.rand(CM, device=device, generator=g) - 0.5, torch.rand(4 * CM, device=device, generator=g) - 0.5
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=device)
    outs = [torch.full_like(want, 777.0) for _ in range(2)]
    for trial in range(4):
        for o in outs:
            o.fill_(777.0)
        side.wait_stream(torch.cuda.current_stream(device))
        tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=outs[0], io_waves=True)
        with torch.cuda.stream(side):
            tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=outs[1], io_waves=True)
        torch.cuda.current_stream(device).wait_stream(side)
        for k, o in enumerate(outs):
            bad = int((o != want).sum())
            assert bad == 0, f"trial {trial}, stream {k}: {bad} of {o.numel()} outputs differ"


def test_backbone_with_role_split_tails_equals_the_one_role_kernel(tspn, device):
    """ResNetC4 on bf16 maps: `tail_io_waves` on (res4's fused tails on tspn_bottleneck_tail_io_bf16) and off give the same
    res4 maps bit for bit, on one stream and on two, with frame chunks whose pixel count is no multiple of the tile; the
    role-split kernel really ran -- for the 256-channel tails only."""
    net, _ = _backbone_and_weights(tspn, device, 64, 256, (3, 4, 3))
    img = t(tspn.hashrng.uniform(98, "img", (5, 96, 144, 3), -1, 1)).to(device)
    net.frame_chunk = 2
    calls = []
    real = tspn.ops.bottleneck_tail_bf16
    outs = []
    try:
        def spy(*a, **k):
            calls.append((a[0].shape[3], bool(k.get("io_waves", False))))
            return real(*a, **k)
        tspn.ops.bottleneck_tail_bf16 = spy
        tspn.roi_head.ops.bottleneck_tail_bf16 = spy
        for streams in (1, 2):
            for on in (True, False):
                net.streams = streams
                net.tail_io_waves = on
                n0 = len(calls)
                outs.append(net(img, bf16=True))
                torch.cuda.synchronize(device)
                mine = calls[n0:]
                assert sum(1 for cm, io in mine if cm == 256) == 3 * 3          # 3 frame chunks x 3 res4 blocks
                assert all(io == (on and cm == 256) for cm, io in mine)
    finally:
        tspn.ops.bottleneck_tail_bf16 = real
        tspn.roi_head.ops.bottleneck_tail_bf16 = real
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_backbone_chain_with_next_conv1_equals_separate_launches(tspn, device):
    """ResNetC4 with res4 blocks of 256 bottleneck channels: `fuse_next_conv1` (every tail launch computes the follower's
    conv1) on and off give the same maps bit for bit, on one stream and on two, with frame chunks whose pixel count
    is not a multiple of the tile."""
    net, _ = _backbone_and_weights(tspn, device, 64, 256, (1, 1, 4))
    assert net.res4[1].conv1.weight.shape[:2] == (256, 1024)
    img = t(tspn.hashrng.uniform(98, "img", (5, 80, 112, 3), -1, 1)).to(device)
    net.frame_chunk = 2
    net.fuse_first_blocks = False            # res2.0 through its fused tail here (the launch counts below)
    outs = []
    calls = []
    real = tspn.ops.bottleneck_tail_bf16
    try:
        def spy(*a, **k):
            calls.append(k.get("next_frag1") is not None)
            return real(*a, **k)
        tspn.ops.bottleneck_tail_bf16 = spy
        tspn.roi_head.ops.bottleneck_tail_bf16 = spy
        for nxt, ns in ((False, 1), (True, 1), (True, 2), (False, 2)):
            net.fuse_next_conv1, net.streams = nxt, ns
            calls.clear()
            outs.append(net(img, bf16=True))
            torch.cuda.synchronize()
            # three frame chunks x (res2 + res3 + four res4 blocks); with the hand-over on, res4 blocks 0..2 of every chunk
            # compute their follower's conv1 (block 0 has a projection shortcut itself, its FOLLOWER qualifies)
            assert len(calls) == 3 * 6 and sum(calls) == (9 if nxt else 0)
    finally:
        tspn.ops.bottleneck_tail_bf16 = real
        tspn.roi_head.ops.bottleneck_tail_bf16 = real
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert net.res4[2]._can_take_h1(outs[0]) is False            # switched off by the last configuration
    net.fuse_next_conv1 = True
    net(img, bf16=True)
    assert net.res4[2]._can_take_h1(outs[0]) and not net.res4[0]._can_take_h1(outs[0])


def test_backbone_and_roi_head_fused_equal_unfused(tspn, device):
    """ResNetC4 / Res5RoIHead with fuse_bottlenecks on and off: the same bf16 maps and features, bit for bit (every
    block incl. the stride-2 ones with a projection shortcut goes through the fused tail)."""
    blocks = (2, 2, 2)
    net, _ = _backbone_and_weights(tspn, device, 64, 256, blocks)
    img = t(tspn.hashrng.uniform(91, "img", (3, 64, 96, 3), -1, 1)).to(device)
    net.fuse_bottlenecks = True
    a = net(img, bf16=True)
    net.fuse_bottlenecks = False
    b = net(img, bf16=True)
    assert a.dtype == torch.bfloat16 and torch.equal(a, b)
    head = tspn.Res5RoIHead(1024, 128, 512).to(device)
    boxes = torch.tensor([[[4.0, 4, 60, 50], [10, 8, 90, 60], [0, 0, 95, 63]], [[0, 0, 95, 63], [30, 20, 50, 40], [1, 2, 3, 4]]],
                         device=device)
    head.fuse_bottlenecks = True
    fa = head(a, boxes)
    head.fuse_bottlenecks = False
    fb = head(a, boxes)
    assert torch.equal(fa, fb)
    with pytest.raises(ValueError):
        tspn.ops.bottleneck_tail_bf16(torch.zeros((1, 4, 4, 32), dtype=torch.bfloat16, device=device), a, a.float(), a, a.float(), a)


def test_roi_align_bin_stride_and_presampled_first_block(tspn, device):
    """ROIAlign with bin_stride = s == every s-th bin of the full grid, bit for bit (fp32 and bf16, odd P); and the RoI
    head that lets ROIAlign produce only the bins res5's strided 1x1 convs read equals the head on the full 14 x 14 grid."""
    feat = tspn.hashrng.uniform(92, "f", (2, 9, 12, 64), -1, 1)
    rois = np.array([[0, 10, 20, 100, 120], [1, 0, 0, 191, 143], [1, -30, -10, 60, 50], [0, 150, 100, 260, 200]], dtype=np.float32)
    for P, bs in ((14, 2), (5, 2), (7, 3), (4, 1)):
        full = tspn.ops.roi_align_nhwc(t(feat).to(device), t(rois).to(device), P, 1.0 / 16)
        sub = tspn.ops.roi_align_nhwc(t(feat).to(device), t(rois).to(device), P, 1.0 / 16, bin_stride=bs)
        assert torch.equal(sub, full[:, ::bs, ::bs].contiguous())
        f16 = t(feat).to(torch.bfloat16).to(device)
        assert torch.equal(tspn.ops.roi_align_nhwc(f16, t(rois).to(device), P, 1.0 / 16, bin_stride=bs),
                           tspn.ops.roi_align_nhwc(f16, t(rois).to(device), P, 1.0 / 16)[:, ::bs, ::bs].contiguous())
    with pytest.raises(ValueError):
        tspn.ops.roi_align_nhwc(t(feat).to(device), t(rois).to(device), 7, 1.0 / 16, bin_stride=0)
    head, _ = _head_and_weights(tspn, device, 128, 64, 256, roi_chunk=7)
    fm = t(tspn.hashrng.uniform(93, "fm", (4, 10, 12, 128), 0, 1))
    boxes = t((tspn.synth.make_video(94, 3, 4, 16)["tracklet_boxes"] * np.float32(0.15)).astype(np.float32)).to(device)
    for maps in (fm.to(device), fm.to(torch.bfloat16).to(device)):
        head.subsample_roi_align = True
        a = head(maps, boxes)
        head.subsample_roi_align = False
        b = head(maps, boxes)
        assert torch.equal(a, b)


@pytest.mark.parametrize("blocks", [(1, 1, 2), (1, 1, 1)])
def test_backbone_and_roi_head_on_two_streams_equal_one(tspn, device, blocks):
    """ResNetC4.streams / Res5RoIHead.streams: chunks alternating between HIP streams give the same maps and features
    as one stream, bit for bit, also when the caller itself works on a non-default stream.  With more than one res4
    block every chunk's last block writes its frames of the result in place (bf16: the fused tail's `out`; fp32: a
    copy); a one-block res4 (its only block strided) takes the concatenating path."""
    net, _ = _backbone_and_weights(tspn, device, 64, 256, blocks)
    net.frame_chunk = 2
    img = t(tspn.hashrng.uniform(96, "img", (7, 64, 96, 3), -1, 1)).to(device)
    head = tspn.Res5RoIHead(1024, 128, 512, roi_chunk=5).to(device)
    boxes = t(tspn.hashrng.uniform(96, "bx", (3, 7, 2), 0, 30)).to(device)
    boxes = torch.cat([boxes, boxes + 25], dim=2).contiguous()
    outs = []
    for ns, ctx in ((1, None), (2, None), (3, torch.cuda.Stream(device=device))):
        net.streams, head.streams = ns, ns
        if ctx is None:
            m = net(img, bf16=True)
            f = head(m, boxes)
        else:
            ctx.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(ctx):
                m = net(img, bf16=True)
                f = head(m, boxes)
        torch.cuda.synchronize()
        outs.append((m, f))
    for m, f in outs[1:]:
        assert torch.equal(m, outs[0][0]) and torch.equal(f, outs[0][1])
    m32 = net(img, bf16=False)                       # fp32 path through the stream-alternating loop as well
    net.streams = 1
    assert torch.equal(m32, net(img, bf16=False))


def test_first_forward_on_two_streams_behind_a_busy_gpu(tspn, device):
    """the very FIRST forward of a fresh backbone / RoI head with streams = 2, queued behind a
    long-running kernel so that both side streams are released together: the folded / packed weights must have been
    built on the caller's stream before either side stream may read them.  Reference result: a second, identical
    module on one stream."""
    nets = [_backbone_and_weights(tspn, device, 64, 256, (1, 1, 2))[0] for _ in range(2)]
    heads = [tspn.Res5RoIHead(1024, 128, 512, roi_chunk=5).to(device) for _ in range(2)]
    heads[1].load_state_dict(heads[0].state_dict())
    img = t(tspn.hashrng.uniform(97, "img", (6, 64, 96, 3), -1, 1)).to(device)
    boxes = t(tspn.hashrng.uniform(97, "bx", (3, 6, 2), 0, 30)).to(device)
    boxes = torch.cat([boxes, boxes + 25], dim=2).contiguous()
    for net in nets:
        net.frame_chunk = 2
    nets[0].streams, heads[0].streams = 1, 1
    ref_m = nets[0](img, bf16=True)
    ref_f = heads[0](ref_m, boxes)
    torch.cuda.synchronize()
    nets[1].streams, heads[1].streams = 2, 2
    busy = torch.randn(8192, 8192, device=device)
    for _ in range(6):
        busy = busy @ busy * 1e-4              # tens of milliseconds of queued work on the caller's stream
    m = nets[1](img, bf16=True)                # caches are cold here
    f = heads[1](m, boxes)
    torch.cuda.synchronize()
    assert torch.equal(m, ref_m) and torch.equal(f, ref_f)


@pytest.mark.parametrize("seed", range(6))
def test_bottleneck_tail_and_stem_random_shapes(tspn, device, seed):
    """Shape fuzz of the two round-3 backbone kernels (hash-RNG shapes): the fused tail against the two conv launches
    it replaces (bit-identical) on odd maps -- single rows / columns, tiles that straddle images --, the stem against
    the bf16 restatement on odd image sizes."""
    r = lambda tag, lo, hi: int(tspn.hashrng.integers(700 + seed, tag, (1,), lo, hi)[0])   # noqa: E731
    CM = (64, 128, 256)[r("cm", 0, 3)]
    NB, H, W = r("nb", 1, 5), r("h", 1, 24), r("w", 1, 40)
    h1 = t(tspn.hashrng.uniform(700 + seed, "h1", (NB, H, W, CM), 0, 1)).to(device).to(torch.bfloat16)
    res = t(tspn.hashrng.uniform(700 + seed, "res", (NB, H, W, 4 * CM), -1, 1)).to(device).to(torch.bfloat16)
    w2 = t(tspn.hashrng.normal(700 + seed, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))).to(device)
    w3 = t(tspn.hashrng.normal(700 + seed, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))).to(device)
    b2 = t(tspn.hashrng.normal(700 + seed, "b2", (CM,), std=0.1)).to(device)
    b3 = t(tspn.hashrng.normal(700 + seed, "b3", (4 * CM,), std=0.1)).to(device)
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    h2 = tspn.ops.conv2d_nhwc_bf16(h1, f2, (3, 3), 1, 1, bias=b2, relu=True)
    want = tspn.ops.conv2d_nhwc_bf16(h2, f3, (1, 1), 1, 0, bias=b3, residual=res, relu=True)
    got = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
    assert torch.equal(got, want), (CM, NB, H, W)
    IH, IW, Cout = r("ih", 1, 70), r("iw", 1, 300), (32, 64)[r("co", 0, 2)]
    x = tspn.hashrng.uniform(700 + seed, "x", (NB, IH, IW, 3), -2, 2)
    w = tspn.hashrng.normal(700 + seed, "w", (Cout, 3, 7, 7), std=0.1)
    b = tspn.hashrng.normal(700 + seed, "b", (Cout,), std=0.1)
    ref = ro.conv2d_bf16(t(x).permute(0, 3, 1, 2), t(w), t(b), stride=2, padding=3, relu=True).permute(0, 2, 3, 1)
    y = tspn.ops.stem_conv_bf16(t(x).to(device), tspn.ops.pack_stem_bf16(t(w).to(device)), t(b).to(device))
    assert tuple(y.shape) == tuple(ref.shape), (IH, IW, Cout)
    err = (y.cpu().double() - ref).abs()
    assert float(err.max()) <= 2.0 ** -8 * max(float(ref.abs().max()), 1e-3), (IH, IW, Cout, float(err.max()))
    frag = tspn.ops.pack_stem_bf16(t(w).to(device))
    assert torch.equal(tspn.ops.stem_pool_bf16(t(x).to(device), frag, t(b).to(device)),
                       tspn.ops.max_pool_nhwc_bf16(y, 3, 2, 1)), (IH, IW, Cout)


@pytest.mark.parametrize("seed", range(8))
def test_bottleneck_tail_role_split_random_shapes(tspn, device, seed):
    """Shape fuzz of the role-split res4 tail (hash-RNG shapes: single rows / columns, maps narrower than a tile, tiles that
    straddle images): bit-identical to the one-role kernel into a poisoned output."""
    r = lambda tag, lo, hi: int(tspn.hashrng.integers(900 + seed, tag, (1,), lo, hi)[0])   # noqa: E731
    CM = 256
    NB, H, W = r("nb", 1, 6), r("h", 1, 24), r("w", 1, 40)
    h1 = t(tspn.hashrng.uniform(900 + seed, "h1", (NB, H, W, CM), 0, 1)).to(device).to(torch.bfloat16)
    res = t(tspn.hashrng.uniform(900 + seed, "res", (NB, H, W, 4 * CM), -1, 1)).to(device).to(torch.bfloat16)
    w2 = t(tspn.hashrng.normal(900 + seed, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))).to(device)
    w3 = t(tspn.hashrng.normal(900 + seed, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))).to(device)
    b2 = t(tspn.hashrng.normal(900 + seed, "b2", (CM,), std=0.1)).to(device)
    b3 = t(tspn.hashrng.normal(900 + seed, "b3", (4 * CM,), std=0.1)).to(device)
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
    out = torch.full_like(want, 777.0)
    tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=out, io_waves=True)
    bad = int((out != want).sum())
    assert bad == 0, (NB, H, W, bad)


def test_bottleneck_tail_role_split_on_maps_beyond_two_gigabytes(tspn, device):
    """300 res4 maps (residual and output 2.2 GB each): the role-split tail bases its descriptors at the tile, so image by
    image it must give what the same image gives alone -- first, middle and last image."""
    NB, H, W, CM = 300, 45, 80, 256
    assert NB * H * W * 4 * CM * 2 > 2 ** 31
    g = torch.Generator(device=device).manual_seed(7)
    h1 = torch.empty((NB, H, W, CM), dtype=torch.bfloat16, device=device)
    res = torch.empty((NB, H, W, 4 * CM), dtype=torch.bfloat16, device=device)
    for lo in range(0, NB, 50):                                   # filled in slices: no multi-GB fp32 temporaries
        h1[lo:lo + 50] = torch.rand((50, H, W, CM), device=device, generator=g).to(torch.bfloat16)
        res[lo:lo + 50] = (torch.rand((50, H, W, 4 * CM), device=device, generator=g) - 0.5).to(torch.bfloat16)
    w2 = (torch.rand((CM, CM, 3, 3), device=device, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=device, generator=g) - 0.5) * 0.1
    b2, b3 = torch.rand(CM, device=device, generator=g) - 0.5, torch.rand(4 * CM, device=device, generator=g) - 0.5
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    out = torch.full((NB, H, W, 4 * CM), 777.0, dtype=torch.bfloat16, device=device)
    tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=out, io_waves=True)
    for i in (0, NB // 2, NB - 1):
        alone = tspn.ops.bottleneck_tail_bf16(h1[i:i + 1].contiguous(), f2, b2, f3, b3, res[i:i + 1].contiguous())
        assert torch.equal(out[i:i + 1], alone), f"image {i}"
    assert int((out == 777.0).sum()) < out.numel() // 100        # (a ReLU output is never 777 by construction; zeros are common)


def test_bf16_convs_on_maps_beyond_two_gigabytes(tspn, device):
    """The operand pieces, residual reads and stores of the bf16 conv kernels are buffer instructions with 32-bit offsets
    from a per-tile base: a map of more than 2^31 bytes (76 res2 maps of 720p) must give, image by image, what the same
    image gives alone -- first, middle and last image, the generic 1x1 conv and the fused tail."""
    NB, H, W, C4, CM = 76, 180, 320, 256, 64
    assert NB * H * W * C4 * 2 > 2 ** 31
    g = torch.Generator(device=device).manual_seed(5)
    one = lambda shape, s=1.0: ((torch.rand(shape, device=device, generator=g) - 0.5) * s).to(torch.bfloat16)
    x = torch.empty((NB, H, W, C4), dtype=torch.bfloat16, device=device)
    for lo in range(0, NB, 19):                                   # filled in slices: no 9-GB fp32 temporary
        x[lo:lo + 19] = one((min(19, NB - lo), H, W, C4))
    w1 = (torch.rand((CM, C4, 1, 1), device=device, generator=g) - 0.5) * 0.1
    w2 = (torch.rand((CM, CM, 3, 3), device=device, generator=g) - 0.5) * 0.1
    w3 = (torch.rand((C4, CM, 1, 1), device=device, generator=g) - 0.5) * 0.1
    b1, b2, b3 = (torch.rand(n, device=device, generator=g) - 0.5 for n in (CM, CM, C4))
    f1, f2, f3 = (tspn.ops.pack_conv2d_frag_bf16(w) for w in (w1, w2, w3))
    h1 = tspn.ops.conv2d_nhwc_bf16(x, f1, (1, 1), 1, 0, bias=b1, relu=True)
    y = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, x)
    for i in (0, NB // 2, NB - 1):
        h1_i = tspn.ops.conv2d_nhwc_bf16(x[i:i + 1].contiguous(), f1, (1, 1), 1, 0, bias=b1, relu=True)
        assert torch.equal(h1[i:i + 1], h1_i), i
        y_i = tspn.ops.bottleneck_tail_bf16(h1_i, f2, b2, f3, b3, x[i:i + 1].contiguous())
        assert torch.equal(y[i:i + 1], y_i), i
