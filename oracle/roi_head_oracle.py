"""CPU restatement of the RoI feature head (TEST INFRASTRUCTURE — see oracle/__init__.py for who may import).

SURVEY.md §8 row f4.  The reference owns no code for this step: lib/detectron/trainer.py:23-33 only
configures detectron2's `COCO-Detection/faster_rcnn_R_101_C4_3x.yaml` model (35 classes), which is NOT
installed here and not vendored in /root/reference.  Its ROI head is therefore restated from detectron2's
published algorithm (v0.6, `modeling/roi_heads/roi_heads.py:Res5ROIHeads`, `layers/roi_align.py:ROIAlign`
with `aligned=True`, `layers/csrc/ROIAlign/ROIAlign_cpu.cpp`, `modeling/backbone/resnet.py:BottleneckBlock`
with `stride_in_1x1=True`, `layers/batch_norm.py:FrozenBatchNorm2d`):

    box features = ROIAlign(res4 map, boxes; 14 x 14, scale 1/16, sampling_ratio 0, aligned)
                   -> res5 = 3 BottleneckBlocks (first: stride 2, projection shortcut) -> mean over (H, W)

PARITY UNPINNED by the reference (no runnable counterpart, no fixtures): the convolution / batch-norm
pieces are torch's own CPU operators; ROIAlign is restated here line by line from the cited CPU kernel.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = ["roi_align_nhwc", "frozen_bn", "bottleneck_block", "res5_roi_head", "make_res5_weights",
           "conv2d_bf16", "res5_roi_head_bf16", "resnet_c4", "resnet_c4_bf16", "stem_bf16", "make_backbone_weights"]

BN_EPS = 1e-5


def _bilinear(fmap, y, x):
    """ROIAlign_cpu.cpp:pre_calc_for_bilinear_interpolate (one sample point), float32 arithmetic.
    fmap [H,W,C] float32 numpy.  Returns [C] float32."""
    H, W, _ = fmap.shape
    f = np.float32
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros(fmap.shape[2], dtype=np.float32)
    y = max(f(y), f(0))
    x = max(f(x), f(0))
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = f(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = f(xl)
    else:
        xh = xl + 1
    ly, lx = f(y - f(yl)), f(x - f(xl))
    hy, hx = f(f(1) - ly), f(f(1) - lx)
    w1, w2, w3, w4 = f(hy * hx), f(hy * lx), f(ly * hx), f(ly * lx)
    return (w1 * fmap[yl, xl] + w2 * fmap[yl, xh] + w3 * fmap[yh, xl] + w4 * fmap[yh, xh]).astype(np.float32)


def roi_align_nhwc(feat, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    """detectron2 ROIAlign (layers/csrc/ROIAlign/ROIAlign_cpu.cpp:ROIAlignForward) on a channels-last map.
    feat [NF,H,W,C], rois [R,5] = (map index, x1, y1, x2, y2) -> [R,P,P,C]; float32 arithmetic in the
    kernel's order (loops: small cases only)."""
    feat = np.asarray(feat, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    f = np.float32
    P = int(output_size)
    R = rois.shape[0]
    out = np.zeros((R, P, P, feat.shape[3]), dtype=np.float32)
    scale = f(spatial_scale)
    off = f(0.5) if aligned else f(0)
    for r in range(R):
        fmap = feat[int(rois[r, 0])]
        sw, sh = f(rois[r, 1] * scale - off), f(rois[r, 2] * scale - off)
        rw, rh = f(f(rois[r, 3] * scale - off) - sw), f(f(rois[r, 4] * scale - off) - sh)
        if not aligned:
            rw, rh = max(rw, f(1)), max(rh, f(1))
        bin_h, bin_w = f(rh / f(P)), f(rw / f(P))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(f(rh / f(P))))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(f(rw / f(P))))
        count = f(max(gh * gw, 1))
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(feat.shape[3], dtype=np.float32)
                for iy in range(gh):
                    y = f(f(sh + f(f(ph) * bin_h)) + f(f(f(iy) + f(0.5)) * bin_h) / f(gh))
                    for ix in range(gw):
                        x = f(f(sw + f(f(pw) * bin_w)) + f(f(f(ix) + f(0.5)) * bin_w) / f(gw))
                        acc = acc + _bilinear(fmap, y, x)
                out[r, ph, pw] = acc / count
    return torch.from_numpy(out)


def frozen_bn(x, p, prefix):
    """detectron2 FrozenBatchNorm2d.forward (layers/batch_norm.py): x * scale + bias with
    scale = weight * rsqrt(running_var + eps), bias = bias - running_mean * scale.  x NCHW."""
    scale = p[prefix + "weight"] * (p[prefix + "running_var"] + BN_EPS).rsqrt()
    bias = p[prefix + "bias"] - p[prefix + "running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def bottleneck_block(x, p, prefix, stride):
    """detectron2 BottleneckBlock.forward (modeling/backbone/resnet.py), stride_in_1x1=True, FrozenBN.
    x NCHW; `p` maps '<prefix>conv1.weight', '<prefix>conv1.norm.weight', ... to tensors."""
    out = F.relu(frozen_bn(F.conv2d(x, p[prefix + "conv1.weight"], stride=stride), p, prefix + "conv1.norm."))
    out = F.relu(frozen_bn(F.conv2d(out, p[prefix + "conv2.weight"], padding=1), p, prefix + "conv2.norm."))
    out = frozen_bn(F.conv2d(out, p[prefix + "conv3.weight"]), p, prefix + "conv3.norm.")
    if prefix + "shortcut.weight" in p:
        sc = frozen_bn(F.conv2d(x, p[prefix + "shortcut.weight"], stride=stride), p, prefix + "shortcut.norm.")
    else:
        sc = x
    return F.relu(out + sc)


def res5_roi_head(feat_nhwc, tracklet_boxes, p, num_blocks=3, pooler_resolution=14, spatial_scale=1.0 / 16,
                  sampling_ratio=0, first_stride=2, dtype=torch.float32):
    """Res5ROIHeads._shared_roi_transform + mean([2,3]) (modeling/roi_heads/roi_heads.py) for tracklet boxes:
    feat_nhwc [T,Hf,Wf,C] (frame t's res4 map), tracklet_boxes [N,T,4] (l,t,r,b) -> [N,T,Cout]."""
    N, T, _ = tracklet_boxes.shape
    idx = torch.arange(T, dtype=torch.float32).repeat(N)
    rois = torch.cat([idx[:, None], tracklet_boxes.reshape(N * T, 4).float()], dim=1)
    x = roi_align_nhwc(feat_nhwc, rois, pooler_resolution, spatial_scale, sampling_ratio, True)
    x = x.permute(0, 3, 1, 2).contiguous().to(dtype)
    pp = {k: v.to(dtype) for k, v in p.items()}
    for b in range(num_blocks):
        x = bottleneck_block(x, pp, f"res5.{b}.", first_stride if b == 0 else 1)
    return x.mean(dim=(2, 3)).reshape(N, T, -1).float()


def make_res5_weights(rng_uniform, rng_normal, in_channels, bottleneck_channels, out_channels, num_blocks=3):
    """Deterministic random res5 parameters (detectron2 key names) from the build's hash RNG callables
    rng_uniform(name, shape, lo, hi) / rng_normal(name, shape, std) -> numpy float32."""
    p = {}
    cin = in_channels
    for b in range(num_blocks):
        pre = f"res5.{b}."
        convs = [("conv1", bottleneck_channels, cin, 1), ("conv2", bottleneck_channels, bottleneck_channels, 3),
                 ("conv3", out_channels, bottleneck_channels, 1)]
        if cin != out_channels:
            convs.append(("shortcut", out_channels, cin, 1))
        for name, co, ci, k in convs:
            std = math.sqrt(2.0 / (ci * k * k))
            p[pre + name + ".weight"] = torch.from_numpy(rng_normal(pre + name + ".w", (co, ci, k, k), std))
            p[pre + name + ".norm.weight"] = torch.from_numpy(rng_uniform(pre + name + ".g", (co,), 0.5, 1.5))
            p[pre + name + ".norm.bias"] = torch.from_numpy(rng_uniform(pre + name + ".b", (co,), -0.2, 0.2))
            p[pre + name + ".norm.running_mean"] = torch.from_numpy(rng_uniform(pre + name + ".m", (co,), -0.2, 0.2))
            p[pre + name + ".norm.running_var"] = torch.from_numpy(rng_uniform(pre + name + ".v", (co,), 0.5, 1.5))
        cin = out_channels
    return p


def _r16(x):
    return x.float().to(torch.bfloat16).double()


def conv2d_bf16(x, w, bias, stride=1, padding=0, residual=None, relu=False):
    """bf16-operand conv (build-defined semantics of csrc/tspn_roi_bf16.hip): x, w, residual are bf16
    VALUES (rounded here), products exact, sums in float64 here (fp32 on the GPU), bias fp32;
    act(sum + bias + residual) rounded to bf16 once.  x NCHW, returns NCHW float64 holding bf16 values."""
    y = F.conv2d(_r16(x), _r16(w), bias.double() if bias is not None else None, stride=stride, padding=padding)
    if residual is not None:
        y = y + _r16(residual)
    if relu:
        y = F.relu(y)
    return _r16(y)


def res5_roi_head_bf16(feat_nhwc, tracklet_boxes, p, num_blocks=3, pooler_resolution=14, spatial_scale=1.0 / 16,
                       sampling_ratio=0, first_stride=2):
    """The head with bf16 operands: ROIAlign in fp32 on the bf16 map values -> bf16; every conv as
    conv2d_bf16 with the batch norm folded in fp32 (w * scale, bias - mean * scale) before the single
    rounding of the weight; spatial mean rounded to bf16."""
    N, T, _ = tracklet_boxes.shape
    idx = torch.arange(T, dtype=torch.float32).repeat(N)
    rois = torch.cat([idx[:, None], tracklet_boxes.reshape(N * T, 4).float()], dim=1)
    fm = feat_nhwc.float().to(torch.bfloat16).float()
    x = _r16(roi_align_nhwc(fm, rois, pooler_resolution, spatial_scale, sampling_ratio, True).permute(0, 3, 1, 2))

    def fold(prefix):
        scale = p[prefix + "norm.weight"] * (p[prefix + "norm.running_var"] + BN_EPS).rsqrt()
        return (p[prefix + "weight"] * scale.reshape(-1, 1, 1, 1)), (p[prefix + "norm.bias"] - p[prefix + "norm.running_mean"] * scale)

    for b in range(num_blocks):
        pre, s = f"res5.{b}.", (first_stride if b == 0 else 1)
        w1, b1 = fold(pre + "conv1.")
        w2, b2 = fold(pre + "conv2.")
        w3, b3 = fold(pre + "conv3.")
        out = conv2d_bf16(x, w1, b1, stride=s, relu=True)
        out = conv2d_bf16(out, w2, b2, padding=1, relu=True)
        if pre + "shortcut.weight" in p:
            ws, bs = fold(pre + "shortcut.")
            sc = conv2d_bf16(x, ws, bs, stride=s)
        else:
            sc = x
        x = conv2d_bf16(out, w3, b3, residual=sc, relu=True)
    return _r16(x.mean(dim=(2, 3))).reshape(N, T, -1).float()


def _make_stage(p, rng_uniform, rng_normal, name, nblocks, cin, cout, gamma3=(0.5, 1.5)):
    for b in range(nblocks):
        pre = f"{name}.{b}."
        mid = cout // 4
        convs = [("conv1", mid, cin, 1, (0.5, 1.5)), ("conv2", mid, mid, 3, (0.5, 1.5)), ("conv3", cout, mid, 1, gamma3)]
        if cin != cout:
            convs.append(("shortcut", cout, cin, 1, (0.5, 1.5)))
        for cname, co, ci, k, gam in convs:
            std = math.sqrt(2.0 / (ci * k * k))
            p[pre + cname + ".weight"] = torch.from_numpy(rng_normal(pre + cname + ".w", (co, ci, k, k), std))
            p[pre + cname + ".norm.weight"] = torch.from_numpy(rng_uniform(pre + cname + ".g", (co,), gam[0], gam[1]))
            p[pre + cname + ".norm.bias"] = torch.from_numpy(rng_uniform(pre + cname + ".b", (co,), -0.2, 0.2))
            p[pre + cname + ".norm.running_mean"] = torch.from_numpy(rng_uniform(pre + cname + ".m", (co,), -0.2, 0.2))
            p[pre + cname + ".norm.running_var"] = torch.from_numpy(rng_uniform(pre + cname + ".v", (co,), 0.5, 1.5))
        cin = cout
    return cin


def make_backbone_weights(rng_uniform, rng_normal, stem_out, res2_out, blocks):
    """Deterministic random C4-backbone parameters with detectron2 key names (stem.conv1.*, res2..res4.*); the
    last batch norm of every block is damped (gamma in [0.1, 0.4]) so that activations stay O(1) over depth."""
    p = {}
    std = math.sqrt(2.0 / (3 * 49))
    p["stem.conv1.weight"] = torch.from_numpy(rng_normal("stem.w", (stem_out, 3, 7, 7), std))
    p["stem.conv1.norm.weight"] = torch.from_numpy(rng_uniform("stem.g", (stem_out,), 0.5, 1.5))
    p["stem.conv1.norm.bias"] = torch.from_numpy(rng_uniform("stem.b", (stem_out,), -0.2, 0.2))
    p["stem.conv1.norm.running_mean"] = torch.from_numpy(rng_uniform("stem.m", (stem_out,), -0.2, 0.2))
    p["stem.conv1.norm.running_var"] = torch.from_numpy(rng_uniform("stem.v", (stem_out,), 0.5, 1.5))
    cin, cout = stem_out, res2_out
    for i, nb in enumerate(blocks):
        cin = _make_stage(p, rng_uniform, rng_normal, f"res{i + 2}", nb, cin, cout, gamma3=(0.1, 0.4))
        cout *= 2
    return p


def resnet_c4(images_nhwc, p, blocks, dtype=torch.float64):
    """detectron2 ResNet.forward up to res4 (modeling/backbone/resnet.py: BasicStem = conv 7x7/2 + FrozenBN +
    ReLU + max_pool2d(3, 2, 1); stages of BottleneckBlocks, stride 2 in the first block of res3 and res4).
    images [T,H,W,3] -> res4 map [T,H/16,W/16,C] channels-last."""
    pp = {k: v.to(dtype) for k, v in p.items()}
    x = images_nhwc.permute(0, 3, 1, 2).to(dtype)
    x = F.relu(frozen_bn(F.conv2d(x, pp["stem.conv1.weight"], stride=2, padding=3), pp, "stem.conv1.norm."))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for i, nb in enumerate(blocks):
        for b in range(nb):
            x = bottleneck_block(x, pp, f"res{i + 2}.{b}.", 2 if (b == 0 and i > 0) else 1)
    return x.permute(0, 2, 3, 1).contiguous().float()


def _fold(p, prefix):
    scale = p[prefix + "norm.weight"] * (p[prefix + "norm.running_var"] + BN_EPS).rsqrt()
    return (p[prefix + "weight"] * scale.reshape(-1, 1, 1, 1)), (p[prefix + "norm.bias"] - p[prefix + "norm.running_mean"] * scale)


def _bottleneck_bf16(x, p, pre, s):
    w1, b1 = _fold(p, pre + "conv1.")
    w2, b2 = _fold(p, pre + "conv2.")
    w3, b3 = _fold(p, pre + "conv3.")
    out = conv2d_bf16(x, w1, b1, stride=s, relu=True)
    out = conv2d_bf16(out, w2, b2, padding=1, relu=True)
    if pre + "shortcut.weight" in p:
        ws, bs = _fold(p, pre + "shortcut.")
        sc = conv2d_bf16(x, ws, bs, stride=s)
    else:
        sc = x
    return conv2d_bf16(out, w3, b3, residual=sc, relu=True)


def stem_bf16(images_nhwc, p):
    """detectron2 BasicStem on bf16 operands: relu(conv7x7/2/pad3 + folded FrozenBN) rounded to bf16 once, then
    max_pool2d(3, 2, 1).  images [T,H,W,3] -> NCHW float64 holding bf16 values."""
    w, b = _fold(p, "stem.conv1.")
    x = conv2d_bf16(images_nhwc.permute(0, 3, 1, 2), w, b, stride=2, padding=3, relu=True)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def resnet_c4_bf16(images_nhwc, p, blocks):
    """The backbone as the GPU runs it with bf16=True: EVERY conv a conv2d_bf16 (bf16 operands -- the image and
    the folded stem weight included since round 3 --, exact products, one rounding of act(sum + bias)); the max
    pool acts on the rounded stem map (max commutes with the monotone rounding, so this equals pool-then-round)."""
    x = stem_bf16(images_nhwc, p)
    for i, nb in enumerate(blocks):
        for bidx in range(nb):
            x = _bottleneck_bf16(x, p, f"res{i + 2}.{bidx}.", 2 if (bidx == 0 and i > 0) else 1)
    return x.permute(0, 2, 3, 1).contiguous().float()
