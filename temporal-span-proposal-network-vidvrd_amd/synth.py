"""Synthetic VidVRD-shaped inputs and random-init weights.

Distributions follow the reference's initialisers:
  classifier / conv / 1x1 heads ~ N(0, 0.01^2), bias 0   (model.py:81-83, dpn.py:65-67)
  PPN MLPs ~ U(+-1/sqrt(fan_in))                          (nn.Linear default, ppn.py:95-105)
Inputs: RoI feats ~ U[0,1); boxes integer-valued x,y in [0,900), w,h in [10,300),
stored (l,t,r,b) = (x, y, x+w, y+h); class logits ~ U[0,1).
Seeds: weights seed 0, inputs seed 1 + video index.
All values come from hashrng (platform independent), as numpy arrays.
"""
import numpy as np

from . import hashrng


def make_video(seed, n, t, d, num_obj=35):
    feats = hashrng.uniform(seed, "tracklet_feats", (n, t, d))
    xy = hashrng.integers(seed, "box_xy", (n, t, 2), 0, 900)
    wh = hashrng.integers(seed, "box_wh", (n, t, 2), 10, 300)
    boxes = np.concatenate([xy, xy + wh], axis=-1).astype(np.float32)
    cls = hashrng.uniform(seed, "track_cls_logits", (n, num_obj))
    return {"tracklet_feats": feats, "tracklet_boxes": boxes, "track_cls_logits": cls}


def make_baseline_features(seed, p, f=11070):
    """cfg1: `features [P,F] ~ U[0,1)` (before `_feature_preprocess`)."""
    return hashrng.uniform(seed, "baseline_features", (p, f))


def make_weights(seed, c, a=4, k=132, feat_dim=None, ppn=(35, 64, 35), bias_std=0.0):
    """State-dict-shaped weights (reference key names).

    `c` = DPN.IN_CHANNELS, `feat_dim` = PREDICT.FEATURE_DIM (defaults to c).
    `bias_std` > 0 gives non-zero biases so tests catch bias-handling bugs
    (the reference initialises them to 0).
    """
    feat_dim = c if feat_dim is None else feat_dim
    pin, ph, pout = ppn
    pre = "relpn.duration_proposal_network.dpn_head."
    ppre = "relpn.pair_proposal_network.ppn_head."

    def b(tag, n):
        if bias_std == 0.0:
            return np.zeros((n,), np.float32)
        return hashrng.normal(seed, tag, (n,), std=bias_std)

    sd = {
        pre + "conv.weight": hashrng.normal(seed, "conv.weight", (c, c, 3), std=0.01),
        pre + "conv.bias": b("conv.bias", c),
        pre + "duration_pred.weight": hashrng.normal(seed, "duration_pred.weight", (2 * a, c, 1), std=0.01),
        pre + "duration_pred.bias": b("duration_pred.bias", 2 * a),
        pre + "relness_pred.weight": hashrng.normal(seed, "relness_pred.weight", (a, c, 1), std=0.01),
        pre + "relness_pred.bias": b("relness_pred.bias", a),
        "classifier.rel_predictor.weight": hashrng.normal(seed, "rel_predictor.weight", (k, feat_dim), std=0.01),
        "classifier.rel_predictor.bias": b("rel_predictor.bias", k),
    }
    for emb in ("sub_emb", "obj_emb"):
        for idx, (fo, fi) in (("0", (ph, pin)), ("2", (pout, ph))):
            bound = 1.0 / np.sqrt(fi)
            sd[f"{ppre}{emb}.{idx}.weight"] = hashrng.uniform(seed, f"{emb}.{idx}.weight", (fo, fi), -bound, bound)
            sd[f"{ppre}{emb}.{idx}.bias"] = hashrng.uniform(seed, f"{emb}.{idx}.bias", (fo,), -bound, bound)
    return sd


def _conv_bn(sd, seed, prefix, tag, cout, cin, k, gamma=(0.5, 1.5)):
    """One detectron2 `Conv2d(bias=False, norm=FrozenBN)`: He-normal weight, FrozenBN statistics away from the
    identity so that the folding is exercised."""
    std = float(np.sqrt(2.0 / (cin * k * k)))
    sd[prefix + "weight"] = hashrng.normal(seed, tag + ".w", (cout, cin, k, k), std=std)
    sd[prefix + "norm.weight"] = hashrng.uniform(seed, tag + ".g", (cout,), gamma[0], gamma[1])
    sd[prefix + "norm.bias"] = hashrng.uniform(seed, tag + ".b", (cout,), -0.2, 0.2)
    sd[prefix + "norm.running_mean"] = hashrng.uniform(seed, tag + ".m", (cout,), -0.2, 0.2)
    sd[prefix + "norm.running_var"] = hashrng.uniform(seed, tag + ".v", (cout,), 0.5, 1.5)


def _stage(sd, seed, name, nblocks, cin, cout, gamma3):
    for b in range(nblocks):
        pre, mid = f"{name}.{b}.", cout // 4
        _conv_bn(sd, seed, pre + "conv1.", pre + "conv1", mid, cin, 1)
        _conv_bn(sd, seed, pre + "conv2.", pre + "conv2", mid, mid, 3)
        _conv_bn(sd, seed, pre + "conv3.", pre + "conv3", cout, mid, 1, gamma3)
        if cin != cout:
            _conv_bn(sd, seed, pre + "shortcut.", pre + "shortcut", cout, cin, 1)
        cin = cout
    return cin


def make_backbone_weights(seed, blocks=(3, 4, 23), stem_out=64, res2_out=256):
    """Random-init C4 backbone (detectron2 key names: stem.conv1.*, res2..res4.<b>.{conv1,conv2,conv3,shortcut}.*)
    for `ResNetC4.load_state_dict`; the last batch norm of every block is damped (gamma in [0.1, 0.4]) so that
    activations stay O(1) over the 33 blocks of R-101.  BASELINE cfg5 (detectron/trainer.py:23-33: R101-C4)."""
    sd = {}
    _conv_bn(sd, seed, "stem.conv1.", "stem", stem_out, 3, 7)
    sd["stem.conv1.weight"] = hashrng.normal(seed, "stem.w", (stem_out, 3, 7, 7), std=float(np.sqrt(2.0 / (3 * 49))))
    cin, cout = stem_out, res2_out
    for i, nb in enumerate(blocks):
        cin = _stage(sd, seed, f"res{i + 2}", nb, cin, cout, (0.1, 0.4))
        cout *= 2
    return sd


def make_res5_weights(seed, in_channels=1024, bottleneck_channels=512, out_channels=2048, num_blocks=3):
    """Random-init res5 of the ROI head (`Res5RoIHead.load_state_dict`): three bottleneck blocks, the first with
    a projection shortcut."""
    sd, cin = {}, in_channels
    for b in range(num_blocks):
        pre = f"res5.{b}."
        _conv_bn(sd, seed, pre + "conv1.", pre + "conv1", bottleneck_channels, cin, 1)
        _conv_bn(sd, seed, pre + "conv2.", pre + "conv2", bottleneck_channels, bottleneck_channels, 3)
        _conv_bn(sd, seed, pre + "conv3.", pre + "conv3", out_channels, bottleneck_channels, 1)
        if cin != out_channels:
            _conv_bn(sd, seed, pre + "shortcut.", pre + "shortcut", out_channels, cin, 1)
        cin = out_channels
    return sd
