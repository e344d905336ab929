"""Drop-in `BaseModel` for the TSPN relation-scoring hot path on MI355X.

Mirrors the reference's module tree so that `BaseModel(cfg)`, `forward(pair_list,
target_list)`, the return conventions and the `state_dict` key names are the same
(reference lib/modeling/model.py:7-88, relpn/relpn.py:9-60, relpn/ppn.py:7-112,
relpn/dpn.py:9-81), while every tensor operation of the eval
forward runs in the hand-written HIP library (ops.py -> C ABI).  The nn.Linear /
nn.Conv1d objects below are parameter holders only; their own forward is never
called.

Differences from the reference, all in code the reference cannot execute:
  * USE_DPN=True: the reference raises NameError (relpn/dpn.py:24-28).  Here the
    temporal branch runs: pair builder -> temporal encoder -> relationness +
    span-regression heads, RelOIPool over the segment, predicate head
    (semantics frozen in DESIGN.md §2); in train mode it returns the loss the
    reference intended (`loss_duration`, relpn/dpn.py:40-49).
  * `relness_pred` (relpn/dpn_anchor.py:88-90) is an extra parameter; reference
    checkpoints that lack it still load (see DPNHead._load_from_state_dict).
There is no CPU execution path: without the HIP library or a HIP device,
forward raises.
"""
import warnings
import weakref
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _abi, ops
from .sampler import BalancedPositiveNegativePairSampler

TemporalProposals = namedtuple("TemporalProposals", ["relness", "duration", "heads", "geom"], defaults=[None])
TemporalProposals.__doc__ = """Per-segment output of the temporal branch:
relness [P,A,T] relationness logits, duration [P,2A,T] span regression (views of heads [P,3A,T]);
geom [P,8,T]: the bbox half of the pair builder (relative box geometry per pair and frame, DESIGN.md §2) when
the segment carries 'tracklet_boxes', else None."""


def _compute_device(*tensors):
    """Device the HIP path runs on: that of the first HIP tensor seen, else the current HIP device."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("TSPN BaseModel: no HIP device available; this build has no CPU path "
                           "(the reference's CPU forward is restated only as the test oracle)")
    return torch.device("cuda", torch.cuda.current_device())


def invalidate_caches(module):
    """Drop every packed / device-resident weight copy held under `module` (see _DeviceCache)."""
    for m in module.modules():
        for cache in vars(m).values():
            if isinstance(cache, _DeviceCache):
                cache.clear()
        if hasattr(m, "_logits_token"):
            m._logits_token = None       # never let a decode ride the ready-event of a pass made with other weights


class _CachedWeightsMixin:
    """nn.Module mix-in: anything that can replace parameter values wholesale clears the weight caches."""

    def invalidate_caches(self):
        invalidate_caches(self)

    def train(self, mode=True):
        invalidate_caches(self)
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        invalidate_caches(self)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        invalidate_caches(self)
        return super().load_state_dict(*args, **kwargs)


def _f32(t, device):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.to(device=device, dtype=torch.float32).contiguous()


def _consecutive_view(tensors):
    """Zero-copy batch of equal-shape tensors that already lie back to back in ONE allocation (the slices
    `big[b*N:(b+1)*N]` of a batched tensor a loader handed out per segment): the [len * rows, ...] view of that
    allocation, or None when the tensors are not laid out that way."""
    t0 = tensors[0]
    if not isinstance(t0, torch.Tensor) or t0.dim() == 0 or not t0.is_contiguous():
        return None
    step = t0.numel() * t0.element_size()
    base = t0.untyped_storage().data_ptr()
    for k, t in enumerate(tensors):
        if (not isinstance(t, torch.Tensor) or t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device
                or not t.is_contiguous() or t.data_ptr() != t0.data_ptr() + k * step
                or t.untyped_storage().data_ptr() != base):
            return None
    if t0.storage_offset() * t0.element_size() + len(tensors) * step > t0.untyped_storage().nbytes():
        return None
    return torch.as_strided(t0, (len(tensors) * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride())


def _batch_rows(tensors, device, dtype=torch.float32):
    """cat(tensors) on `device` as `dtype` — without the copy when they are consecutive slices of one tensor."""
    t0 = tensors[0]
    if isinstance(t0, torch.Tensor) and t0.device == device and t0.dtype == dtype:
        v = _consecutive_view(tensors)
        if v is not None:
            return v
    conv = []
    for t in tensors:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        conv.append(t.to(device=device, dtype=dtype).contiguous())
    return conv[0] if len(conv) == 1 else torch.cat(conv)


def _segment_pairs(plist, n, device):
    """int64 [P,2] pair table of a tracklet segment: its 'tracklet_pairs' field or all ordered pairs."""
    if plist.has_field("tracklet_pairs") and plist.get_field("tracklet_pairs") is not None:
        p = plist.get_field("tracklet_pairs")
        p = (p.detach() if isinstance(p, torch.Tensor) else torch.as_tensor(np.asarray(p))).long().to(device)
        if p.dim() != 2 or p.shape[1] != 2:
            raise ValueError("tracklet_pairs must be [P,2]")
        if p.numel() and (int(p.min()) < 0 or int(p.max()) >= n):
            raise IndexError("tracklet_pairs index out of range")
        return p.contiguous()
    return ops.pair_index(n, device)


_VERIFY_WEIGHTS = __import__("os").environ.get("TSPN_VERIFY_WEIGHTS", "0") not in ("", "0")


class _DeviceCache:
    """Device-resident (and packed) copies of parameters, refreshed when a parameter changes.

    Contract: a change is seen through the parameter's storage pointer, its autograd version counter and
    its device — optimiser steps, `load_state_dict`, `.to()` / `.cuda()` and `train()` all refresh the
    cache (the last three through `invalidate_caches`).  In-place edits through `.data` (`p.data.copy_`,
    EMA updates, legacy loaders) bump none of these: call `model.invalidate_caches()` after them, or run with
    TSPN_VERIFY_WEIGHTS=1 (a content fingerprint per parameter and forward: safe with any training loop, slower)."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, device, build):
        sig = tuple((p.data_ptr(), p._version, str(p.device)) for p in params) + (str(device),)
        if _VERIFY_WEIGHTS:
            # TSPN_VERIFY_WEIGHTS=1: a content fingerprint joins the signature, so in-place `.data` edits (EMA, legacy
            # loaders) refresh the cache too -- at the price of a device reduction and a host sync per parameter and call
            with torch.no_grad():
                sig += tuple((float(p.detach().double().sum()), float(p.detach().double().abs().sum())) for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            value = build([_f32(p.detach(), device) for p in params])
        self._store[key] = (sig, value)
        return value

    def clear(self):
        self._store.clear()


def _nt(a, b, bias=None, sigmoid=False):
    """a [M,K] . b [N,K]^T (+ bias [N]) on the HIP split-K MFMA GEMM of the predicate head (tspn_predicate_head_f32): the
    one GEMM form the library has, so every product of the training step is brought to it with HIP transposes."""
    m, k = a.shape
    n = b.shape[0]
    if m == 0 or n == 0 or k == 0:
        # an empty product (no sampled pairs / an empty batch): the sum over nothing is zero -- the C entry refuses
        # zero-length contractions, the reference's torch.mm / einsum return zeros (and zero gradients) here
        out = torch.zeros((m, n), dtype=torch.float32, device=a.device)
        if bias is not None:
            out = out + bias
        return torch.sigmoid(out) if sigmoid else out
    return ops.predicate_head(a.contiguous(), b.contiguous(), None if bias is None else bias.contiguous(), apply_sigmoid=sigmoid)


def _tr(a):
    """[M,N] -> [N,M] contiguous (tspn_transpose_td_f32)."""
    return ops.transpose_td(a.contiguous().unsqueeze(0))[0]


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with HIP forward AND backward (round 4; the backward was torch / rocBLAS GEMMs before):
        dx = g W = nt(g, W^T),   dW = g^T x = nt(g^T, x^T),   db = column sums of g  (tspn_temporal_sum_f32)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _nt(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = _nt(g, _tr(w)) if ctx.needs_input_grad[0] else None
        gw = _nt(_tr(g), _tr(x)) if ctx.needs_input_grad[1] else None
        gb = ops.temporal_sum(g.unsqueeze(0))[0] if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _MatmulNTFn(torch.autograd.Function):
    """a b^T with HIP forward and backward: da = g b = nt(g, b^T), db = g^T a = nt(g^T, a^T)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _nt(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = _nt(g, _tr(b)) if ctx.needs_input_grad[0] else None
        gb = _nt(_tr(g), _tr(a)) if ctx.needs_input_grad[1] else None
        return ga, gb


class _PredicateHeadFn(torch.autograd.Function):
    """sigmoid(x W^T + b): HIP forward (sigmoid inside the GEMM's reduce kernel) and HIP backward (see _LinearFn)."""

    @staticmethod
    def forward(ctx, x, w, b):
        out = ops.predicate_head(x, w, b, apply_sigmoid=True)
        ctx.save_for_backward(x, w, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        gz = (g * out * (1.0 - out)).contiguous()
        gx = _nt(gz, _tr(w)) if ctx.needs_input_grad[0] else None
        gw = _nt(_tr(gz), _tr(x)) if ctx.needs_input_grad[1] else None
        gb = ops.temporal_sum(gz.unsqueeze(0))[0] if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _rows_of(x):
    """[R, C, T] -> [C, R*T] contiguous (a torch permute + copy: byte movement, no arithmetic)."""
    return x.permute(1, 0, 2).reshape(x.shape[1], -1).contiguous()


def _conv3_weight_grad(x_cf, dz):
    """dL/dW of y = Conv1d(k=3, padding=1)(x): W[m, c, k] = sum_{n,t} dz[n, m, t] x[n, c, t + k - 1]: per tap one product
    dZ [M, R T] . X_k [C, R T]^T on the library's HIP GEMM (round 4; three torch einsums before); x_cf [R, Cin, T], dz [R, M, T]."""
    xp = F.pad(x_cf, (1, 1))
    t = x_cf.shape[2]
    dzr = _rows_of(dz)
    return torch.stack([_nt(dzr, _rows_of(xp[:, :, k:k + t])) for k in range(3)], dim=2)


def _heads_backward(head_w, g, act):
    """Backward of heads = head_w . act (+ b) for g [P,H,T], act [P,C,T] = relu(...):
    dZ [P,C,T] = (head_w^T g) . [act > 0]  and  d_head_w [H,C] = sum_{p,t} g act -- two products on the HIP GEMM."""
    p, h, t = g.shape
    c = act.shape[1]
    gt = g.permute(0, 2, 1).reshape(p * t, h).contiguous()              # [(p,t), H]
    dz = _nt(gt, _tr(head_w)).view(p, t, c).permute(0, 2, 1) * (act > 0)  # [(p,t), C] -> [P,C,T]
    d_head_w = _nt(_rows_of(g), _rows_of(act))                          # [H, P T] . [C, P T]^T
    return dz.contiguous(), d_head_w


class _TemporalHeadsDenseFn(torch.autograd.Function):
    """DPNHead on a materialised x [P,C,T] (reference relpn/dpn.py:69-73) for training: HIP forward
    (tspn_temporal_encoder_heads_f32); backward recomputes the encoder activation with the HIP conv and forms every
    product on the library's HIP GEMM / conv kernels (round 4; torch einsums before)."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, head_w, head_b):
        heads = ops.temporal_encoder_heads(x, ops.pack_conv3(conv_w), conv_b, head_w, head_b)
        ctx.save_for_backward(x, conv_w, conv_b, head_w)
        return heads

    @staticmethod
    def backward(ctx, g):
        x, conv_w, conv_b, head_w = ctx.saved_tensors
        g = g.contiguous()
        act = ops.conv3(x, ops.pack_conv3(conv_w), conv_b, relu=True)          # relu(conv(x) + b) [P,C,T]
        dz, d_head_w = _heads_backward(head_w, g, act)
        dx = None
        if ctx.needs_input_grad[0]:
            # conv1d_input = the k=3 conv of dZ with the taps reversed and the channel roles swapped
            dx = ops.conv3(dz, ops.pack_conv3(conv_w.flip(2).transpose(0, 1).contiguous()), None, relu=False)
        return dx, _conv3_weight_grad(x, dz), dz.sum((0, 2)), d_head_w, g.sum((0, 2))


class _TemporalHeadsTrackletFn(torch.autograd.Function):
    """Factorised pair encoder + heads on tracklet tensors for training.  Forward: the HIP
    projections U‖V (tspn_conv3_tc_f32 on the split-packed weight) and the indexed pair stage
    (tspn_heads_f32).  Backward: dZ_p = (head_w^T g_p) . [relu(U[s]+V[o]) > 0] per block of pairs (HIP GEMM),
    scattered back onto dU[s], dV[o]; the conv weight gradient is then a per-TRACKLET contraction (the
    same N-1 saving as the forward)."""

    PAIR_BLOCK = 256

    @staticmethod
    def forward(ctx, feats, pairs, conv_w, conv_b, head_w, head_b):
        d = feats.shape[2]
        c = 2 * d
        bias2 = torch.cat([conv_b, torch.zeros_like(conv_b)])
        y = ops.conv3_tc(feats, ops.pack_conv3(conv_w, split=d), bias2)          # [NT, 2C, T]
        s, o = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()
        heads = ops.heads(y[:, :c].contiguous(), head_w, head_b, b=y[:, c:].contiguous(), ia=s, ib=o)
        ctx.save_for_backward(feats, pairs, conv_w, conv_b, head_w)
        return heads

    @staticmethod
    def backward(ctx, g):
        feats, pairs, conv_w, conv_b, head_w = ctx.saved_tensors
        g = g.contiguous()
        d = feats.shape[2]
        c = 2 * d
        bias2 = torch.cat([conv_b, torch.zeros_like(conv_b)])
        y = ops.conv3_tc(feats, ops.pack_conv3(conv_w, split=d), bias2)
        u, v = y[:, :c], y[:, c:]
        du, dv = torch.zeros_like(u), torch.zeros_like(v)
        d_head_w = torch.zeros_like(head_w)
        for lo in range(0, pairs.shape[0], _TemporalHeadsTrackletFn.PAIR_BLOCK):
            blk = slice(lo, lo + _TemporalHeadsTrackletFn.PAIR_BLOCK)
            s, o, gb = pairs[blk, 0], pairs[blk, 1], g[blk].contiguous()
            act = torch.relu(u[s] + v[o])
            dz, dhw = _heads_backward(head_w, gb, act)
            d_head_w += dhw
            du.index_add_(0, s, dz)
            dv.index_add_(0, o, dz)
        x_cf = feats.transpose(1, 2)
        d_conv_w = torch.cat([_conv3_weight_grad(x_cf, du), _conv3_weight_grad(x_cf, dv)], dim=1)
        return None, None, d_conv_w, du.sum((0, 2)), d_head_w, g.sum((0, 2))


class RelationPredictor(nn.Module):
    """Predicate-classification head (reference lib/modeling/model.py:76-88)."""

    # block-L1 layout of the 11070-d baseline feature (lib/dataset/vrdataset.py:227-236)
    PREPROCESS_BLOCKS = (70, 1000, 8)

    def __init__(self, in_channels, out_channels, fuse_preprocess=False):
        super().__init__()
        self.rel_predictor = nn.Linear(in_channels, out_channels)
        nn.init.normal_(self.rel_predictor.weight, std=0.01)
        nn.init.constant_(self.rel_predictor.bias, 0)
        # PREDICT.FUSE_PREPROCESS: features arrive RAW (the DataLoader skips _feature_preprocess) and
        # the normalisation is folded into the predicate GEMM; eval only
        self.fuse_preprocess = bool(fuse_preprocess)
        self._cache = _DeviceCache()

    def forward(self, reloi_feats):
        dev = _compute_device(reloi_feats, self.rel_predictor.weight)
        x = _f32(reloi_feats, dev)
        w, b = self.rel_predictor.weight, self.rel_predictor.bias
        if torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
            if not w.is_cuda:
                raise RuntimeError("training needs the model on the HIP device (model.cuda())")
            out = _PredicateHeadFn.apply(x, w.contiguous(), b.contiguous())
        else:
            wd, bd = self._cache.get("cls", (w, b), dev, lambda ts: ts)
            first, block, nblocks = self.PREPROCESS_BLOCKS
            norm = self.PREPROCESS_BLOCKS if (self.fuse_preprocess and x.dim() == 2
                                              and x.shape[1] >= first + block * nblocks) else None
            out = ops.predicate_head(x, wd, bd, apply_sigmoid=True, norm=norm)
        return out.to(reloi_feats.device)


class RelOIPool:
    """Relation-of-interest pooling (reference lib/modeling/model.py:68-73).

    2-D feats [P,F] pass through (duration_proposals None).  Temporal feats
    [P,C,T] are averaged over the segment (build-defined)."""

    def __call__(self, feats, duration_proposals):
        if duration_proposals is None:
            return feats
        return [ops.temporal_mean(f, layout_tc=False) if f.dim() == 3 else f for f in feats]


class PPNHead(nn.Module):
    """Tracklet-level relationness embeddings (reference lib/modeling/relpn/ppn.py:92-112)."""

    def __init__(self, in_channels, hidden_channels, out_channels):
        super().__init__()
        self.sub_emb = nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.ReLU(True),
                                     nn.Linear(hidden_channels, out_channels))
        self.obj_emb = nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.ReLU(True),
                                     nn.Linear(hidden_channels, out_channels))

    def weights(self):
        names = ["sub_emb.0.weight", "sub_emb.0.bias", "sub_emb.2.weight", "sub_emb.2.bias",
                 "obj_emb.0.weight", "obj_emb.0.bias", "obj_emb.2.weight", "obj_emb.2.bias"]
        sd = dict(self.named_parameters())
        return names, [sd[n] for n in names]

    def forward(self, sub_logits, obj_logits):
        """Autograd (training) form: every GEMM of the two MLPs and of the pair matrix, forward and backward, on the
        library's HIP GEMM (round 4; torch.mm before); ReLU / sigmoid are elementwise torch ops.  Eval goes through
        PPN.propose (one fused launch)."""
        def mlp(seq, x):
            h = torch.relu(_LinearFn.apply(x.contiguous(), seq[0].weight, seq[0].bias))
            return _LinearFn.apply(h, seq[2].weight, seq[2].bias)
        if not sub_logits.is_cuda:
            raise RuntimeError("training needs the inputs on the HIP device")
        s = mlp(self.sub_emb, sub_logits.float())
        o = mlp(self.obj_emb, obj_logits.float())
        return torch.sigmoid(_MatmulNTFn.apply(s, o))


class PPN(nn.Module):
    """Pair Proposal Network (reference lib/modeling/relpn/ppn.py:7-90)."""

    def __init__(self, cfg):
        super().__init__()
        self.num_pair_proposals = cfg.RELPN.PPN.NUM_PAIR_PROPOSALS
        self.ppn_head = PPNHead(cfg.RELPN.PPN.IN_CHANNELS, cfg.RELPN.PPN.HIDDEN_CHANNELS,
                                cfg.RELPN.PPN.OUT_CHANNELS)
        # the reference also constructs (and never calls) a balanced sampler: ppn.py:20-23
        self.batch_size_per_segment = cfg.RELPN.PPN.BATCH_SIZE_PER_SEGMENT
        self.positive_fraction = cfg.RELPN.PPN.POSITIVE_FRACTION
        # constructed like the reference (relpn/ppn.py:20-23); it never calls it either
        self.fg_bg_sampler = BalancedPositiveNegativePairSampler(
            batch_size_per_image=self.batch_size_per_segment, positive_fraction=self.positive_fraction)
        self._cache = _DeviceCache()

    def device_weights(self, dev):
        """Device-resident copies of the two MLPs (cached).  BaseModel calls this on the CALLER's stream before it
        sends `propose` to its side stream, so the copies are made, and their memory owned, by the caller's stream."""
        names, params = self.ppn_head.weights()
        return self._cache.get("ppn", params, dev, lambda ts: dict(zip(names, ts)))

    def propose(self, cls_logits):
        """Eval: one fused HIP launch per group of equal-N segments -> (matrices, top-k indices)."""
        _, params = self.ppn_head.weights()
        dev = _compute_device(*cls_logits, params[0])
        w = self.device_weights(dev)
        mats, idxs = [None] * len(cls_logits), [None] * len(cls_logits)
        groups = {}
        for i, c in enumerate(cls_logits):
            groups.setdefault(tuple(c.shape), []).append(i)
        for shape, members in groups.items():
            batch = _batch_rows([cls_logits[i] for i in members], dev).view((len(members),) + tuple(shape))
            mat, idx = ops.ppn_pair_matrix_topk(batch, w, self.num_pair_proposals)
            for k, i in enumerate(members):
                mats[i] = mat[k].to(cls_logits[i].device)
                idxs[i] = idx[k].to(cls_logits[i].device)
        return mats, idxs

    @staticmethod
    def _gt_matrices(pair_list, target_list):
        """N x N 0/1 matrices: 1 where any predicate is set for (s,o) (ppn.py:36-49, host side)."""
        out = []
        for plist, tlist in zip(pair_list, target_list):
            pairs = torch.as_tensor(np.asarray(plist.get_field("tracklet_pairs"))).long().cpu()
            n = int(plist.get_field("num_tracklets"))
            pos = (tlist.target.detach().sum(dim=1) > 0).cpu()
            gt = torch.zeros(n, n)
            m = min(len(pairs), len(pos))   # zip(track_pair, pred_label) stops at the shorter one (ppn.py:44)
            sel = pairs[:m][pos[:m]]
            gt[sel[:, 0], sel[:, 1]] = 1
            out.append(gt)
        return out

    def forward(self, pair_list, target_list=None):
        cls_logits = [plist.get_field("track_cls_logits") for plist in pair_list]
        if not self.training:
            _, idx = self.propose(cls_logits)
            return idx, {}
        gts = self._gt_matrices(pair_list, target_list)
        proposals, loss = [], 0
        for c, gt in zip(cls_logits, gts):
            pm = self.ppn_head(c, c)
            loss = loss + F.binary_cross_entropy(pm, gt.to(pm.device))
            order = torch.sort(pm.detach().view(-1), descending=True, stable=True)[1]
            proposals.append(order[: self.num_pair_proposals])
        return proposals, {"loss_pair": loss}


class DPNHead(nn.Module):
    """Temporal context encoder + span-regression (+ relationness) heads
    (reference lib/modeling/relpn/dpn.py:55-73; relness_pred from relpn/dpn_anchor.py:82-108)."""

    def __init__(self, in_channels, num_windows):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.duration_pred = nn.Conv1d(in_channels, num_windows * 2, kernel_size=1, stride=1)
        self.relness_pred = nn.Conv1d(in_channels, num_windows, kernel_size=1, stride=1)
        for layer in (self.conv, self.duration_pred, self.relness_pred):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.in_channels = in_channels
        self.num_windows = num_windows

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        # reference checkpoints (dpn.py) carry no relness_pred: keep the initialised values
        for leaf in ("relness_pred.weight", "relness_pred.bias"):
            if prefix + leaf not in state_dict:
                state_dict[prefix + leaf] = dict(self.named_parameters())[leaf].detach().clone()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def head_params(self):
        return (self.relness_pred.weight, self.relness_pred.bias,
                self.duration_pred.weight, self.duration_pred.bias)


class DPN(nn.Module):
    """Duration Proposal Network (reference lib/modeling/relpn/dpn.py:9-52), runnable."""

    def __init__(self, cfg, in_channels, num_windows):
        super().__init__()
        self.dpn_head = DPNHead(in_channels, num_windows)
        self.top_k_proposals = cfg.RELPN.DPN.NUM_DURATION_PROPOSALS  # RelNMS stub: rel_nms.py:11
        self._cache = _DeviceCache()

    # packed device weights -------------------------------------------------
    def _head_weights(self, dev):
        def build(ts):
            rw, rb, dw, db = ts
            return (torch.cat([rw[:, :, 0], dw[:, :, 0]], dim=0).contiguous(),
                    torch.cat([rb, db]).contiguous())
        return self._cache.get("heads", self.dpn_head.head_params(), dev, build)

    def _conv_dense(self, dev):
        c = self.dpn_head.conv
        return self._cache.get("conv_dense", (c.weight, c.bias), dev,
                               lambda ts: (ops.pack_conv3(ts[0]), ts[1]))

    def _conv_split(self, dev, winograd=False):
        """Packed weights of the factorised pair form (subject / object halves stacked along M): Winograd
        F(6,3) fragment-major weights (needs D % 32 == 0) or the three direct taps."""
        c = self.dpn_head.conv
        half = self.dpn_head.in_channels // 2
        if winograd:
            return self._cache.get("conv_split_wino63", (c.weight, c.bias), dev,
                                   lambda ts: (ops.pack_conv3_wino63(ts[0], split=half), ts[1]))
        return self._cache.get("conv_split", (c.weight, c.bias), dev,
                               lambda ts: (ops.pack_conv3(ts[0], split=half), ts[1]))

    def _conv_raw(self, dev):
        """conv.weight as it is (fp32, contiguous, on `dev`): what the accuracy guard recomputes outputs from."""
        return self._cache.get("conv_raw", (self.dpn_head.conv.weight,), dev, lambda ts: ts[0])

    def _bf16_weights(self, dev):
        """bf16 operand set of the factorised path (csrc/tspn_bf16.hip): packed conv / head weights
        in bf16, biases as fp32 tensors holding bf16-rounded values."""
        c = self.dpn_head.conv
        half = self.dpn_head.in_channels // 2
        rnd = lambda t: ops.cast_bf16(t.contiguous()).float()

        def build(ts):
            cw, cb, rw, rb, dw, db = ts
            hw = torch.cat([rw[:, :, 0], dw[:, :, 0]], dim=0).contiguous()
            return (ops.pack_conv3_bf16(cw, split=half), rnd(cb), ops.pack_heads_bf16(hw),
                    rnd(torch.cat([rb, db])))
        return self._cache.get("bf16", (c.weight, c.bias) + tuple(self.dpn_head.head_params()), dev, build)

    def _wrap(self, heads, geom=None):
        a = self.dpn_head.num_windows
        return TemporalProposals(heads[:, :a], heads[:, a:], heads, geom)

    def forward_dense(self, feats):
        """feats: list of materialised [P,C,T] pair tensors (the layout DPNHead consumes)."""
        out = []
        for f in feats:
            dev = _compute_device(f, self.dpn_head.conv.weight)
            packed, cbias = self._conv_dense(dev)
            hw, hb = self._head_weights(dev)
            out.append(self._wrap(ops.temporal_encoder_heads(_f32(f, dev), packed, cbias, hw, hb)))
        return out

    def _train_heads(self, plist):
        """Heads [P,3A,T] of one segment with autograd onto the DPNHead parameters."""
        h = self.dpn_head
        if not h.conv.weight.is_cuda:
            raise RuntimeError("training needs the model on the HIP device (model.cuda())")
        dev = h.conv.weight.device
        rw, rb, dw, db = h.head_params()
        head_w = torch.cat([rw[:, :, 0], dw[:, :, 0]], dim=0)
        head_b = torch.cat([rb, db])
        if plist.has_field("tracklet_feats"):
            f = _f32(plist.get_field("tracklet_feats"), dev)
            if 2 * f.shape[2] != h.in_channels:
                raise ValueError(f"tracklet_feats dim D={f.shape[2]} needs RELPN.DPN.IN_CHANNELS = 2*D")
            return _TemporalHeadsTrackletFn.apply(f, _segment_pairs(plist, f.shape[0], dev), h.conv.weight,
                                                  h.conv.bias, head_w, head_b)
        x = plist.features
        if x.dim() != 3:
            raise ValueError("RELPN.USE_DPN=True needs temporal inputs: PairList.features [P,C,T] or the "
                             "field 'tracklet_feats' [N,T,D] (the reference raises NameError here, "
                             "relpn/dpn.py:24-28)")
        return _TemporalHeadsDenseFn.apply(_f32(x, dev), h.conv.weight, h.conv.bias, head_w, head_b)

    def _forward_train(self, pair_list, target_list):
        """The reference's intent (relpn/dpn.py:40-49; its own code raises NameError):
        duration_proposals = dpn_head(pair feats); loss_duration = BCEWithLogits(duration_proposals,
        target.get_field('duration')), summed over the segments like loss_rel (model.py:62-64).  A target
        field 'relness' [P,A,T], when present, adds `loss_relationness` on the relationness head (loss name
        from relpn/dpn_anchor.py:62-65)."""
        if target_list is None:
            raise ValueError("DPN training needs target_list with a 'duration' field [P,2A,T]")
        props, loss_dur, loss_rel = [], 0, None
        for plist, tlist in zip(pair_list, target_list):
            tp = self._wrap(self._train_heads(plist))
            gt = tlist.get_field("duration")
            gt = (gt if isinstance(gt, torch.Tensor) else torch.as_tensor(np.asarray(gt))).to(tp.duration)
            if gt.shape != tp.duration.shape:
                raise ValueError(f"target field 'duration' must be [P,2A,T] = {tuple(tp.duration.shape)}, "
                                 f"got {tuple(gt.shape)}")
            loss_dur = loss_dur + F.binary_cross_entropy_with_logits(tp.duration, gt)
            if tlist.has_field("relness"):
                gr = tlist.get_field("relness")
                gr = (gr if isinstance(gr, torch.Tensor) else torch.as_tensor(np.asarray(gr))).to(tp.relness)
                loss_rel = (0 if loss_rel is None else loss_rel) + F.binary_cross_entropy_with_logits(tp.relness, gr)
            props.append(tp)
        losses = {"loss_duration": loss_dur}
        if loss_rel is not None:
            losses["loss_relationness"] = loss_rel
        return props, losses

    def forward(self, pair_list, target_list=None):
        if self.training:
            return self._forward_train(pair_list, target_list)
        return self.forward_dense([plist.features for plist in pair_list]), {}


class RelPN(nn.Module):
    """PPN + DPN dispatcher (reference lib/modeling/relpn/relpn.py:9-60)."""

    def __init__(self, cfg):
        super().__init__()
        self.use_ppn = cfg.RELPN.USE_PPN
        self.use_dpn = cfg.RELPN.USE_DPN
        self.pair_proposal_network = PPN(cfg)
        self.duration_proposal_network = DPN(cfg, in_channels=cfg.RELPN.DPN.IN_CHANNELS,
                                             num_windows=cfg.RELPN.DPN.NUM_ANCHORS_PER_LOCATION)

    def forward(self, pair_list, target_list=None):
        losses, pair_props, dur_props = {}, None, None
        if self.use_ppn:
            pair_props, l = self.pair_proposal_network(pair_list, target_list)
            losses.update(l)
        if self.use_dpn:
            dur_props, l = self.duration_proposal_network(pair_list, target_list)
            losses.update(l)
        return pair_props, dur_props, losses


def make_relpn(cfg):
    return RelPN(cfg)


def _usable_cpus():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container that shows 256
    cores and grants 16 throttles a 128-thread copy to a crawl)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


_STAGE_POOL = None


def _stage_pool():
    """A few single-threaded copy workers for the pageable -> pinned staging copies (torch's intra-op pool is sized by
    the visible cores, not by the quota, and is shared with everything else the caller runs)."""
    global _STAGE_POOL
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        nt = max(1, min(8, _usable_cpus() - 1))
        _STAGE_POOL = (ThreadPoolExecutor(nt, thread_name_prefix="tspn-stage", initializer=lambda: torch.set_num_threads(1)), nt)
    return _STAGE_POOL


class _HostPipeline:
    """Transfers of one forward whose tracklet features live in HOST memory (what the reference's predict.py:50-57
    hands over: CPU PairLists from a DataLoader): the videos of a group go to the device in chunks on a copy stream,
    the fused pass runs chunk by chunk behind the chunks' copy events, and every chunk's results come back on a third
    stream into pinned tensors - so chunk k+1 uploads, and chunk k-1 downloads, under the encoder of chunk k.
    Pinned sources (`PairList.pin_memory()`, `DataLoader(pin_memory=True)`) are DMA'd as they are; pageable ones
    are staged through a pinned buffer kept here (a multi-threaded host copy), one chunk ahead of the GPU."""

    def __init__(self, dev):
        self.dev = dev
        self.h2d = torch.cuda.Stream(device=dev)
        self.d2h = torch.cuda.Stream(device=dev)
        self.dev_buf, self.pin_buf = {}, {}
        self.compute_done = None        # event on the caller's stream: the last pass that read dev_buf
        self.h2d_done = None            # event on the copy stream: the last DMA out of pin_buf

    @staticmethod
    def schedule(nm, chunk):
        """[(lo, hi)) video ranges.  Only the first upload is exposed, and a fused pass over few videos is less efficient
        than one over many (cfg2: 2.41 ms per video at 2 videos, 2.07 at 4, 1.72 at 16): start with half a chunk, then
        `chunk`, then double while every upload still hides under the pass before it; the last range takes the rest
        (16 videos, chunk 4: 2 + 4 + 10)."""
        chunk = max(1, int(chunk))
        out, lo, size = [], 0, max(1, chunk // 2)
        while lo < nm:
            rest = nm - lo
            nxt = chunk if not out else 2 * size
            hi = nm if (out and rest <= nxt + nxt // 4) else min(nm, lo + size)
            out.append((lo, hi))
            lo, size = hi, nxt
        return out

    def begin(self, src, dtype):
        """Device buffer [nm*n, t, d] for the group + the staging state; nothing is copied yet."""
        nm, (n, t, d) = len(src), tuple(src[0].shape)
        key = (nm, n, t, d, dtype)
        for cache in (self.dev_buf, self.pin_buf):
            if key not in cache and len(cache) >= 4:
                cache.clear()
        if key not in self.dev_buf:
            self.dev_buf[key] = torch.empty((nm * n, t, d), dtype=dtype, device=self.dev)
        self.src, self.n, self.dst = src, n, self.dev_buf[key]
        self.direct = all(x.dtype == dtype and x.is_contiguous() and x.is_pinned() for x in src)
        self.pin = None
        if not self.direct:
            if key not in self.pin_buf:
                self.pin_buf[key] = torch.empty((nm * n, t, d), dtype=dtype, pin_memory=True)
            self.pin = self.pin_buf[key]
            if self.h2d_done is not None:
                self.h2d_done.synchronize()        # the previous forward's DMAs out of the staging buffer are done
        if self.compute_done is not None:
            self.h2d.wait_event(self.compute_done)  # ... and its passes no longer read the device buffer
        # dev_buf comes out of the CALLER stream's pool: a freshly handed-out block may still be in use by kernels
        # queued on that stream (the allocator only orders reuse within one stream), and the first write into it is a
        # DMA on h2d -- so the copy stream starts behind everything the caller has queued so far
        self.h2d.wait_stream(torch.cuda.current_stream(self.dev))
        return self.dst

    def stage(self, lo, hi):
        """Videos [lo, hi): (host copy into the pinned staging buffer,) async DMA on the copy stream; returns the event
        the compute stream has to wait for."""
        n = self.n
        if not self.direct:
            # host copies into the pinned staging buffer (cast included), row blocks spread over the copy workers
            pool, nt = _stage_pool()
            parts = max(1, min(n, -(-nt // (hi - lo))))
            step = -(-n // parts)
            jobs = [(v * n + r, min(v * n + r + step, (v + 1) * n), v, r) for v in range(lo, hi) for r in range(0, n, step)]
            list(pool.map(lambda j: self.pin[j[0]:j[1]].copy_(self.src[j[2]][j[3]:j[3] + (j[1] - j[0])]), jobs))
        for v in range(lo, hi):
            x = self.src[v] if self.direct else self.pin[v * n:(v + 1) * n]
            with torch.cuda.stream(self.h2d):
                self.dst[v * n:(v + 1) * n].copy_(x, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self.h2d)
        if not self.direct:
            self.h2d_done = ev
        return ev

    def download(self, main, pairs):
        """`pairs` = [(pinned host tensor, device tensor), ...] of one chunk, complete on `main` now: copied back on the
        download stream."""
        ev = torch.cuda.Event()
        ev.record(main)
        self.d2h.wait_event(ev)
        with torch.cuda.stream(self.d2h):
            for host, devt in pairs:
                host.copy_(devt, non_blocking=True)

    def finish(self, main):
        ev = torch.cuda.Event()
        ev.record(main)
        self.compute_done = ev
        self.d2h.synchronize()             # the caller gets host tensors: they must be complete


class BaseModel(_CachedWeightsMixin, nn.Module):
    """RelPN -> RelOIPool -> predicate classifier (reference lib/modeling/model.py:7-65)."""

    def __init__(self, cfg):
        super().__init__()
        self.use_ppn = cfg.RELPN.USE_PPN
        self.use_dpn = cfg.RELPN.USE_DPN
        self.relpn = make_relpn(cfg)
        self.rel_of_interest_pool = RelOIPool()
        self.classifier = RelationPredictor(in_channels=cfg.PREDICT.FEATURE_DIM,
                                            out_channels=cfg.PREDICT.PREDICATE_NUM,
                                            fuse_preprocess=getattr(cfg.PREDICT, "FUSE_PREPROCESS", False))
        self._anchor_sizes_cfg = getattr(cfg.RELPN.DPN, "ANCHOR_SIZES", None)
        self.pool_top_span = bool(getattr(cfg.RELPN.DPN, "POOL_TOP_SPAN", False))
        # build extension: also return the bbox half of the pair builder (relative geometry [P,8,T]) from the
        # eval forward.  Nothing downstream of forward consumes it, so it is off unless asked for
        # (`model.pair_geometry(pair_list)` computes it on demand).
        self.pair_geometry_in_forward = bool(getattr(cfg.RELPN.DPN, "PAIR_GEOMETRY", False))
        self._workspaces = {}        # (device index, stream) -> uint8 workspace of the fused pass, grown on demand
        self._pair_tables = {}       # (device, B, N) -> (batched canonical pair table [B*P,2], per-video table [B,P,2])
        self._conv_events = None     # optional (begin, end) torch.cuda.Event pair around the dominant kernel
        # build extension: the small latency-bound kernels of a step (PPN; the top-k triplet decode of `decode`) run on
        # a second HIP stream under the encoder of the same forward: PPN needs the class logits only, decode only the
        # predicate logits, which the fused pass computes FIRST and signals with an event.  Both join the caller's
        # stream before their results are handed out, so the caller sees ordinary stream semantics.
        self.overlap_tail = bool(getattr(cfg.RELPN, "OVERLAP_TAIL", True))
        self._side = {}              # device index -> (side stream, logits-ready event)
        self._logits_token = None    # (device index, the last fused pass's batched logits tensor, caller stream[, ids of
        #                              the host tensors handed out for them])
        # host-resident inputs (predict.py hands CPU PairLists): videos per pipelined chunk, see _HostPipeline
        self.host_chunk_videos = int(getattr(cfg.RELPN.DPN, "HOST_CHUNK_VIDEOS", 4))
        self._host_pipes = {}
        self.conv_algo = str(getattr(cfg.RELPN.DPN, "CONV_ALGO", "auto"))
        if self.conv_algo not in ("auto", "direct"):
            raise ValueError(f"RELPN.DPN.CONV_ALGO must be auto or direct (got {self.conv_algo})")
        # accuracy guard of "auto" (round 6): every fused pass that runs F(6,3) also spot-checks CONV_CHECK_ROWS output
        # rows against a float64 recomputation (csrc/tspn_conv_guard.hip); a measured error above CONV_TOL makes this
        # model warn once and use the direct kernel from its next call on (`conv_fallback`)
        self.conv_tol = float(getattr(cfg.RELPN.DPN, "CONV_TOL", 1e-4))
        self.conv_check_rows = int(getattr(cfg.RELPN.DPN, "CONV_CHECK_ROWS", 128))
        self.conv_fallback = False
        self.conv_err_seen = 0.0     # largest spot-check error this model has read back so far

    def forward(self, pair_list, target_list=None):
        if self.training:
            return self._forward_train(pair_list, target_list)
        return self._forward_test(pair_list)

    _conv_guard_owner = {}       # device index -> weakref of the model whose guarded F(6,3) pass ran last on that device

    def _winograd(self, d, dev):
        """Temporal conv algorithm of this call: True = Winograd F(6,3) (RELPN.DPN.CONV_ALGO "auto", D % 32 == 0, and the
        accuracy guard has not tripped).  Reads what the guard measured in EARLIER calls from the device's status block
        (pinned host memory: no synchronisation) -- the contract is a plain fp32 Conv1d (reference relpn/dpn.py:69-73) to
        1e-4, and F(6,3)'s fp32 error depends on the data."""
        if self.conv_algo != "auto" or d % 32 or self.conv_fallback:
            return False
        if self.conv_check_rows > 0:
            words = ops.status_words(dev)
            # the words are per DEVICE: a measurement belongs to the model whose guarded pass ran last there; what another
            # model (or a direct caller of ops.conv3_spot_check) left is discarded, not attributed to this one
            owner = BaseModel._conv_guard_owner.get(dev.index)
            mine = owner is not None and owner() is self
            BaseModel._conv_guard_owner[dev.index] = weakref.ref(self)
            if not mine:
                words[_abi.STATUS_CONV_ERR] = 0
                words[_abi.STATUS_CONV_CHECKS] = 0
                return True
            err = float(words[_abi.STATUS_CONV_ERR:_abi.STATUS_CONV_ERR + 1].view(np.float32)[0])
            self.conv_err_seen = max(self.conv_err_seen, err)
            if err > self.conv_tol:
                checks = int(words[_abi.STATUS_CONV_CHECKS])
                words[_abi.STATUS_CONV_ERR] = 0
                words[_abi.STATUS_CONV_CHECKS] = 0
                self.conv_fallback = True
                warnings.warn(
                    f"TSPN: the Winograd F(6,3) temporal conv measured an absolute error of {err:.3g} against float64 "
                    f"on this model's inputs ({checks} outputs spot-checked), above RELPN.DPN.CONV_TOL = {self.conv_tol:g}: "
                    "using the direct kernel (reference-order fp32 taps, about 2.2x slower) from this call on.  Set "
                    "RELPN.DPN.CONV_ALGO = 'direct' to start there, or raise CONV_TOL if the features' scale makes "
                    "1e-4 absolute meaningless (INTEGRATION.md §3).", RuntimeWarning, stacklevel=3)
                return False
        return True

    def _conv_guard(self, dpn, dev, winograd):
        """(raw conv.weight, rows) for the fused pass's spot check, or (None, 0)."""
        if winograd and self.conv_check_rows > 0:
            return dpn._conv_raw(dev), self.conv_check_rows
        return None, 0

    # ------------------------------------------------------------------ train
    def _forward_train(self, pair_list, target_list):
        loss_dict = {}
        duration_proposals = None
        feats = [plist.features for plist in pair_list]
        targets = [tlist.target for tlist in target_list]
        if self.use_ppn or self.use_dpn:
            _, duration_proposals, relpn_losses = self.relpn(pair_list, target_list)
            loss_dict.update(relpn_losses)
        if duration_proposals is not None:
            # RelOIPool over the segment; tracklet samples: cat of the two tracklet means
            dev = self.classifier.rel_predictor.weight.device
            feats = [self._pooled_pair_feats(p, dev) if self._is_tracklet_sample(p) else _f32(p.features, dev)
                     for p in pair_list]
        reloi_feats = self.rel_of_interest_pool(feats, duration_proposals)
        loss_relation = 0
        for reloi_feat, target in zip(reloi_feats, targets):
            rel_logit = self.classifier(reloi_feat)
            loss_relation = loss_relation + F.binary_cross_entropy(rel_logit, target)
        loss_dict["loss_rel"] = loss_relation
        return loss_dict

    # ------------------------------------------------------------------- eval
    @staticmethod
    def _is_tracklet_sample(plist):
        return plist.has_field("tracklet_feats")

    @staticmethod
    def _pooled_pair_feats(plist, dev):
        """[P, 2D] = cat(mean_t f[s], mean_t f[o]): the segment-pooled pair feature of a tracklet sample."""
        f = _f32(plist.get_field("tracklet_feats"), dev)
        return ops.pair_rows(ops.temporal_mean(f, layout_tc=True), _segment_pairs(plist, f.shape[0], dev))

    def profile_conv_events(self, events):
        """(begin, end) torch.cuda.Event pair (enable_timing=True, recorded once) re-recorded around the dominant
        kernel of the next fused forwards — the hook bench.py's roofline figure uses; None switches it off."""
        self._conv_events = events

    def _workspace(self, dev, nbytes):
        """Scratch of the fused pass, kept on the module and reused by every forward on the same (device, stream):
        launches on one stream are ordered, so the buffer is free again when the next pass starts."""
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            self._workspaces.pop(key, None)
            while len(self._workspaces) >= 4:        # streams come and go: never hold more than four (5 GB each at cfg2)
                self._workspaces.pop(next(iter(self._workspaces)))
            ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
            self._workspaces[key] = ws
        return ws

    def _canonical_pairs(self, dev, b, n):
        """(global pair table [B*N(N-1), 2] with video b's ids offset by b*N, per-video table [B, N(N-1), 2]) of B
        equal-N videos, built once per (device, B, N)."""
        key = (str(dev), b, n)
        hit = self._pair_tables.get(key)
        if hit is None:
            if len(self._pair_tables) >= 16:
                self._pair_tables.clear()
            local = ops.pair_index(n, dev)
            allp = (local.unsqueeze(0) + (torch.arange(b, device=dev, dtype=torch.int64) * n).view(b, 1, 1))
            hit = (allp.reshape(-1, 2).contiguous(), local.unsqueeze(0).expand(b, -1, -1).contiguous())
            self._pair_tables[key] = hit
        return hit

    def _side_stream(self, dev):
        hit = self._side.get(dev.index)
        if hit is None:
            ev = torch.cuda.Event()
            with torch.cuda.device(dev):
                side = torch.cuda.Stream(device=dev)
                ev.record()                        # creates the HIP event handle the ABI hook re-records
            hit = (side, ev)
            self._side[dev.index] = hit
        return hit

    def _forward_test(self, pair_list):
        with torch.no_grad():
            if self.use_dpn and len(pair_list) and all(self._is_tracklet_sample(p) for p in pair_list):
                return self._forward_test_fused(pair_list)
            if self.use_dpn and any(p.features.dim() != 3 for p in pair_list):
                raise ValueError(
                    "RELPN.USE_DPN=True needs temporal inputs: either PairList.features of shape "
                    "[P,C,T] or the fields 'tracklet_feats' [N,T,D] (+ 'tracklet_boxes'); got 2-D "
                    "features (the reference raises NameError here, relpn/dpn.py:24-28)")
            feats = [plist.features for plist in pair_list]
            pair_proposals, duration_proposals, _ = self.relpn(pair_list)
            reloi_feats = self.rel_of_interest_pool(
                [_f32(f, _compute_device(f, self.classifier.rel_predictor.weight)) if f.dim() == 3 else f
                 for f in feats], duration_proposals)
            rel_logits = [self.classifier(r).to(f.device) for r, f in zip(reloi_feats, feats)]
            if duration_proposals is not None:
                duration_proposals = [TemporalProposals(*(t.to(f.device) if t is not None else None for t in d))
                                      for d, f in zip(duration_proposals, feats)]
            return pair_proposals, duration_proposals, rel_logits

    def _forward_test_fused(self, pair_list):
        """Temporal path on tracklet tensors: one fused HIP pass per group of equal-shape segments."""
        dpn = self.relpn.duration_proposal_network
        cls = self.classifier.rel_predictor
        first = pair_list[0].get_field("tracklet_feats")
        dev = _compute_device(first, dpn.dpn_head.conv.weight)
        d_feat = first.shape[2]
        if 2 * d_feat != dpn.dpn_head.in_channels:
            raise ValueError(f"tracklet_feats dim D={d_feat} needs RELPN.DPN.IN_CHANNELS = 2*D "
                             f"(got {dpn.dpn_head.in_channels})")
        if cls.in_features != 2 * d_feat:
            raise ValueError(f"PREDICT.FEATURE_DIM must equal 2*D = {2 * d_feat} on the temporal path "
                             f"(got {cls.in_features})")
        hw, hb = dpn._head_weights(dev)
        cw, cb = self.classifier._cache.get("cls", (cls.weight, cls.bias), dev, lambda ts: ts)

        main = torch.cuda.current_stream(dev)
        overlap = self.overlap_tail and all(p.get_field("tracklet_feats").is_cuda for p in pair_list)
        side, ev_logits = self._side_stream(dev) if overlap else (None, None)
        if side is not None:
            # everything the side stream reads that THIS call may build lazily is built here, on the caller's stream,
            # before the side stream is made to wait for it: the PPN weight copies and the canonical pair tables of
            # every group (read by the pair-geometry launch here and by `decode` later)
            if self.use_ppn:
                self.relpn.pair_proposal_network.device_weights(dev)
            shapes = {}
            for p in pair_list:
                f = p.get_field("tracklet_feats")
                if not (p.has_field("tracklet_pairs") and p.get_field("tracklet_pairs") is not None):
                    k = (tuple(f.shape), f.dtype == torch.bfloat16)
                    shapes[k] = shapes.get(k, 0) + 1
            for (shape, _), cnt in shapes.items():
                self._canonical_pairs(dev, cnt, shape[0])
            side.wait_stream(main)                 # class logits / boxes may have been produced on the caller's stream
        pair_proposals = None
        if self.use_ppn:
            if overlap and all(p.get_field("track_cls_logits").is_cuda for p in pair_list):
                with torch.cuda.stream(side):
                    pair_proposals, _ = self.relpn.pair_proposal_network(pair_list)
                    for t in pair_proposals:
                        t.record_stream(main)      # allocated under the side stream, consumed on the caller's
            else:
                pair_proposals, _ = self.relpn.pair_proposal_network(pair_list)
        self._logits_token = None
        join_side = False

        n_seg = len(pair_list)
        durations, logits = [None] * n_seg, [None] * n_seg
        def custom_pairs(plist):
            if not plist.has_field("tracklet_pairs"):
                return None
            p = plist.get_field("tracklet_pairs")
            if p is None:
                return None
            if isinstance(p, torch.Tensor):
                return p.detach().cpu().long()
            return torch.as_tensor(np.asarray(p)).long()

        groups = {}
        for i, plist in enumerate(pair_list):
            f = plist.get_field("tracklet_feats")
            # segments with an explicit pair table are scored on their own
            key = (tuple(f.shape), i + 1 if custom_pairs(plist) is not None else 0, f.dtype == torch.bfloat16)
            groups.setdefault(key, []).append(i)
        for (shape, custom_key, bf16), members in groups.items():
            n, t, d = shape
            nm = len(members)
            per = n * (n - 1)
            src = [pair_list[i].get_field("tracklet_feats") for i in members]
            if (self.host_chunk_videos > 0 and custom_key == 0 and not self.pool_top_span and per > 0
                    and (d % 16 == 0 if bf16 else True)
                    and all(isinstance(x, torch.Tensor) and not x.is_cuda for x in src)):
                # host-resident features, canonical pair table: chunked upload / compute / download pipeline
                self._forward_host_group(pair_list, members, shape, bf16, dev, (hw, hb, cw, cb), durations, logits,
                                         track_token=len(groups) == 1)
                continue
            if bf16:
                # bf16 tracklet features select the bf16-operand kernels (BASELINE config 3)
                if any(custom_pairs(pair_list[i]) is not None for i in members):
                    raise NotImplementedError("the bf16 path scores the canonical pair table only")
                if d % 16:
                    raise ValueError(f"the bf16 path needs D % 16 == 0 (D={d})")
                feats = _batch_rows(src, dev, torch.bfloat16)
                allp, _ = self._canonical_pairs(dev, nm, n)
                packed, cbias, hpk, hb16 = dpn._bf16_weights(dev)
                cw16, cb16 = self.classifier._cache.get(
                    "cls_bf16", (cls.weight, cls.bias), dev,
                    lambda ts: tuple(ops.cast_bf16(x.contiguous()).float() for x in ts))
                need = ops.fused_bf16_workspace_bytes(nm, n, t, d, hb16.numel() // 3, cw16.shape[0], allp.shape[0])
                heads, lg = ops.forward_fused_bf16(feats, allp, nm, n, packed, cbias, hpk, hb16, cw16, cb16,
                                                   workspace=self._workspace(dev, need), conv_events=self._conv_events,
                                                   logits_event=ev_logits if (overlap and len(groups) == 1) else None)
                counts = [per] * nm
            else:
                feats = _batch_rows(src, dev)
                customs = [custom_pairs(pair_list[i]) for i in members]
                canonical = all(p is None for p in customs)
                if canonical:
                    allp, _ = self._canonical_pairs(dev, nm, n)
                    counts = [per] * nm
                else:
                    pairs = []
                    for k, p in enumerate(customs):
                        if p is None:
                            pairs.append(ops.pair_index(n, dev, base=k * n))
                            continue
                        if p.dim() != 2 or p.shape[1] != 2:
                            raise ValueError("tracklet_pairs must be [P,2]")
                        if p.numel() and (int(p.min()) < 0 or int(p.max()) >= n):
                            raise IndexError("tracklet_pairs index out of range")
                        pairs.append(p.to(dev) + k * n)
                    counts = [p.shape[0] for p in pairs]
                    allp = torch.cat(pairs).contiguous()
                    if side is not None and self.pair_geometry_in_forward:
                        side.wait_stream(main)     # the geometry launch on the side stream reads this table
                # temporal conv algorithm: RELPN.DPN.CONV_ALGO = "auto" (Winograd F(6,3) when D % 32 == 0: 4/9 of the
                # MFMA work; its fp32 error bound is in DESIGN.md §4) or "direct" (the k=3 taps as one implicit GEMM)
                wino = self._winograd(d, dev)
                packed, cbias = dpn._conv_split(dev, winograd=wino)
                craw, crows = self._conv_guard(dpn, dev, wino)
                need = ops.fused_workspace_bytes(nm, n, t, d, hb.numel() // 3, cw.shape[0], allp.shape[0])
                heads, lg = ops.forward_fused(feats, allp, nm, n, packed, cbias, hw, hb, cw, cb,
                                              workspace=self._workspace(dev, need), check_pairs=False,
                                              canonical_pairs=canonical, conv_events=self._conv_events,
                                              conv_weight=craw, conv_check=crows,
                                              logits_event=ev_logits if (overlap and len(groups) == 1) else None)
                if self.pool_top_span and allp.shape[0]:
                    # RelOIPool over each pair's best span (decode + NMS, top-1) instead of the whole segment
                    top = ops.decode_spans(heads, self.anchor_sizes(t), top_k=1)["span"][:, 0].contiguous()
                    lg = ops.span_predicate(feats, allp, top, cw, cb)
            if overlap and len(groups) == 1 and not (self.pool_top_span and not bf16):
                # `decode` may start behind the logits-ready event of THIS pass (same logits tensor, same caller stream)
                # (the token keeps `lg` alive: its address cannot be handed to another tensor while the token stands)
                self._logits_token = (dev.index, lg, main.cuda_stream)
            geom = None
            if self.pair_geometry_in_forward:
                boxes_on_dev = all(pair_list[i].has_field("tracklet_boxes") and isinstance(pair_list[i].get_field("tracklet_boxes"), torch.Tensor)
                                   and pair_list[i].get_field("tracklet_boxes").is_cuda for i in members)
                if side is not None and boxes_on_dev:
                    # the bbox half of the pair builder needs the boxes only: second stream, under the encoder
                    with torch.cuda.stream(side):
                        geom = self._pair_geometry_batch(pair_list, members, allp, dev)
                        if geom is not None:
                            geom.record_stream(main)
                    join_side = True
                else:
                    geom = self._pair_geometry_batch(pair_list, members, allp, dev)
            # per-segment results are VIEWS of the batched outputs (copies only when a segment's inputs live on
            # another device, e.g. the host tensors predict.py hands over)
            off = 0
            for k, i in enumerate(members):
                src_dev = src[k].device
                sl = slice(off, off + counts[k])
                durations[i] = dpn._wrap(heads[sl].to(src_dev), None if geom is None else geom[sl].to(src_dev))
                logits[i] = lg[sl].to(src_dev)
                off += counts[k]
        if side is not None and (self.use_ppn or join_side):
            main.wait_stream(side)                 # PPN / geometry are complete for whatever the caller does next
        return pair_proposals, durations, logits

    @staticmethod
    def _host_chunk_schedule(nm, chunk):
        """Video ranges the host-input pipeline scores one after the other (bench.py reads the last one)."""
        return _HostPipeline.schedule(nm, chunk)

    def _forward_host_group(self, pair_list, members, shape, bf16, dev, weights, durations, logits, track_token):
        """One group of equal-shape segments whose `tracklet_feats` are HOST tensors (the reference's predict.py:50-57
        hands CPU PairLists): upload, fused pass and download run as a three-stage pipeline over chunks of videos
        (_HostPipeline); results are pinned host tensors, complete when this returns (same device as the inputs,
        reference model.py:53-65).  Same kernels, same per-video results as the resident path."""
        dpn = self.relpn.duration_proposal_network
        cls = self.classifier.rel_predictor
        hw, hb, cw, cb = weights
        n, t, d = shape
        nm, per = len(members), n * (n - 1)
        src = [pair_list[i].get_field("tracklet_feats") for i in members]
        pipe = self._host_pipes.get(dev.index)
        if pipe is None:
            pipe = self._host_pipes[dev.index] = _HostPipeline(dev)
        main = torch.cuda.current_stream(dev)
        if bf16:
            packed, cbias, hpk, hb16 = dpn._bf16_weights(dev)
            cw16, cb16 = self.classifier._cache.get(
                "cls_bf16", (cls.weight, cls.bias), dev,
                lambda ts: tuple(ops.cast_bf16(x.contiguous()).float() for x in ts))
            a3, k_out = hb16.numel(), cw16.shape[0]
        else:
            wino = self._winograd(d, dev)
            packed, cbias = dpn._conv_split(dev, winograd=wino)
            craw, crows = self._conv_guard(dpn, dev, wino)
            a3, k_out = hb.numel(), cw.shape[0]
        chunks = pipe.schedule(nm, self.host_chunk_videos)
        feats = pipe.begin(src, torch.bfloat16 if bf16 else torch.float32)
        heads_dev = torch.empty((nm * per, a3, t), dtype=torch.float32, device=dev)
        lg_dev = torch.empty((nm * per, k_out), dtype=torch.float32, device=dev)
        heads_host = torch.empty((nm * per, a3, t), dtype=torch.float32, pin_memory=True)
        lg_host = torch.empty((nm * per, k_out), dtype=torch.float32, pin_memory=True)
        want_geom = self.pair_geometry_in_forward and all(
            pair_list[i].has_field("tracklet_boxes") and pair_list[i].get_field("tracklet_boxes") is not None for i in members)
        geom_host = torch.empty((nm * per, 8, t), dtype=torch.float32, pin_memory=True) if want_geom else None
        # like heads_dev / lg_dev: lives until finish() has synchronised the download stream (a per-chunk temporary
        # would go back to the caller stream's pool while d2h still reads it)
        geom_dev = torch.empty((nm * per, 8, t), dtype=torch.float32, device=dev) if want_geom else None
        cmax = max(hi - lo for lo, hi in chunks)
        wsb = ops.fused_bf16_workspace_bytes if bf16 else ops.fused_workspace_bytes
        ws = self._workspace(dev, wsb(cmax, n, t, d, a3 // 3, k_out, cmax * per))
        # the small side inputs go up NOW, while the caller's stream is empty: a blocking .to(device) issued after a
        # pass has been queued would wait for that pass and stall the pipeline
        boxes_dev = _batch_rows([pair_list[i].get_field("tracklet_boxes") for i in members], dev) if want_geom else None
        ready = pipe.stage(*chunks[0])
        for k, (lo, hi) in enumerate(chunks):
            c = hi - lo
            main.wait_event(ready)
            allp, _ = self._canonical_pairs(dev, c, n)
            rows = slice(lo * per, hi * per)
            if bf16:
                ops.forward_fused_bf16(feats[lo * n:hi * n], allp, c, n, packed, cbias, hpk, hb16, cw16, cb16, workspace=ws,
                                       conv_events=self._conv_events if k == len(chunks) - 1 else None,
                                       out_heads=heads_dev[rows], out_logits=lg_dev[rows])
            else:
                ops.forward_fused(feats[lo * n:hi * n], allp, c, n, packed, cbias, hw, hb, cw, cb, workspace=ws,
                                  check_pairs=False, canonical_pairs=True, conv_weight=craw, conv_check=crows,
                                  conv_events=self._conv_events if k == len(chunks) - 1 else None,
                                  out_heads=heads_dev[rows], out_logits=lg_dev[rows])
            back = [(heads_host[rows], heads_dev[rows]), (lg_host[rows], lg_dev[rows])]
            if want_geom:
                ops.pair_gather(None, boxes_dev[lo * n:hi * n], allp, want_feat=False, check_pairs=False,
                                out_geom=geom_dev[rows])
                back.append((geom_host[rows], geom_dev[rows]))
            pipe.download(main, back)
            if k + 1 < len(chunks):
                ready = pipe.stage(*chunks[k + 1])         # host staging of chunk k+1 runs while the GPU works on chunk k
        pipe.finish(main)
        for k, i in enumerate(members):
            sl = slice(k * per, (k + 1) * per)
            durations[i] = dpn._wrap(heads_host[sl], None if geom_host is None else geom_host[sl])
            logits[i] = lg_host[sl]
        if track_token:
            # `decode` on these very host tensors reads the device copy instead of uploading them again
            self._logits_token = (dev.index, lg_dev, main.cuda_stream,
                                  tuple((logits[i].data_ptr(), logits[i]._version) for i in members))

    @staticmethod
    def _pair_geometry_batch(pair_list, members, allp, dev):
        """The bbox half of the N^2 pair builder for one group of equal-shape segments, ONE launch: relative
        box geometry [P_total, 8, T] of every scored pair (pair_geometry_kernel: one lane per (pair, frame),
        the motion channels through a wavefront shuffle).  None unless every segment carries 'tracklet_boxes'."""
        if not all(pair_list[i].has_field("tracklet_boxes") and pair_list[i].get_field("tracklet_boxes") is not None
                   for i in members) or allp.shape[0] == 0:
            return None
        boxes = _batch_rows([pair_list[i].get_field("tracklet_boxes") for i in members], dev)
        _, geom = ops.pair_gather(None, boxes, allp, want_feat=False, check_pairs=False)   # allp was validated above
        return geom

    def decode(self, pair_list, rel_logits, topk_per_pair=20, topk_per_seg=200, num_obj=35, overlap=True):
        """Top-k triplet decode of `forward`'s rel_logits on the GPU (replaces the Python of
        reference lib/modeling/predict.py:59-117).  Per segment returns
        (scores [M], triplets int64 [M,3] = (subject class, predicate, object class), pair_tids [M,2]).

        Baseline segments (2-D `features` [P,F>=70]) reproduce predict.py:88-89 as is: class =
        argmax of feature row (N-1)*tid, columns 0:35 / 35:70.  Tracklet segments use
        'track_cls_logits' [N,35] directly.  Segments with < 2 tracklets yield empty results
        (predict.py:61-64 skips them).

        Stream contract of the overlapped tail (`RELPN.OVERLAP_TAIL`): when `rel_logits` are the very tensors the last
        `forward` on this stream returned, the decode runs on the module's side stream behind that forward's
        logits-ready event, i.e. under its encoder.  The side stream was ordered after the caller's stream at the START
        of that forward, so `pair_list`'s fields must be the ones that forward was given; class logits (re)written on
        the caller's stream BETWEEN forward and decode are not waited for — pass `overlap=False` (or set
        `model.overlap_tail = False`) for such a decode."""
        out = [None] * len(pair_list)
        groups = {}
        for i, (plist, lg) in enumerate(zip(pair_list, rel_logits)):
            n = int(plist.get_field("num_tracklets")) if plist.has_field("num_tracklets") else None
            quirk = plist.features.dim() == 2 and plist.features.shape[1] >= 2 * num_obj
            if n is None:
                n = int(plist.get_field("track_cls_logits").shape[0])
            if n <= 1 or lg.shape[0] == 0:
                dev = lg.device
                out[i] = (torch.empty(0, device=dev), torch.empty((0, 3), dtype=torch.int64, device=dev),
                          torch.empty((0, 2), dtype=torch.int64, device=dev))
                continue
            groups.setdefault((n, tuple(lg.shape), quirk, tuple(plist.features.shape) if quirk else None), []).append(i)
        for (n, lshape, quirk, _), members in groups.items():
            dev = _compute_device(*[rel_logits[i] for i in members])
            nm = len(members)
            tok = self._logits_token
            if (tok is not None and len(tok) == 4 and tok[0] == dev.index and tok[1].numel() == nm * lshape[0] * lshape[1]
                    and tok[2] == torch.cuda.current_stream(dev).cuda_stream
                    and tok[3] == tuple((rel_logits[i].data_ptr(), rel_logits[i]._version) for i in members)):
                # the host tensors the last forward handed out, untouched: their device copy is still there
                lg = tok[1].view(nm, lshape[0], lshape[1])
            else:
                lg = _batch_rows([rel_logits[i] for i in members], dev).view(nm, lshape[0], lshape[1])
            # these are the logits of the last fused forward, still on the stream that produced them: decode on the
            # side stream behind their ready-event, i.e. UNDER that forward's encoder, and join afterwards
            main = torch.cuda.current_stream(dev)
            custom = [pair_list[i].has_field("tracklet_pairs") and pair_list[i].get_field("tracklet_pairs") is not None
                      for i in members]
            side, tok = None, self._logits_token
            if (overlap and self.overlap_tail and tok is not None and len(tok) == 3 and tok[0] == dev.index and tok[2] == main.cuda_stream
                    and tok[1].data_ptr() == lg.data_ptr() and tok[1].numel() == lg.numel() and not quirk
                    and not any(custom) and all(pair_list[i].get_field("track_cls_logits").is_cuda for i in members)):
                side, ev_logits = self._side_stream(dev)
                side.wait_event(ev_logits)
                self._logits_token = None          # one decode per forward rides the event
            if not any(custom):
                for i in members:
                    if n * (n - 1) != rel_logits[i].shape[0]:
                        raise ValueError(f"decode: segment {i}: rel_logits has {rel_logits[i].shape[0]} rows but "
                                         f"{n} tracklets give {n * (n - 1)} pairs and no 'tracklet_pairs' field is set")
                pairs = self._canonical_pairs(dev, nm, n)[1]
                trusted = True
            else:
                pairs = []
                for i in members:
                    plist = pair_list[i]
                    if plist.has_field("tracklet_pairs") and plist.get_field("tracklet_pairs") is not None:
                        p = plist.get_field("tracklet_pairs")
                        p = p.detach().long() if isinstance(p, torch.Tensor) else torch.as_tensor(np.asarray(p)).long()
                        if tuple(p.shape) != (rel_logits[i].shape[0], 2):
                            raise ValueError(f"decode: segment {i}: 'tracklet_pairs' must be [P,2] with one row per "
                                             f"rel_logits row (P={rel_logits[i].shape[0]}), got {tuple(p.shape)}")
                        pairs.append(p.to(dev))
                    else:
                        if n * (n - 1) != rel_logits[i].shape[0]:
                            raise ValueError(f"decode: segment {i}: rel_logits has {rel_logits[i].shape[0]} rows but "
                                             f"{n} tracklets give {n * (n - 1)} pairs and no 'tracklet_pairs' field is set")
                        pairs.append(ops.pair_index(n, dev))
                pairs = torch.stack(pairs).contiguous()
                trusted = False
            if quirk:
                fshape = tuple(pair_list[members[0]].features.shape)
                cls = _batch_rows([pair_list[i].features for i in members], dev).view((nm,) + fshape)
                res = ops.decode_topk(lg, pairs, cls, row_mul=n - 1, num_obj=num_obj,
                                      topk_per_pair=topk_per_pair, topk_per_seg=topk_per_seg, check_pairs=not trusted)
            else:
                cshape = tuple(pair_list[members[0]].get_field("track_cls_logits").shape)
                cls_src = [pair_list[i].get_field("track_cls_logits") for i in members]
                if side is not None:
                    with torch.cuda.stream(side):
                        # the class logits are batched UNDER the side stream: when they are not consecutive slices of
                        # one allocation this is a cat / cast kernel, which on the caller's stream would sit behind the
                        # whole encoder while the decode launch on the side stream read its (unwritten) result
                        cls = _batch_rows(cls_src, dev).view((nm,) + cshape)
                        res = ops.decode_topk(lg, pairs, cls, row_mul=1, num_obj=num_obj, topk_per_pair=topk_per_pair,
                                              topk_per_seg=topk_per_seg, check_pairs=not trusted)
                        for r in res:
                            r.record_stream(main)      # allocated under the side stream, consumed on the caller's
                    main.wait_stream(side)             # the caller's stream sees complete results from here on
                else:
                    cls = _batch_rows(cls_src, dev).view((nm,) + cshape)
                    res = ops.decode_topk(lg, pairs, cls, row_mul=1, num_obj=num_obj, topk_per_pair=topk_per_pair,
                                          topk_per_seg=topk_per_seg, check_pairs=not trusted)
            for k, i in enumerate(members):
                tgt = rel_logits[i].device
                out[i] = tuple(r[k].to(tgt) for r in res)
        return out

    def anchor_sizes(self, num_frames):
        """Anchor widths (frames) of the A anchors per location.  cfg.RELPN.DPN.ANCHOR_SIZES when it
        is a sequence of length A; the reference ships a placeholder int there (defaults.py:66), so
        the default is the proportions of its own example (anchor_generator.py:116-123:
        sizes (15,30,45,60) on T=60): size_a = (a+1)*T/A."""
        a = self.relpn.duration_proposal_network.dpn_head.num_windows
        cfg_sizes = getattr(self, "_anchor_sizes_cfg", None)
        if isinstance(cfg_sizes, (list, tuple)) and len(cfg_sizes) == a:
            return [float(v) for v in cfg_sizes]
        return [(i + 1) * float(num_frames) / a for i in range(a)]

    def decode_spans(self, duration_proposals, sizes=None, top_k=None, nms_threshold=0.5):
        """Temporal span proposals per pair from `forward`'s duration_proposals (span decode +
        temporal NMS on the GPU; completes the reference's stub RelNMS, rel_nms.py:5-15).
        Per segment a dict: anchor [P,k], span int64 [P,k,2] (frames [s,e)), span_f, score, count."""
        top_k = top_k or self.relpn.duration_proposal_network.top_k_proposals
        out = []
        for dp in duration_proposals:
            heads = dp.heads
            dev = _compute_device(heads)
            sz = sizes if sizes is not None else self.anchor_sizes(heads.shape[2])
            res = ops.decode_spans(_f32(heads, dev), sz, top_k=top_k, nms_threshold=nms_threshold)
            out.append({k: v.to(heads.device) for k, v in res.items()})
        return out

    def classify_spans(self, pair_list, spans):
        """Predicate logits with RelOIPool restricted to given spans: per segment `spans[i]` int64 [P,2]
        frames [start,end) (e.g. decode_spans(...)[i]["span"][:, j]); the build-defined meaning of
        RelOIPool.__call__(feats, duration_proposals) (reference model.py:68-73) + RelationPredictor."""
        cls = self.classifier.rel_predictor
        out = []
        with torch.no_grad():
            for plist, sp in zip(pair_list, spans):
                f = plist.get_field("tracklet_feats")
                dev = _compute_device(f, cls.weight)
                cw, cb = self.classifier._cache.get("cls", (cls.weight, cls.bias), dev, lambda ts: ts)
                n = f.shape[0]
                if plist.has_field("tracklet_pairs") and plist.get_field("tracklet_pairs") is not None:
                    p = plist.get_field("tracklet_pairs")
                    p = (p.detach() if isinstance(p, torch.Tensor) else torch.as_tensor(np.asarray(p))).long().to(dev)
                else:
                    p = ops.pair_index(n, dev)
                sp = (sp if isinstance(sp, torch.Tensor) else torch.as_tensor(np.asarray(sp))).long().to(dev)
                out.append(ops.span_predicate(_f32(f, dev), p.contiguous(), sp.contiguous(), cw, cb).to(f.device))
        return out

    def pair_geometry(self, pair_list):
        """Relative box geometry [P,8,T] per segment from 'tracklet_boxes' (pair builder side output)."""
        out = []
        for plist in pair_list:
            boxes = plist.get_field("tracklet_boxes")
            dev = _compute_device(boxes)
            n = boxes.shape[0]
            _, g = ops.pair_gather(None, _f32(boxes, dev), ops.pair_index(n, dev), want_feat=False)
            out.append(g.to(boxes.device))
        return out
