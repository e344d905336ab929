#!/usr/bin/env python3
"""End-to-end on one MI355X at the VidVRD shape (BASELINE configs[4], single GPU, ROI head onwards):
res4 feature maps of T frames + N tracklet boxes -> Res5RoIHead -> pair builder / temporal encoder / heads
(BaseModel.forward, fused path) -> top-k triplet decode.  bf16 by default (bf16 maps select the bf16 MFMA
kernels of both stages); --fp32 for the fp32 path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--t", type=int, default=150)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--pixels", action="store_true", help="start from 720p frames: run the ResNet-101-C4 backbone too")
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
D = 2048
cfg = tspn.load_cfg(None, **{"RELPN.USE_PPN": True, "RELPN.USE_DPN": True, "RELPN.DPN.IN_CHANNELS": 2 * D,
                             "PREDICT.FEATURE_DIM": 2 * D})
model = tspn.BaseModel(cfg).to(dev).eval()
head = tspn.Res5RoIHead().to(dev)
fm = torch.rand((args.t, 45, 80, 1024), device=dev, generator=g)
if not args.fp32:
    fm = fm.to(torch.bfloat16)
net, img = None, None
if args.pixels:
    net = tspn.ResNetC4(depth=101, frame_chunk=16).to(dev)
    img = torch.rand((args.t, 720, 1280, 3), device=dev, generator=g) - 0.5
xy = torch.rand((args.n, args.t, 2), device=dev, generator=g) * torch.tensor([900.0, 400.0], device=dev)
wh = 40 + torch.rand((args.n, args.t, 2), device=dev, generator=g) * 260
boxes = torch.cat([xy, xy + wh], dim=2).contiguous()
cls = torch.rand((args.n, 35), device=dev, generator=g)


def maps():
    return net(img, bf16=not args.fp32) if args.pixels else fm


def run():
    feats = head(maps(), boxes)
    plist = tspn.PairList.from_tracklets(feats, boxes, cls)
    with torch.no_grad():
        pp, dur, logits = model([plist], None)
    return model.decode([plist], logits)[0], feats


run()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
        torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
for a0, a, m, b in evs:
    a0.record()
    cur = maps()
    a.record()
    feats = head(cur, boxes)
    m.record()
    plist = tspn.PairList.from_tracklets(feats, boxes, cls)
    with torch.no_grad():
        pp, dur, logits = model([plist], None)
    trip = model.decode([plist], logits)[0]
    b.record()
torch.cuda.synchronize()
tot = sorted(a0.elapsed_time(b) for a0, a, m, b in evs)[len(evs) // 2]
bb = sorted(a0.elapsed_time(a) for a0, a, m, b in evs)[len(evs) // 2]
roi = sorted(a.elapsed_time(m) for a0, a, m, b in evs)[len(evs) // 2]
P = args.n * (args.n - 1)
print(f"end to end [{'fp32' if args.fp32 else 'bf16'}{', from 720p frames' if args.pixels else ', from res4 maps'}], one video "
      f"(N={args.n}, T={args.t}): {tot:.1f} ms ("
      + (f"ResNet-101-C4 backbone {bb:.1f} ms, " if args.pixels else "") +
      f"RoI head {roi:.1f} ms, scoring + decode {tot - bb - roi:.1f} ms) -> {1e3 / tot:.1f} videos/s, "
      f"{P * 1e3 / tot:.0f} tracklet-pairs/s, {args.n * args.t * 1e3 / tot:.0f} RoIs/s; "
      f"feats {tuple(feats.shape)} {feats.dtype}, {trip[0].shape[0]} triplets", flush=True)
