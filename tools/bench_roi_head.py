#!/usr/bin/env python3
"""Micro-benchmark of the RoI feature head (first slice) at the VidVRD shape:
N=32 tracklets x T=150 frames = 4800 RoIs per video on a res4 map of a 720p frame (45 x 80 x 1024)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--t", type=int, default=150)
ap.add_argument("--chunk", type=int, default=2400)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--bf16", action="store_true", help="bf16 feature maps: bf16 MFMA kernels")
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
head = tspn.Res5RoIHead(roi_chunk=args.chunk).to(dev)
fm = torch.rand((args.t, 45, 80, 1024), device=dev, generator=g)
if args.bf16:
    fm = fm.to(torch.bfloat16)
peak = 2500.0 if args.bf16 else 157.3
xy = torch.rand((args.n, args.t, 2), device=dev, generator=g) * torch.tensor([900.0, 400.0], device=dev)
wh = 40 + torch.rand((args.n, args.t, 2), device=dev, generator=g) * 260
boxes = torch.cat([xy, xy + wh], dim=2).contiguous()
R = args.n * args.t
# flop per RoI: block 0 (1x1 s2 1024->512, 3x3 512, 1x1 512->2048, shortcut 1024->2048) + 2 x (2048->512, 3x3, 512->2048), 7x7
f0 = 49 * 2 * (1024 * 512 + 9 * 512 * 512 + 512 * 2048 + 1024 * 2048)
f1 = 49 * 2 * (2048 * 512 + 9 * 512 * 512 + 512 * 2048)
flop = R * (f0 + 2 * f1)
head(fm, boxes)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
for a, b in evs:
    a.record()
    out = head(fm, boxes)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
print(f"res5 RoI head [{'bf16' if args.bf16 else 'fp32'}]: {R} RoIs ({args.n} x {args.t}) in {ms:.1f} ms -> {R / ms * 1e3:.0f} RoIs/s, "
      f"{flop / ms / 1e9:.1f} TFLOP/s ({flop / ms / 1e9 / peak * 100:.1f} % of the {'bf16' if args.bf16 else 'fp32'} MFMA peak), "
      f"{flop / R / 1e9:.2f} GFLOP per RoI; out {tuple(out.shape)}", flush=True)
# single conv shapes of the head
for (name, nb, h, cin, cout, k, s, p) in [("1x1 1024->512 s2", args.chunk, 14, 1024, 512, 1, 2, 0), ("3x3 512->512", args.chunk, 7, 512, 512, 3, 1, 1),
                                          ("1x1 512->2048", args.chunk, 7, 512, 2048, 1, 1, 0), ("1x1 2048->512", args.chunk, 7, 2048, 512, 1, 1, 0)]:
    x = torch.rand((nb, h, h, cin), device=dev, generator=g)
    wt = (torch.rand((cout, cin, k, k), device=dev, generator=g) - 0.5) * 0.05
    line = f"  conv {name}:"
    if args.bf16:
        x = x.to(torch.bfloat16)
        variants = (("bf16 MFMA", tspn.ops.pack_conv2d_frag_bf16(wt), tspn.ops.conv2d_nhwc_bf16),)
    else:
        variants = (("registers-direct weights", tspn.ops.pack_conv2d_frag(wt), tspn.ops.conv2d_nhwc),
                    ("weights through LDS", tspn.ops.pack_conv2d(wt), tspn.ops.conv2d_nhwc))
    for label, w, fn in variants:
        for _ in range(2):
            y = fn(x, w, (k, k), s, p, relu=True)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in evs:
            a.record()
            y = fn(x, w, (k, k), s, p, relu=True)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
        fl = 2.0 * y.numel() * cin * k * k
        line += f" {label} {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / peak * 100:.1f} %);"
    print(line, flush=True)
