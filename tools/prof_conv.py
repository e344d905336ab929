#!/usr/bin/env python3
"""Run only the dominant kernel (conv3 MFMA, tracklet-projection shape of cfg2) a few times —
a small target for `rocprofv3 --pmc ...` passes and for A/B timing of kernel variants."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--cin", type=int, default=2048)
ap.add_argument("--m", type=int, default=8192)
ap.add_argument("--t", type=int, default=150)
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--tc", action="store_true", help="channels-last x (tracklet layout) kernel")
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((args.videos * args.n, args.cin, args.t), device=dev, generator=g)
conv = tspn.ops.conv3
if args.tc:
    x = x.transpose(1, 2).contiguous()
    conv = tspn.ops.conv3_tc
w = (torch.rand((3, args.cin, args.m), device=dev, generator=g) - 0.5) * 0.02
for _ in range(2):
    y = conv(x, w)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
for a, b in evs:
    a.record()
    y = conv(x, w)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)
flop = 2.0 * args.m * 3 * args.cin * x.shape[0] * args.t
shape = f"B={x.shape[0]} Cin={args.cin} T={args.t} M={args.m}" + (" [channels-last]" if args.tc else "")
print(f"conv3 {shape}: median {ms[len(ms)//2]:.3f} ms "
      f"min {ms[0]:.3f} ms -> {flop / ms[len(ms)//2] / 1e9:.1f} direct-equivalent TFLOP/s (median)")
