#!/usr/bin/env python3
"""Per-layer time of the ResNet-101-C4 backbone (bf16 by default): which conv shapes cost what."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402
from tspn_mi355x import roi_head as rh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--fp32", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = tspn.ResNetC4(depth=101, frame_chunk=args.frames).to(dev)
img = torch.rand((args.frames, 720, 1280, 3), device=dev) - 0.5
net(img, bf16=not args.fp32)
torch.cuda.synchronize()
records = []
orig = rh.ConvFrozenBN.forward


def timed(self, x, residual=None, relu=False, stride=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = orig(self, x, residual=residual, relu=relu, stride=stride)
    b.record()
    records.append((tuple(x.shape), tuple(self.weight.shape), self.stride, residual is not None, a, b, y.numel(), x.dtype))
    return y


rh.ConvFrozenBN.forward = timed
net(img, bf16=not args.fp32)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for xs, ws, st, res, a, b, on, dt in records:
    key = (xs[1:3], ws, st, res)
    ms = a.elapsed_time(b)
    fl = 2.0 * on * ws[1] * ws[2] * ws[3]
    esz = 2 if dt == torch.bfloat16 else 4
    by = (xs[0] * xs[1] * xs[2] * xs[3] + on * (2 if res else 1)) * esz
    e = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
tot = sum(e[1] for e in agg.values())
print(f"{args.frames} frames, {'fp32' if args.fp32 else 'bf16'}: conv total {tot:.2f} ms")
for (hw, ws, st, res), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  in {hw[0]}x{hw[1]} w{ws} s{st}{' +res' if res else ''}: x{n} {ms:.2f} ms ({ms / tot * 100:.0f} %), "
          f"{fl / ms / 1e9:.0f} TFLOP/s, {by / ms / 1e6:.0f} GB/s")
