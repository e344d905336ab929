#!/usr/bin/env python3
"""Where the bf16 ResNet-101-C4 backbone spends its time, launch by launch on ONE stream (no overlap between launches):
every ops call of a forward of `--frames` frames is bracketed by HIP events and summed by (op, shapes).  Bytes = the
tensors a launch has to read and write once (algorithmic), so GB/s says how far a memory-bound layer is from HBM."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
from tspn_mi355x import ops

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=9)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = tspn.ResNetC4(depth=101, frame_chunk=args.frames).to(dev)
net.streams = 1
img = torch.rand((args.frames, 720, 1280, 3), device=dev) - 0.5
net(img, bf16=True)
torch.cuda.synchronize()
rec = []


def wrap(name, meta):
    real = getattr(ops, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real(*a, **k)
        e1.record()
        rec.append((name,) + meta(y, *a, **k) + (e0, e1))
        return y
    setattr(ops, name, f)


def meta_conv(y, x, frag, ks, stride=1, padding=0, bias=None, residual=None, relu=False):
    cout = y.shape[-1]
    fl = 2.0 * y.numel() * x.shape[-1] * ks[0] * ks[1]
    by = 2 * (x.numel() / (stride * stride if ks[0] == 1 else 1) + y.numel() * (2 if residual is not None else 1))
    return (f"{x.shape[1]}x{x.shape[2]} {x.shape[3]}->{cout} k{ks[0]} s{stride}{' +res' if residual is not None else ''}", fl, by)


def meta_tail(y, h1, f2, b2, f3, b3, res, **k):
    y0 = y[0] if isinstance(y, tuple) else y
    cm = h1.shape[-1]
    return (f"{h1.shape[1]}x{h1.shape[2]} tail CM={cm}", 2.0 * h1.numel() * cm * 13, 2 * (h1.numel() + 2 * y0.numel()))


def meta_stem(y, x, *a, **k):
    return (f"{x.shape[1]}x{x.shape[2]} stem+pool", 2.0 * y.numel() * 4 * 147, 4 * x.numel() + 2 * y.numel())


wrap("conv2d_nhwc_bf16", meta_conv)
wrap("bottleneck_tail_bf16", meta_tail)
wrap("stem_pool_bf16", meta_stem)
for _ in range(args.reps):
    net(img, bf16=True)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, fl, by, e0, e1 in rec:
    e = agg.setdefault((name, key), [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += e0.elapsed_time(e1) * 1e3; e[2] += fl; e[3] += by
tot = sum(e[1] for e in agg.values()) / args.reps
print(f"{args.frames} frames of 720p, one stream: {tot / 1e3:.2f} ms of launches per forward = {tot / args.frames:.1f} us per frame")
for (name, key), (n, us, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {key:34s} x{n // args.reps:3d}  {us / n:7.1f} us each  {us / args.reps / tot * 100:5.1f} %  {fl / us / 1e6:6.0f} TFLOP/s  "
          f"{by / us / 1e6:5.2f} TB/s")
