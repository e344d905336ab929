#!/usr/bin/env python3
"""1x1 / strided bf16 convs of the backbone and the RoI head: tspn_conv2d_nhwc_bf16 with the deep ring (default) and, with
TSPN_CONV_BF16_DEEP=0 in the environment, the two-stage form.   python tools/time_conv1x1.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
tag = "two-stage" if os.environ.get("TSPN_CONV_BF16_DEEP", "1") == "0" else "deep"
shapes = [("res4 conv1 1024->256, 9 frames", (9, 45, 80, 1024), 256, 1, False),
          ("res3 conv1 512->128, 9 frames", (9, 90, 160, 512), 128, 1, False),
          ("res2 conv1 256->64, 9 frames", (9, 180, 320, 256), 64, 1, False),
          ("res4.0 shortcut 512->1024 /2, 9 frames", (9, 90, 160, 512), 1024, 2, False),
          ("res5 conv1 2048->512, 2400 RoIs", (2400, 7, 7, 2048), 512, 1, False),
          ("res5 conv3 512->2048 + residual, 2400 RoIs", (2400, 7, 7, 512), 2048, 1, True),
          ("res5.0 conv1 1024->512, 2400 RoIs", (2400, 7, 7, 1024), 512, 1, False)]
for name, xs, cout, stride, res in shapes:
    x = torch.rand(xs, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.rand((cout, xs[3], 1, 1), device=dev, generator=g) - 0.5) * 0.05
    f = tspn.ops.pack_conv2d_frag_bf16(w)
    b = torch.zeros(cout, device=dev)
    oh, ow = (xs[1] - 1) // stride + 1, (xs[2] - 1) // stride + 1
    r = torch.rand((xs[0], oh, ow, cout), device=dev, generator=g).to(torch.bfloat16) if res else None
    fn = lambda: tspn.ops.conv2d_nhwc_bf16(x, f, (1, 1), stride, 0, bias=b, residual=r, relu=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    npix = xs[0] * oh * ow
    fl = 2.0 * npix * cout * xs[3]
    by = (xs[0] * xs[1] * xs[2] * xs[3] / (stride * stride) + npix * cout * (2 if res else 1)) * 2
    print(f"[{tag}] {name}: {us:.1f} us, {fl / us / 1e6:.0f} TFLOP/s, {by / us / 1e6:.2f} TB/s", flush=True)
