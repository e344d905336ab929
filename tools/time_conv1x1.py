#!/usr/bin/env python3
"""1x1 convolutions of the backbone / RoI head on the generic bf16 kernel, HIP events, median of 11 launches with the cache state of
the chain (the input is rewritten by a copy kernel before every launch, as the previous layer would have left it).

    python tools/time_conv1x1.py [frames ...]        (TSPN_LIB_PATH selects a probe build, tools/build_variant.sh)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
LAYERS = (("res4 conv1", 45, 80, 1024, 256, False), ("res3 conv1", 90, 160, 512, 128, False),
          ("res4.0 shortcut", 45, 80, 512, 1024, False))


def med(fn, prep, n=11):
    ts = []
    for _ in range(n + 2):
        prep()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    return sorted(ts[2:])[n // 2]


frames_list = [int(a) for a in sys.argv[1:]] or [9, 18, 36]
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "shipped"))
for name, h, w_, cin, cout, res in LAYERS:
    for frames in frames_list:
        src = torch.rand((frames, h, w_, cin), device=dev, generator=g).to(torch.bfloat16)
        x = torch.empty_like(src)
        w = (torch.rand((cout, cin, 1, 1), device=dev, generator=g) - 0.5) * 0.05
        b = torch.rand(cout, device=dev, generator=g) - 0.5
        f = tspn.ops.pack_conv2d_frag_bf16(w)
        fn = lambda: tspn.ops.conv2d_nhwc_bf16(x, f, (1, 1), 1, 0, bias=b, relu=True)
        t = med(fn, lambda: x.copy_(src))
        fl = 2.0 * frames * h * w_ * cin * cout
        by = frames * h * w_ * (cin + cout) * 2
        print(f"[{tag}] {name} {cin}->{cout} {frames} frames: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s  {by / t / 1e6:5.2f} TB/s (x + out)", flush=True)
