#!/usr/bin/env python3
"""The three layer shapes of a res5 block of the RoI head on one chunk of RoIs (2400 x 7 x 7 pixels, bf16): conv1 2048 -> 512,
the 3x3 512 -> 512, conv3 512 -> 2048 + residual + ReLU (`conv2d_nhwc_bf16_kernel`).  Ten launches back to back per event
pair (the queue never runs dry), median of seven.  TSPN_LIB_PATH selects a probe build (tools/build_variant.sh).
    python tools/time_res5_layers.py [rois]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "shipped"))
for name, cin, cout, k, res in (("conv1 1x1 2048->512", 2048, 512, 1, False), ("conv2 3x3 512->512", 512, 512, 3, False),
                                ("conv3 1x1 512->2048 + residual", 512, 2048, 1, True)):
    x = torch.rand((R, 7, 7, cin), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.rand((cout, cin, k, k), device=dev, generator=g) - 0.5) * 0.05
    b = torch.rand(cout, device=dev, generator=g) - 0.5
    r = (torch.rand((R, 7, 7, cout), device=dev, generator=g) - 0.5).to(torch.bfloat16) if res else None
    f = tspn.ops.pack_conv2d_frag_bf16(w)
    fn = lambda: tspn.ops.conv2d_nhwc_bf16(x, f, (k, k), 1, k // 2, bias=b, residual=r, relu=True)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e2)
    t = sorted(ts)[3]
    fl = 2.0 * R * 49 * cin * cout * k * k
    by = R * 49 * (cin + cout * (2 if res else 1)) * 2
    print(f"[{tag}] {name}, {R} RoIs: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s  {by / t / 1e6:5.2f} TB/s (x + out{' + residual' if res else ''})", flush=True)
