#!/usr/bin/env python3
"""Time the fused bottleneck tail against the two separate conv launches at the backbone's shapes (16 frames of 720p).
    python tools/time_bt.py [frames]        (TSPN_LIB_PATH selects a probe build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
for CM, H, W in ((256, 45, 80), (128, 90, 160), (64, 180, 320)):
    h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
    res = torch.rand((frames, H, W, 4 * CM), device=dev, generator=g).to(torch.bfloat16)
    w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
    b2, b3 = torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev)
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    arms = {"fused": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res),
            "conv3x3": lambda: tspn.ops.conv2d_nhwc_bf16(h1, f2, (3, 3), 1, 1, bias=b2, relu=True)}
    h2 = arms["conv3x3"]()
    arms["expand"] = lambda: tspn.ops.conv2d_nhwc_bf16(h2, f3, (1, 1), 1, 0, bias=b3, residual=res, relu=True)
    out = {}
    for name, fn in arms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        out[name] = sorted(ts)[len(ts) // 2]
    fl = 2.0 * frames * H * W * CM * CM * 13
    print(f"[{tag}] CM={CM} {frames}x{H}x{W}: fused {out['fused']:.1f} us ({fl / out['fused'] / 1e6:.0f} TFLOP/s), "
          f"3x3 {out['conv3x3']:.1f} + expand {out['expand']:.1f} = {out['conv3x3'] + out['expand']:.1f} us", flush=True)
