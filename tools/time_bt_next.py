#!/usr/bin/env python3
"""res4 block pair: [tail of block b] + [conv1 of block b + 1] as two launches against tspn_bottleneck_tail_next_bf16
(one launch), at the backbone's shape.   python tools/time_bt_next.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
CM, H, W = 256, 45, 80
h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
res = torch.rand((frames, H, W, 4 * CM), device=dev, generator=g).to(torch.bfloat16)
w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
w1 = (torch.rand((CM, 4 * CM, 1, 1), device=dev, generator=g) - 0.5) * 0.05
b2, b3, b1 = torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev), torch.zeros(CM, device=dev)
f2, f3, f1 = (tspn.ops.pack_conv2d_frag_bf16(w) for w in (w2, w3, w1))
out = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
arms = {"tail": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res),
        "conv1": lambda: tspn.ops.conv2d_nhwc_bf16(out, f1, (1, 1), 1, 0, bias=b1, relu=True),
        "tail+conv1": lambda: tspn.ops.conv2d_nhwc_bf16(tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res), f1, (1, 1), 1, 0, bias=b1, relu=True),
        "tail_next": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, next_frag1=f1, next_bias1=b1)}
t = {}
for name, fn in arms.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    t[name] = sorted(ts)[len(ts) // 2]
fl = 2.0 * frames * H * W * CM * CM * 17
print(f"res4 {frames}x{H}x{W}: tail {t['tail']:.1f} us, conv1 {t['conv1']:.1f}, back to back {t['tail+conv1']:.1f}; "
      f"one launch {t['tail_next']:.1f} us ({fl / t['tail_next'] / 1e6:.0f} TFLOP/s)", flush=True)
