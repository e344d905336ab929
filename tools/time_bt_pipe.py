#!/usr/bin/env python3
"""res4 tail: one launch per 128-pixel tile (tspn_bottleneck_tail_bf16) against the persistent pipelined kernel
(tspn_bottleneck_tail_pipe_bf16), several frame counts.   python tools/time_bt_pipe.py [frames ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
CM, H, W = 256, 45, 80
w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
b2, b3 = torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev)
f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
for frames in [int(a) for a in sys.argv[1:]] or [9, 18, 36, 72]:
    h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
    res = torch.rand((frames, H, W, 4 * CM), device=dev, generator=g).to(torch.bfloat16)
    arms = {"tiles": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res),
            "persistent": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, persistent=True)}
    t = {}
    for name, fn in arms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t[name] = sorted(ts)[len(ts) // 2]
    fl = 2.0 * frames * H * W * CM * CM * 13
    same = torch.equal(arms["tiles"](), arms["persistent"]())
    print(f"res4 tail {frames}x{H}x{W} ({-(-frames * H * W // 128)} tiles): one launch per tile {t['tiles']:.1f} us "
          f"({fl / t['tiles'] / 1e6:.0f} TFLOP/s), persistent {t['persistent']:.1f} us ({fl / t['persistent'] / 1e6:.0f} TFLOP/s), "
          f"equal {same}", flush=True)
