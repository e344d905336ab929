#!/usr/bin/env python3
"""Stem of the bf16 backbone on 720p frames: conv launch + pool launch against the fused kernel.
    python tools/time_stem.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((frames, 720, 1280, 3), device=dev, generator=g) - 0.5
w = (torch.rand((64, 3, 7, 7), device=dev, generator=g) - 0.5) * 0.2
b = torch.zeros(64, device=dev)
frag = tspn.ops.pack_stem_bf16(w)
arms = {"conv": lambda: tspn.ops.stem_conv_bf16(x, frag, b),
        "conv+pool": lambda: tspn.ops.max_pool_nhwc_bf16(tspn.ops.stem_conv_bf16(x, frag, b), 3, 2, 1),
        "fused": lambda: tspn.ops.stem_pool_bf16(x, frag, b)}
assert torch.equal(arms["conv+pool"](), arms["fused"]())
for name, fn in arms.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    print(f"{name}: {sorted(ts)[4]:.1f} us for {frames} frames (space-to-depth pass included)", flush=True)
