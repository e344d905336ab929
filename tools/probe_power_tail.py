#!/usr/bin/env python3
"""Is the res4 tail power-limited?  The same launches on random operands and on all-zero operands (identical instruction
streams and byte counts; zeros switch far fewer wires): python tools/probe_power_tail.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 72
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
CM, H, W = 256, 45, 80
for kind in ("random", "zeros", "random"):
    z = kind == "zeros"
    mk = (lambda *s: torch.zeros(s, device=dev)) if z else (lambda *s: torch.rand(s, device=dev, generator=g) - 0.5)
    w2, w3 = mk(CM, CM, 3, 3) * 0.1, mk(4 * CM, CM, 1, 1) * 0.2
    b2, b3 = torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev)
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    h1 = (mk(frames, H, W, CM) + (0 if z else 0.5)).to(torch.bfloat16)
    res = mk(frames, H, W, 4 * CM).to(torch.bfloat16)
    out = {}
    for name, kw in (("tiles", {}), ("persistent", {"persistent": True})):
        fn = lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, **kw)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) / 20 * 1e3
    print(f"{kind:7s} operands, {frames} frames: one launch per tile {out['tiles']:.1f} us, persistent {out['persistent']:.1f} us", flush=True)
