#!/usr/bin/env python3
"""Same kernel, same instruction stream, different operand DATA: does the run time depend on the values?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
xr = torch.rand((videos * 32, 150, 2048), device=dev, generator=g)
pr = (torch.rand((6, 2048, 8192), device=dev, generator=g) - 0.5) * 0.02
fr = tspn.ops.repack_wino43_frag(pr)
cases = [("x random, w random", xr, fr), ("x zeros,  w random", torch.zeros_like(xr), fr),
         ("x random, w zeros ", xr, torch.zeros_like(fr)), ("x zeros,  w zeros ", torch.zeros_like(xr), torch.zeros_like(fr)),
         ("x ones,   w const ", torch.ones_like(xr), torch.full_like(fr, 0.01)),
         ("x random, w random", xr, fr)]
for name, x, w in cases:
    for _ in range(2):
        tspn.ops.conv3_tc_wino43r(x, w)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in evs:
        a.record()
        tspn.ops.conv3_tc_wino43r(x, w)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    flop = 0.5 * (152 / 150) * 2.0 * 8192 * 3 * 2048 * x.shape[0] * 150
    print(f"{name}: median {ms[3]:.3f} ms -> {flop / ms[3] / 1e9:.1f} TFLOP/s ({flop / ms[3] / 1e9 / 157.3 * 100:.1f} %)", flush=True)
