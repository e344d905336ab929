#!/usr/bin/env python3
"""A/B of the two Winograd F(4,3) kernels: bit-identity on edge shapes, then timing at the cfg2 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (B, T, Cin, M) in [(1, 4, 8, 32), (2, 5, 8, 32), (3, 7, 16, 64), (2, 30, 24, 96), (5, 150, 32, 160),
                       (3, 33, 40, 128), (7, 150, 64, 256), (1, 1, 8, 32), (4, 2, 16, 288), (33, 13, 48, 128)]:
    x = torch.rand((B, T, Cin), device=dev, generator=g)
    w = (torch.rand((M, Cin, 3), device=dev, generator=g) - 0.5) * 0.2
    b = torch.rand((M,), device=dev, generator=g)
    p6 = tspn.ops.pack_conv3_wino43(w)
    fr = tspn.ops.repack_wino43_frag(p6)
    for relu in (False, True):
        y0 = tspn.ops.conv3_tc_wino43(x, p6, b, relu=relu)
        y1 = tspn.ops.conv3_tc_wino43r(x, fr, b, relu=relu)
        torch.cuda.synchronize()
        same = torch.equal(y0, y1)
        if not same:
            bad += 1
        print(f"B={B} T={T} Cin={Cin} M={M} relu={relu}: {'bit-identical' if same else 'DIFF max %g' % (y0 - y1).abs().max().item()}",
              flush=True)
print("edge shapes:", "OK" if bad == 0 else f"{bad} FAILED", flush=True)
if bad:
    sys.exit(1)

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand((videos * 32, 150, 2048), device=dev, generator=g)
p6 = (torch.rand((6, 2048, 8192), device=dev, generator=g) - 0.5) * 0.02
fr = tspn.ops.repack_wino43_frag(p6)
y0 = tspn.ops.conv3_tc_wino43(x, p6)
y1 = tspn.ops.conv3_tc_wino43r(x, fr)
torch.cuda.synchronize()
print("cfg2 shape bit-identical:", torch.equal(y0, y1), flush=True)
del y0, y1
for name, fn, w in (("wino43 ", tspn.ops.conv3_tc_wino43, p6), ("wino43r", tspn.ops.conv3_tc_wino43r, fr),
                    ("wino43 ", tspn.ops.conv3_tc_wino43, p6), ("wino43r", tspn.ops.conv3_tc_wino43r, fr)):
    for _ in range(2):
        fn(x, w)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in evs:
        a.record()
        fn(x, w)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    flop = 0.5 * (152 / 150) * 2.0 * 8192 * 3 * 2048 * x.shape[0] * 150
    print(f"{name} videos={videos}: median {ms[3]:.3f} ms min {ms[0]:.3f} -> {flop / ms[3] / 1e9:.1f} TFLOP/s executed "
          f"({flop / ms[3] / 1e9 / 157.3 * 100:.1f} % of fp32 MFMA peak)", flush=True)
