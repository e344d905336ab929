#!/usr/bin/env python3
"""res5's 1x1 convs on a chunk of RoIs: the generic bf16 conv kernel (tspn_conv2d_nhwc_bf16: 256 x 128 tiles, weights
straight from L2) against the big-tile GEMM (tspn_conv1x1_big_bf16: 256 x 256 tiles, both operands through LDS); results
compared bit for bit.   python tools/time_conv1x1_big.py [rois ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
roi_counts = [int(a) for a in sys.argv[1:]] or [2400]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


for R in roi_counts:
    shapes = [("res5.0 conv1 1024->512", (R, 7, 7, 1024), 512, 1, False, True),
              ("res5.0 shortcut 1024->2048", (R, 7, 7, 1024), 2048, 1, False, False),
              ("res5 conv1 2048->512", (R, 7, 7, 2048), 512, 1, False, True),
              ("res5 conv3 512->2048 + residual", (R, 7, 7, 512), 2048, 1, True, True)]
    tot = [0.0, 0.0]
    for name, xs, cout, stride, res, relu in shapes:
        x = torch.rand(xs, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.rand((cout, xs[3], 1, 1), device=dev, generator=g) - 0.5) * 0.05
        f = tspn.ops.pack_conv2d_frag_bf16(w)
        rows = tspn.ops.pack_conv1x1_rows_bf16(w)
        b = torch.rand(cout, device=dev, generator=g) - 0.5
        oh, ow = (xs[1] - 1) // stride + 1, (xs[2] - 1) // stride + 1
        r = torch.rand((xs[0], oh, ow, cout), device=dev, generator=g).to(torch.bfloat16) if res else None
        gen = lambda: tspn.ops.conv2d_nhwc_bf16(x, f, (1, 1), stride, 0, bias=b, residual=r, relu=relu)
        big = lambda: tspn.ops.conv1x1_big_bf16(x, rows, stride, bias=b, residual=r, relu=relu)
        same = torch.equal(gen(), big())
        ug, ub = timeit(gen), timeit(big)
        npix = xs[0] * oh * ow
        fl = 2.0 * npix * cout * xs[3]
        mult = 3 if name.startswith("res5 conv") else 1          # blocks 2, 3 (conv1) / all three blocks (conv3)
        if name.startswith("res5 conv1"):
            mult = 2
        tot[0] += mult * ug; tot[1] += mult * ub
        print(f"{R} RoIs  {name}: generic {ug:.1f} us ({fl / ug / 1e6:.0f} TFLOP/s)  big {ub:.1f} us ({fl / ub / 1e6:.0f} TFLOP/s)  "
              f"tiles {tspn.ops.conv1x1_big_tiles(npix, cout)}  equal {same}", flush=True)
    print(f"{R} RoIs  all 1x1 layers of res5 per chunk: generic {tot[0]:.0f} us, big {tot[1]:.0f} us; "
          f"per 57600 RoIs {tot[0] * 57600 / R / 1e3:.1f} -> {tot[1] * 57600 / R / 1e3:.1f} ms", flush=True)
