#!/usr/bin/env python3
"""res4 tail (CM = 256, 45 x 80 maps): the shipped fused tail (4 waves per workgroup, 2 workgroups per CU) against the role-split
form (tspn_bottleneck_tail_io_bf16: 4 MFMA waves + 4 io waves, 1 workgroup per CU), bit-identity included.
    python tools/time_tail_io.py [frames ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
CM, H, W = 256, 45, 80
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
for frames in [int(a) for a in sys.argv[1:]] or [9, 18, 36]:
    h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
    res = (torch.rand((frames, H, W, 4 * CM), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
    b2, b3 = torch.rand(CM, device=dev, generator=g) - 0.5, torch.rand(4 * CM, device=dev, generator=g) - 0.5
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    out = torch.empty_like(res)
    arms = {"shipped": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=out),
            "io_waves": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=out, io_waves=True)}
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
    same = torch.equal(arms["io_waves"](), want)
    t = {}
    for name, fn in arms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):                     # ten launches back to back per event pair: the queue never runs dry, so the
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # host's launch latency is not in it
            a.record()
            for _ in range(10):
                fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e2)
        t[name] = sorted(ts)[len(ts) // 2]
    fl = 2.0 * frames * H * W * CM * CM * 13
    print(f"[{tag}] {frames} frames: shipped {t['shipped']:.1f} us ({fl / t['shipped'] / 1e6:.0f} TFLOP/s)  role-split {t['io_waves']:.1f} us "
          f"({fl / t['io_waves'] / 1e6:.0f} TFLOP/s, {t['io_waves'] / t['shipped'] - 1:+.1%})  bit-identical: {same}", flush=True)
