#!/usr/bin/env python3
"""One-launch identity bottleneck block (tspn_bottleneck_block_bf16) against the chain it replaces (conv1 launch + fused
tail launch) at the backbone's res2 / res3 shapes, 720p frames.
    python tools/time_block.py [frames=9]        (TSPN_LIB_PATH selects a probe build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
for CM, H, W in ((64, 180, 320), (128, 90, 160)):
    x = (torch.rand((frames, H, W, 4 * CM), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    w1 = (torch.rand((CM, 4 * CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
    w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
    w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
    b1, b2, b3 = torch.zeros(CM, device=dev), torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev)
    f1, f2, f3 = (tspn.ops.pack_conv2d_frag_bf16(w) for w in (w1, w2, w3))
    out = torch.empty_like(x)
    h1 = tspn.ops.conv2d_nhwc_bf16(x, f1, (1, 1), 1, 0, bias=b1, relu=True)
    arms = {"conv1": lambda: tspn.ops.conv2d_nhwc_bf16(x, f1, (1, 1), 1, 0, bias=b1, relu=True),
            "tail": lambda: tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, x, out=out),
            "block": lambda: tspn.ops.bottleneck_block_bf16(x, f1, b1, f2, b2, f3, b3, out=out)}
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, x)
    same = torch.equal(arms["block"](), want)
    res = {}
    for name, fn in arms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(11):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        res[name] = sorted(ts)[len(ts) // 2]
    px = frames * H * W
    fl = 2.0 * px * CM * CM * 17
    by = px * 4 * CM * 2 * 2
    chain = res["conv1"] + res["tail"]
    print(f"[{tag}] CM={CM} {frames}x{H}x{W}: conv1 {res['conv1']:.1f} + tail {res['tail']:.1f} = {chain:.1f} us;  one launch "
          f"{res['block']:.1f} us ({res['block'] / chain - 1:+.1%}; {fl / res['block'] / 1e6:.0f} TFLOP/s, "
          f"{by / res['block'] / 1e6:.2f} TB/s of map in + out)  bit-identical: {same}", flush=True)
