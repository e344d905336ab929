#!/usr/bin/env python3
"""Soak of the role-split res4 tail: N launches into a poisoned output on two HIP streams at once, every result compared bit for
bit with the one-role kernel's (timing-dependent faults -- a lost counter hand-over, the store-data hazard -- are rare events).
    python tools/soak_tail_io.py [launches per stream] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 18
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(11)
CM, H, W = 256, 45, 80
h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
res = (torch.rand((frames, H, W, 4 * CM), device=dev, generator=g) - 0.5).to(torch.bfloat16)
w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
b2, b3 = torch.rand(CM, device=dev, generator=g) - 0.5, torch.rand(4 * CM, device=dev, generator=g) - 0.5
f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [torch.empty_like(want) for _ in streams]
bad = torch.zeros(2, dtype=torch.int64, device=dev)
for it in range(N):
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            outs[k].fill_(777.0)
            tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=outs[k], io_waves=True)
            bad[k] += (outs[k] != want).sum()
    if (it + 1) % 100 == 0:
        torch.cuda.synchronize()
        print(f"{it + 1} launches per stream: mismatching outputs so far {bad.tolist()}", flush=True)
torch.cuda.synchronize()
print("SOAK OK" if int(bad.sum()) == 0 else f"SOAK FAILED: {bad.tolist()}")
sys.exit(0 if int(bad.sum()) == 0 else 1)
