#!/usr/bin/env python3
"""Time the bf16 temporal conv at the cfg3 projection shape (4 videos: 256 tracklets x T=900 x D=1024 -> 2C=4096 rows).
    python tools/time_bf16.py [videos] [rounds]      (TSPN_LIB_PATH selects a variant build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
N, T, D = 64, 900, 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = tspn.ops.cast_bf16(torch.rand((videos * N, T, D), device=dev, generator=g))
w = (torch.rand((2 * D, 2 * D, 3), device=dev, generator=g) - 0.5) * 0.02
packed = tspn.ops.pack_conv3_bf16(w, split=D)
del w
times = []
y = tspn.ops.conv3_tc_bf16(x, packed)
torch.cuda.synchronize()
for r in range(rounds):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = tspn.ops.conv3_tc_bf16(x, packed)
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
    del y
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
v = sorted(times)
flop = 2.0 * videos * N * T * 3 * D * 4 * D
print(f"[{tag}] conv3_bf16 videos={videos}: median {v[len(v) // 2]:.3f} ms min {v[0]:.3f} max {v[-1]:.3f} "
      f"-> {flop / v[len(v) // 2] / 1e9:.0f} TFLOP/s", flush=True)
