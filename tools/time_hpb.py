#!/usr/bin/env python3
"""Time the bf16 pair stage at the cfg3 shape (4 videos of N=64, T=900, C=2048) on projections y [B*N, T, 2C].
    python tools/time_hpb.py [videos] [rounds]      (TSPN_LIB_PATH selects a variant build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
N, T, C, H = 64, 900, 2048, 12
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
y = torch.rand((videos * N, T, 2 * C), device=dev, generator=g) - 0.5
hw = tspn.ops.pack_heads_bf16((torch.rand((H, C), device=dev, generator=g) - 0.5) * 0.1)
hb = torch.zeros(H, device=dev)
out = tspn.ops.heads_pairgrid_bf16(y, videos, N, hw, hb, H)
torch.cuda.synchronize()
chk = float(out.double().abs().sum())
del out
times = []
for r in range(rounds):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = tspn.ops.heads_pairgrid_bf16(y, videos, N, hw, hb, H)
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
    del out
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
v = sorted(times)
print(f"[{tag}] heads_pairgrid_bf16 videos={videos}: median {v[len(v) // 2]:.3f} ms min {v[0]:.3f} max {v[-1]:.3f}  checksum {chk:.6e}", flush=True)
