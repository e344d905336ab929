#!/usr/bin/env python3
"""Time the blocked pair-stage kernel alone at the cfg2 shape (A/B of kernel variants)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--c", type=int, default=4096)
ap.add_argument("--t", type=int, default=150)
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
y = torch.rand((args.videos * args.n, 2 * args.c, args.t), device=dev, generator=g) - 0.5
wh = (torch.rand((12, args.c), device=dev, generator=g) - 0.5) * 0.02
bh = torch.zeros(12, device=dev)
for _ in range(2):
    out = tspn.ops.heads_pairgrid(y, args.videos, args.n, wh, bh)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
for a, b in evs:
    a.record()
    out = tspn.ops.heads_pairgrid(y, args.videos, args.n, wh, bh)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)
pairs = args.videos * args.n * (args.n - 1)
print(f"heads_pairgrid B={args.videos} N={args.n} C={args.c} T={args.t}: median {ms[len(ms)//2]:.3f} ms "
      f"min {ms[0]:.3f} ms ({2.0 * pairs * args.t * args.c * 16 / ms[len(ms)//2] / 1e9:.1f} TFLOP/s incl. H padding)")
