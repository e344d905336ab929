#!/usr/bin/env python3
"""Time the bf16-operand path at the VidOR long-clip shape (BASELINE config 3: N=64, T=900, D=1024):
the conv and pair-stage kernels on their own and the whole fused pass.  Target for rocprofv3."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=1)
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--t", type=int, default=900)
ap.add_argument("--d", type=int, default=1024)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--pad", type=int, default=0, help="floats of padding per row of y")
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, N, T, D = args.videos, args.n, args.t, args.d
C, A, K = 2 * D, 4, 132
g = torch.Generator(device=dev).manual_seed(0)
feats = torch.rand((B * N, T, D), device=dev, generator=g).to(torch.bfloat16)
conv_w = (torch.rand((C, C, 3), device=dev, generator=g) - 0.5) * 0.02
packed = tspn.ops.pack_conv3_bf16(conv_w, split=D)
del conv_w
conv_b = torch.zeros(C, device=dev)
hw = (torch.rand((3 * A, C), device=dev, generator=g) - 0.5) * 0.02
hpk = tspn.ops.pack_heads_bf16(hw)
hb = torch.zeros(3 * A, device=dev)
cls_w = ((torch.rand((K, C), device=dev, generator=g) - 0.5) * 0.02).to(torch.bfloat16).float()
cls_b = torch.zeros(K, device=dev)
pairs = torch.cat([tspn.ops.pair_index(N, dev, base=b * N) for b in range(B)])
bias2 = torch.cat([conv_b, torch.zeros(C, device=dev)])


def timed(fn):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for a, b in evs:
        a.record()
        out = fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], out


ms_conv, y = timed(lambda: tspn.ops.conv3_tc_bf16(feats, packed, bias2, ldm=2 * C + args.pad))
flop = 2.0 * (2 * C) * 3 * D * B * N * T
print(f"conv3 bf16  B*N={B * N} T={T} D={D} M={2 * C}: {ms_conv:.3f} ms -> {flop / ms_conv / 1e9:.0f} TFLOP/s")
ms_heads, _ = timed(lambda: tspn.ops.heads_pairgrid_bf16(y, B, N, hpk, hb, 3 * A))
P = B * N * (N - 1)
print(f"pair stage bf16  P={P}: {ms_heads:.3f} ms -> {P * T * C / ms_heads / 1e6:.0f} G activations/s, "
      f"{2.0 * P * T * C * 16 / ms_heads / 1e9:.0f} TFLOP/s (H padded to 16)")
del y
ms_all, _ = timed(lambda: tspn.ops.forward_fused_bf16(feats, pairs, B, N, packed, conv_b, hpk, hb, cls_w, cls_b))
print(f"fused bf16 pass: {ms_all:.3f} ms per {B} video(s) -> {P / ms_all * 1e3:.0f} tracklet-pairs/s")
