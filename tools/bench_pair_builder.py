#!/usr/bin/env python3
"""The materialising N^2 pair builder at BASELINE cfg2's shape (N = 32, T = 150, D = 2048 -> [992, 4096, 150] fp32 = 2.44 GB
written per video, + 8 geometry channels per pair and frame): `transpose_gather_rows_kernel` / `pair_geometry_kernel`
(csrc/tspn_pairs.hip).  It is NOT on the product path -- the fused pass never materialises the pair tensor --
but north_star names the kernel and asks for its HBM rate.  HIP events over back-to-back calls; run it under
`rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the per-kernel evidence.
    python tools/bench_pair_builder.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

dev = torch.device("cuda", 0)
N, T, D = 32, 150, 2048
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device=dev).manual_seed(0)
feats = torch.rand((N, T, D), device=dev, generator=g)
xy = torch.floor(torch.rand((N, T, 2), device=dev, generator=g) * 900.0)
wh = torch.floor(10.0 + torch.rand((N, T, 2), device=dev, generator=g) * 290.0)
boxes = torch.cat([xy, xy + wh], dim=2).contiguous()
pairs = tspn.ops.pair_index(N, dev)
P = pairs.shape[0]
for want_feat, want_geom, name in ((True, False, "transpose_gather_rows_kernel (features)"), (False, True, "pair_geometry_kernel (boxes)"),
                                   (True, True, "both")):
    for _ in range(2):
        out = tspn.ops.pair_gather(feats, boxes, pairs, want_feat=want_feat, want_geom=want_geom, check_pairs=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        out = tspn.ops.pair_gather(feats, boxes, pairs, want_feat=want_feat, want_geom=want_geom, check_pairs=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    wr = (P * 2 * D * T * 4 if want_feat else 0) + (P * 8 * T * 4 if want_geom else 0)
    rd_unique = (N * T * D * 4 if want_feat else 0) + (N * T * 16 if want_geom else 0)
    print(f"{name}: {ms * 1e3:.1f} us per video; written {wr / 1e9:.3f} GB -> {wr / ms / 1e9:.2f} TB/s of writes "
          f"(unique reads {rd_unique / 1e6:.1f} MB; moved r + w = {2 * wr / ms / 1e9:.2f} TB/s with every element read from L2 / MALL once per use)",
          flush=True)
