#!/usr/bin/env python3
"""Time conv3_tc_wino43r (and optionally the canonical kernel) at the cfg2 projection shape.
TSPN_LIB_PATH selects a variant build; results of ablation builds are wrong by construction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 8
both = len(sys.argv) > 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((videos * 32, 150, 2048), device=dev, generator=g)
p6 = (torch.rand((6, 2048, 8192), device=dev, generator=g) - 0.5) * 0.02
fr = tspn.ops.repack_wino43_frag(p6)
todo = [("wino43r", tspn.ops.conv3_tc_wino43r, fr)]
if both:
    todo.append(("wino43 ", tspn.ops.conv3_tc_wino43, p6))
for name, fn, w in todo:
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
    print(f"[{os.path.basename(os.environ.get('TSPN_LIB_PATH', 'default'))}] {name} videos={videos}: median {ms[3]:.3f} ms "
          f"min {ms[0]:.3f} -> {flop / ms[3] / 1e9:.1f} TFLOP/s ({flop / ms[3] / 1e9 / 157.3 * 100:.1f} %)", flush=True)
