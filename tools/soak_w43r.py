#!/usr/bin/env python3
"""Soak test of conv3_tc_wino43r: N launches at the cfg2 shape must be bit-identical to the first and to the
canonical kernel (a race in the counted-wait / implied-completion logic would show up as a flaky mismatch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand((16 * 32, 150, 2048), device=dev, generator=g)
p6 = (torch.rand((6, 2048, 8192), device=dev, generator=g) - 0.5) * 0.02
fr = tspn.ops.repack_wino43_frag(p6)
ref = tspn.ops.conv3_tc_wino43(x, p6)
bad = 0
side = torch.cuda.Stream()
for i in range(n):
    with torch.cuda.stream(side):          # a second stream keeps the memory system busy with other traffic
        junk = x * 1.0001
    y = tspn.ops.conv3_tc_wino43r(x, fr)
    if not torch.equal(y, ref):
        bad += 1
        print(f"launch {i}: MISMATCH max {float((y - ref).abs().max())}", flush=True)
    del y
torch.cuda.synchronize()
print(f"{n} launches at 16 videos: {'all bit-identical to the canonical kernel' if bad == 0 else str(bad) + ' mismatches'}", flush=True)
sys.exit(1 if bad else 0)
