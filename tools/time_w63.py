#!/usr/bin/env python3
"""Time the F(6,3) conv (transform + contract) against the direct kernel at the cfg2 projection shape on one box.
    python tools/time_w63.py [videos] [rounds]      (TSPN_LIB_PATH selects a variant build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((videos * 32, 150, 2048), device=dev, generator=g)
w = (torch.rand((8192, 2048, 3), device=dev, generator=g) - 0.5) * 0.02
pd = tspn.ops.pack_conv3(w)
f63 = tspn.ops.pack_conv3_wino63(w)
del w
lib = tspn._abi.lib()
ws = torch.empty(lib.tspn_conv3_tc_wino63_workspace_bytes(videos * 32, 150, 2048), dtype=torch.uint8, device=dev)
arms = {"direct": lambda: tspn.ops.conv3_tc(x, pd),
        "wino63": lambda: tspn.ops.conv3_tc_wino63(x, f63, workspace=ws)}
times = {k: [] for k in arms}
for fn in arms.values():
    fn()
torch.cuda.synchronize()
for r in range(rounds):
    for k, fn in arms.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = fn()
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b))
        del y
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
for k, v in times.items():
    v = sorted(v)
    print(f"[{tag}] {k:8s} videos={videos}: median {v[len(v) // 2]:.3f} ms min {v[0]:.3f} max {v[-1]:.3f}", flush=True)
