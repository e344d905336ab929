#!/usr/bin/env python3
"""A/B timing of the F(4,3) conv kernels at the cfg2 projection shape on ONE box, interleaved so that both arms
see the same clock: conv3_tc_wino43r (in-kernel input transform) vs conv3_tc_wino43v (transform pass + MFMA
kernel; the two launches are timed together and the MFMA kernel alone is derived by timing the transform alone).
    python tools/time_w43.py [videos] [rounds]
TSPN_LIB_PATH selects a variant build."""
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
p6 = (torch.rand((6, 2048, 8192), device=dev, generator=g) - 0.5) * 0.02
fr = tspn.ops.repack_wino43_frag(p6)
del p6
lib = tspn._abi.lib()
ws = torch.empty(lib.tspn_conv3_tc_wino43v_workspace_bytes(videos * 32, 150, 2048), dtype=torch.uint8, device=dev)
arms = {"wino43r": lambda: tspn.ops.conv3_tc_wino43r(x, fr),
        "wino43v (transform + contract)": lambda: tspn.ops.conv3_tc_wino43v(x, fr, workspace=ws)}
times = {k: [] for k in arms}
for k, fn in arms.items():
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
flop = 0.5 * (152 / 150) * 2.0 * 8192 * 3 * 2048 * x.shape[0] * 150
tag = os.path.basename(os.environ.get("TSPN_LIB_PATH", "default"))
for k, v in times.items():
    v = sorted(v)
    med = v[len(v) // 2]
    print(f"[{tag}] {k:32s} videos={videos}: median {med:.3f} ms min {v[0]:.3f} max {v[-1]:.3f} -> "
          f"{flop / med / 1e9:.1f} TFLOP/s ({flop / med / 1e9 / 157.3 * 100:.1f} % of fp32 MFMA peak)", flush=True)
