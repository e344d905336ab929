#!/usr/bin/env python3
"""How much of the F(6,3) conv's run time does not scale with the channel count (output transform + stores, launch,
first burst): the kernel at Cin = 512 .. 4096 on the cfg2 column shape; 2 t(K) - t(2K) is the K-independent part."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = tspn._abi.lib()
res = {}
for Cin in (512, 1024, 2048, 4096):
    x = torch.rand((512, 150, Cin), device=dev, generator=g)
    w = (torch.rand((8192, Cin, 3), device=dev, generator=g) - 0.5) * 0.02
    f = tspn.ops.pack_conv3_wino63(w); del w
    ws = torch.empty(lib.tspn_conv3_tc_wino63_workspace_bytes(512, 150, Cin), dtype=torch.uint8, device=dev)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = tspn.ops.conv3_tc_wino63(x, f, workspace=ws); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b)); del y
    ts = sorted(ts[2:]); res[Cin] = ts[len(ts) // 2]
    print(Cin, f"{res[Cin]:.3f} ms", flush=True)
    del x, f, ws
print("fixed part (2 t(1024) - t(2048)):", 2 * res[1024] - res[2048], " (2 t(2048) - t(4096)):", 2 * res[2048] - res[4096])
