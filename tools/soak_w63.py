#!/usr/bin/env python3
"""Soak of the F(6,3) conv kernel: (1) full-size launches (16 videos of cfg2) under concurrent memory traffic, every
one bit-identical to the first; (2) random shapes (B, T, Cin, M, relu, bias) against the direct-form kernel.
    python tools/soak_w63.py [full_size_launches] [random_shapes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

n_full = int(sys.argv[1]) if len(sys.argv) > 1 else 80
n_rand = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.rand((512, 150, 2048), device=dev, generator=g)
w = (torch.rand((8192, 2048, 3), device=dev, generator=g) - 0.5) * 0.02
f63 = tspn.ops.pack_conv3_wino63(w)
del w
ws = torch.empty(tspn._abi.lib().tspn_conv3_tc_wino63_workspace_bytes(512, 150, 2048), dtype=torch.uint8, device=dev)
ref = tspn.ops.conv3_tc_wino63(x, f63, workspace=ws)
side = torch.cuda.Stream(device=dev)
bad = 0
for i in range(n_full):
    with torch.cuda.stream(side):
        junk = x * 1.0001  # noqa: F841
    y = tspn.ops.conv3_tc_wino63(x, f63, workspace=ws)
    bad += 0 if torch.equal(y, ref) else 1
    del y
    if i % 20 == 19:
        torch.cuda.synchronize()
        print(f"full size: {i + 1} launches, {bad} differ", flush=True)
torch.cuda.synchronize()
del x, ref, ws, f63
cpu = torch.Generator().manual_seed(11)
worst = 0.0
for i in range(n_rand):
    B = int(torch.randint(1, 9, (1,), generator=cpu))
    T = int(torch.randint(1, 200, (1,), generator=cpu))
    Cin = 32 * int(torch.randint(1, 9, (1,), generator=cpu))
    M = 32 * int(torch.randint(1, 13, (1,), generator=cpu))
    relu = bool(torch.randint(0, 2, (1,), generator=cpu))
    xx = torch.rand((B, T, Cin), device=dev, generator=g) - 0.5
    ww = (torch.rand((M, Cin, 3), device=dev, generator=g) - 0.5) * 0.2
    bb = torch.rand((M,), device=dev, generator=g) - 0.5 if i % 3 else None
    got = tspn.ops.conv3_tc_wino63(xx, tspn.ops.pack_conv3_wino63(ww), bias=bb, relu=relu)
    exp = torch.nn.functional.conv1d(xx.double().transpose(1, 2), ww.double(), None if bb is None else bb.double(),
                                     padding=1)
    if relu:
        exp = torch.relu(exp)
    err = float((got.double() - exp).abs().max())
    worst = max(worst, err)
    if err > 2e-5:
        print(f"MISMATCH B={B} T={T} Cin={Cin} M={M} relu={relu} bias={bb is not None}: {err:.3e}")
        bad += 1
print(f"random shapes: {n_rand} cases, worst |err| vs float64 {worst:.3e}; total failures {bad}")
sys.exit(1 if bad else 0)
