#!/usr/bin/env python3
"""fp32 error of the temporal-conv algorithms against float64 on post-ReLU-like, heavy-tailed features and
trained-scale weights at the headline contraction depth (K = 3 x 2048): absolute, relative to max|y|, and
relative to the natural fp32 bound  eps * sum_k |x_k||w_k|  of each output element."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

dev = torch.device("cuda", 0)
B, T, Cin, M = 3, 150, 2048, 128
EPS = 2.0 ** -24


def heavy(seed, shape, scale=4.0, outlier=50.0, frac=1e-3):
    x = np.abs(tspn.hashrng.normal(seed, "x", shape, std=1.0)) * scale
    u = tspn.hashrng.uniform(seed, "o", shape)
    x = np.where(u < frac, x * outlier, x)
    z = tspn.hashrng.uniform(seed, "z", shape)
    return np.where(z < 0.4, 0.0, x).astype(np.float32)    # post-ReLU: 40 % exact zeros


def run(name, x, w):
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    y64 = torch.nn.functional.conv1d(torch.from_numpy(x).double().transpose(1, 2), torch.from_numpy(w).double(), padding=1).numpy()
    mag = torch.nn.functional.conv1d(torch.from_numpy(np.abs(x)).double().transpose(1, 2), torch.from_numpy(np.abs(w)).double(), padding=1).numpy()
    outs = {"direct": tspn.ops.conv3_tc(xd, tspn.ops.pack_conv3(wd))}
    for nm, pk, fn in (("F(2,3)", "pack_conv3_wino", "conv3_tc_wino"), ("F(4,3)", "pack_conv3_wino43", "conv3_tc_wino43"),
                       ("F(6,3)", "pack_conv3_wino63", "conv3_tc_wino63")):
        if hasattr(tspn.ops, pk):
            outs[nm] = getattr(tspn.ops, fn)(xd, getattr(tspn.ops, pk)(wd))
    print(f"{name}: max|y| {np.abs(y64).max():.3g}, max sum|x||w| {mag.max():.3g}, max|x| {np.abs(x).max():.3g}")
    for nm, y in outs.items():
        e = np.abs(y.cpu().numpy() - y64)
        print(f"   {nm:7s} max abs {e.max():.3e}  / max|y| {e.max() / np.abs(y64).max():.3e}  "
              f"max e/(eps*sum|x||w|) {(e / (EPS * mag + 1e-300)).max():.2f}  rms/rms(y) {np.sqrt((e**2).mean()) / np.sqrt((y64**2).mean()):.3e}")


run("synthetic U[0,1) x, w N(0,0.01^2)", tspn.hashrng.uniform(48, "x", (B, T, Cin)), tspn.hashrng.normal(48, "w", (M, Cin, 3), std=0.01))
run("heavy-tailed |N|*4, 0.1% x50 outliers, 40% zeros; w N(0, 1/(3D))", heavy(71, (B, T, Cin)),
    tspn.hashrng.normal(71, "w", (M, Cin, 3), std=1.0 / np.sqrt(3 * Cin)))
run("same x; w N(0, 0.05^2) with 1% x20 rows", heavy(72, (B, T, Cin)),
    (tspn.hashrng.normal(72, "w", (M, Cin, 3), std=0.05) * np.where(tspn.hashrng.uniform(72, "r", (M, 1, 1)) < 0.02, 20.0, 1.0)).astype(np.float32))
run("slowly varying in time (x_t = base + small noise), trained-scale w", 
    (heavy(73, (B, 1, Cin)) + 0.05 * tspn.hashrng.normal(73, "n", (B, T, Cin), std=1.0)).astype(np.float32),
    tspn.hashrng.normal(73, "w", (M, Cin, 3), std=1.0 / np.sqrt(3 * Cin)))
