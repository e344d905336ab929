import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CM = 256
ok = True
for (NB, H, W) in [(1, 1, 1), (3, 7, 11), (2, 16, 8), (1, 13, 129), (1, 45, 80), (9, 45, 80)]:
    h1 = t(tspn.hashrng.uniform(94, "h1", (NB, H, W, CM), 0, 1)).to(dev).to(torch.bfloat16)
    res = t(tspn.hashrng.uniform(94, "res", (NB, H, W, 4 * CM), -1, 1)).to(dev).to(torch.bfloat16)
    w2 = t(tspn.hashrng.normal(94, "w2", (CM, CM, 3, 3), std=float(np.sqrt(2.0 / (9 * CM))))).to(dev)
    w3 = t(tspn.hashrng.normal(94, "w3", (4 * CM, CM, 1, 1), std=float(np.sqrt(2.0 / CM)))).to(dev)
    b2 = t(tspn.hashrng.normal(94, "b2", (CM,), std=0.1)).to(dev)
    b3 = t(tspn.hashrng.normal(94, "b3", (4 * CM,), std=0.1)).to(dev)
    f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
    want = tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res)
    for trial in range(3):
        out = torch.full_like(want, 777.0)
        tspn.ops.bottleneck_tail_bf16(h1, f2, b2, f3, b3, res, out=out, io_waves=True)
        torch.cuda.synchronize()
        bad = int((out != want).sum())
        print((NB, H, W), "trial", trial, "mismatches", bad, "of", out.numel(), flush=True)
        ok &= bad == 0
print("ALL OK" if ok else "FAILED")
