import os, sys
sys.path.insert(0, "/root/repo")
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n, t = 64, 300
fm = torch.rand((t, 45, 80, 1024), device=dev, generator=g).to(torch.bfloat16)
xy = torch.rand((n, t, 2), device=dev, generator=g) * torch.tensor([900.0, 400.0], device=dev)
wh = 40 + torch.rand((n, t, 2), device=dev, generator=g) * 260
boxes = torch.cat([xy, xy + wh], dim=2).contiguous()
for chunk in (2400, 1200, 4800):
    for ns in (1, 2):
        head = tspn.Res5RoIHead(roi_chunk=chunk).to(dev)
        head.streams = ns
        head(fm, boxes); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); head(fm, boxes); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(f"roi_chunk {chunk} streams {ns}: {sorted(ts)[2]:.2f} ms", flush=True)
