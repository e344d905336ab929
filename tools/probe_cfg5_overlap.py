#!/usr/bin/env python3
"""cfg5: the RoI head of frame group g under the backbone of group g + 1 (second HIP stream) against backbone of the
whole video, then RoI head of the whole video.   python tools/probe_cfg5_overlap.py [frames] [group]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn

T = int(sys.argv[1]) if len(sys.argv) > 1 else 900
group = int(sys.argv[2]) if len(sys.argv) > 2 else 36
N, H, W = 64, 720, 1280
dev = torch.device("cuda", 0)
t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
net = tspn.ResNetC4(depth=101, frame_chunk=9)
net.load_state_dict(t(tspn.synth.make_backbone_weights(0)))
net = net.to(dev)
head = tspn.Res5RoIHead()
head.load_state_dict(t(tspn.synth.make_res5_weights(0)))
head = head.to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((T, H, W, 3), device=dev, generator=gen) - 0.5
xy = torch.rand((N, T, 2), device=dev, generator=gen) * torch.tensor([900.0, 400.0], device=dev)
wh = 40 + torch.rand((N, T, 2), device=dev, generator=gen) * 260
boxes = torch.cat([xy, xy + wh], dim=2).contiguous()
side = torch.cuda.Stream(device=dev)


def sequential():
    return head(net(img, bf16=True), boxes)


def overlapped():
    main = torch.cuda.current_stream(dev)
    feats = torch.empty((N, T, head.out_channels), dtype=torch.bfloat16, device=dev)
    side.wait_stream(main)
    for lo in range(0, T, group):
        hi = min(T, lo + group)
        maps = net(img[lo:hi], bf16=True)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            feats[:, lo:hi] = head(maps, boxes[:, lo:hi].contiguous())
            maps.record_stream(side)
    main.wait_stream(side)
    return feats


res = {}
for name, fn in (("sequential", sequential), ("overlapped", overlapped), ("sequential", sequential), ("overlapped", overlapped)):
    out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    torch.cuda.synchronize()
    res.setdefault(name, []).append(a.elapsed_time(b))
    res[name + "_out"] = out
print({k: v for k, v in res.items() if not k.endswith("_out")}, "equal:", torch.equal(res["sequential_out"], res["overlapped_out"]))
