#!/usr/bin/env python3
"""Does running two half-batches of the backbone on two HIP streams (kernels of different phases overlapping) beat one
full batch?   python tools/bench_backbone_streams.py [frames per stream] [streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
per = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
net = tspn.ResNetC4(depth=101, frame_chunk=per).to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in tspn.synth.make_backbone_weights(0).items()})
net = net.to(dev)
g = torch.Generator(device=dev).manual_seed(0)
imgs = [torch.rand((per, 720, 1280, 3), device=dev, generator=g) - 0.5 for _ in range(ns)]
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
for im in imgs:
    net(im, bf16=True)
torch.cuda.synchronize()
def run():
    for k, (st, im) in enumerate(zip(streams, imgs)):
        with torch.cuda.stream(st):
            net(im, bf16=True)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for st in streams: st.wait_event(a)
    run()
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = sorted(ts)[len(ts) // 2]
print(f"{ns} streams x {per} frames: {ms:.2f} ms = {ms / (ns * per):.3f} ms per frame", flush=True)
