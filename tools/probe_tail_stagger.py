#!/usr/bin/env python3
"""Does running the memory phase of one workgroup set under the MFMA phase of another buy anything on this chip?
Two half-batches of the res4 bottleneck tail on two HIP streams, the second offset by `delay` us, ten launches each,
against one stream with the full batch (lockstep).   python tools/probe_tail_stagger.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
CM, H, W, F = 256, 45, 80, 16
def mk(frames):
    h1 = torch.rand((frames, H, W, CM), device=dev, generator=g).to(torch.bfloat16)
    res = torch.rand((frames, H, W, 4 * CM), device=dev, generator=g).to(torch.bfloat16)
    return h1, res
w2 = (torch.rand((CM, CM, 3, 3), device=dev, generator=g) - 0.5) * 0.05
w3 = (torch.rand((4 * CM, CM, 1, 1), device=dev, generator=g) - 0.5) * 0.1
b2, b3 = torch.zeros(CM, device=dev), torch.zeros(4 * CM, device=dev)
f2, f3 = tspn.ops.pack_conv2d_frag_bf16(w2), tspn.ops.pack_conv2d_frag_bf16(w3)
full = mk(F); halves = [mk(F // 2), mk(F // 2)]
run = lambda t: tspn.ops.bottleneck_tail_bf16(t[0], f2, b2, f3, b3, t[1])
REPS = 10
def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / REPS)
    return sorted(ts)[len(ts) // 2]
def single():
    for _ in range(REPS): run(full)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
clk = 100e3  # torch.cuda._sleep counts ticks of the 100 MHz wall clock on ROCm (s_memrealtime)
def two(delay_us):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        for _ in range(REPS): run(halves[0])
    with torch.cuda.stream(s2):
        if delay_us: torch.cuda._sleep(int(delay_us * clk / 1e3))
        for _ in range(REPS): run(halves[1])
    main.wait_stream(s1); main.wait_stream(s2)
print(f"one stream, {F} frames per launch: {timed(single):.1f} us per launch (lockstep)")
for d in (0, 30, 60, 90):
    print(f"two streams x {F // 2} frames, second offset by {d} us: {timed(lambda: two(d)):.1f} us per pair of launches")
