#!/usr/bin/env python3
"""cfg3 bf16 temporal conv: the lockstep kernel (conv3_bf16_big_kernel) against the ping-pong form (conv3_bf16_pp_kernel),
interleaved in one process (TSPN_CONV3_BF16_PP is read per call), results compared bit for bit.
    python tools/time_bf16_pp.py [videos] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
N, T, D = 64, 900, 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = tspn.ops.cast_bf16(torch.rand((videos * N, T, D), device=dev, generator=g))
w = (torch.rand((2 * D, 2 * D, 3), device=dev, generator=g) - 0.5) * 0.02
packed = tspn.ops.pack_conv3_bf16(w, split=D)
del w


def run(pp):
    os.environ["TSPN_CONV3_BF16_PP"] = str(pp)
    return tspn.ops.conv3_tc_bf16(x, packed)


ya, yb = run(0), run(1)
torch.cuda.synchronize()
print("equal:", torch.equal(ya, yb), flush=True)
del ya, yb
times = {0: [], 1: []}
for r in range(rounds):
    for pp in (0, 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = run(pp); b.record(); torch.cuda.synchronize()
        times[pp].append(a.elapsed_time(b)); del y
flop = 2.0 * videos * N * T * 3 * D * 4 * D
for pp, name in ((0, "lockstep"), (1, "ping-pong")):
    v = sorted(times[pp])
    print(f"{name}: median {v[len(v) // 2]:.3f} ms (min {v[0]:.3f}) -> {flop / v[len(v) // 2] / 1e9:.0f} TFLOP/s", flush=True)
