#!/usr/bin/env python3
"""cfg3 (N=64, T=900, D=1024, bf16): does the temporal conv of chunk k+1 overlap the pair stage of chunk k when the
4-video step is split into two 2-video chunks on two HIP streams?  ("measure the overlap you argued
against".)  Forms, all on the same inputs, interleaved in one process, median of `rounds`:
  one    conv(4 videos) ; pair(4 videos)                         one stream  (what tspn_forward_fused_bf16 does)
  serial conv(c0) ; pair(c0) ; conv(c1) ; pair(c1)               one stream  (the price of 2-video launches alone)
  two    A: conv(c0) ; conv(c1)      B: pair(c0) | pair(c1)       pair(ck) behind conv(ck)'s event
    python tools/probe_cfg3_overlap.py [videos=4] [rounds=9]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
N, T, D, H = 64, 900, 1024, 12
C = 2 * D
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
feats = (torch.rand((videos * N, T, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
cw = (torch.rand((C, C, 3), device=dev, generator=g) - 0.5) * 0.02
packed = tspn.ops.pack_conv3_bf16(cw, split=D)
del cw
bias2 = torch.zeros(2 * C, device=dev)
hw = tspn.ops.pack_heads_bf16((torch.rand((H, C), device=dev, generator=g) - 0.5) * 0.1)
hb = torch.zeros(H, device=dev)
half = videos // 2
chunks = [(0, half), (half, videos)]
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def conv(lo, hi):
    return tspn.ops.conv3_tc_bf16(feats[lo * N:hi * N], packed, bias2)


def pair(y, nv):
    return tspn.ops.heads_pairgrid_bf16(y, nv, N, hw, hb, H)


def form_one():
    y = conv(0, videos)
    return [y, pair(y, videos)]


def form_serial():
    keep = []
    for lo, hi in chunks:
        y = conv(lo, hi)
        keep += [y, pair(y, hi - lo)]
    return keep


def form_two():
    keep = []
    main = torch.cuda.current_stream(dev)
    sa.wait_stream(main)
    sb.wait_stream(main)
    for lo, hi in chunks:
        with torch.cuda.stream(sa):
            y = conv(lo, hi)
            ev = torch.cuda.Event()
            ev.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ev)
            keep += [y, pair(y, hi - lo)]
    main.wait_stream(sa)
    main.wait_stream(sb)
    return keep


forms = {"one": form_one, "serial": form_serial, "two": form_two}
ref = form_one()[1]
torch.cuda.synchronize()
for name in ("serial", "two"):
    out = forms[name]()
    torch.cuda.synchronize()
    got = torch.cat([out[1], out[3]])
    assert torch.equal(got, ref), name          # same launches on slices: same bits
    del out, got
times = {k: [] for k in forms}
for r in range(rounds):
    for name, fn in forms.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        keep = fn()
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b))
        del keep
pairs = videos * N * (N - 1)
for name, v in times.items():
    v = sorted(v)
    med = v[len(v) // 2]
    print(f"{name:7s} median {med:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}   {pairs / med / 1e3:.3f} M pairs/s (conv + pair stage only)", flush=True)
