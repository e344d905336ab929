#!/usr/bin/env python3
"""ResNet-101-C4 backbone on 720p frames, one MI355X: frames -> res4 maps."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tspn_mi355x as tspn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--h", type=int, default=720)
ap.add_argument("--w", type=int, default=1280)
ap.add_argument("--depth", type=int, default=101)
ap.add_argument("--chunk", type=int, default=36)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--no-next", action="store_true", help="(default) fuse_next_conv1 = False: every conv1 as its own launch")
ap.add_argument("--next", action="store_true", help="fuse_next_conv1 = True: res4 tails also compute the follower's conv1")
ap.add_argument("--no-block", action="store_true", help="identity blocks of res2 / res3 as conv1 + fused tail (round 4) instead of one launch")
ap.add_argument("--no-proj", action="store_true", help="first blocks of the stages as conv1 + shortcut + fused tail")
ap.add_argument("--no-io-waves", action="store_true", help="res4 tails on the round-3 kernel (one role per wave) instead of the role-split one")
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = tspn.ResNetC4(depth=args.depth, frame_chunk=args.chunk).to(dev)
net.streams = args.streams
net.fuse_next_conv1 = bool(args.next) and not args.no_next
net.fuse_blocks = not args.no_block
net.fuse_first_blocks = not args.no_proj
net.tail_io_waves = not args.no_io_waves
g = torch.Generator(device=dev).manual_seed(0)
img = torch.rand((args.frames, args.h, args.w, 3), device=dev, generator=g) - 0.5


def conv_flops():
    """2 * MACs of every conv of the backbone for one frame (real channel counts; the stem's zero-padded
    channels are executed but not counted)."""
    def out(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = out(args.h, 7, 2, 3), out(args.w, 7, 2, 3)
    fl = 2.0 * h * w * 64 * 3 * 49
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    cin, cout = 64, 256
    for i, nb in enumerate(tspn.ResNetC4.BLOCKS[args.depth]):
        for b in range(nb):
            s = 2 if (b == 0 and i > 0) else 1
            mid = cout // 4
            h2, w2 = out(h, 1, s, 0), out(w, 1, s, 0)
            fl += 2.0 * h2 * w2 * (mid * cin + 9 * mid * mid + cout * mid + (cout * cin if cin != cout else 0))
            h, w, cin = h2, w2, cout
        cout *= 2
    return fl


fl = conv_flops() * args.frames
out = net(img, bf16=args.bf16)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
for a, b in evs:
    a.record()
    out = net(img, bf16=args.bf16)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
peak = 2500.0 if args.bf16 else 157.3
print(f"ResNet-{args.depth}-C4 [{'bf16' if args.bf16 else 'fp32'}] {args.frames} frames {args.h}x{args.w}: {ms:.1f} ms = "
      f"{ms / args.frames:.2f} ms per frame, {args.frames / ms * 1e3:.1f} frames/s, {fl / args.frames / 1e9:.1f} GFLOP per frame, "
      f"{fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / peak * 100:.1f} % of the {'bf16' if args.bf16 else 'fp32'} MFMA peak); "
      f"res4 {tuple(out.shape)} {out.dtype}", flush=True)
