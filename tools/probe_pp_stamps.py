#!/usr/bin/env python3
"""Cycle shares of a phase of conv3_bf16_pp_kernel (diagnostic build -DTSPN_PP_STAMP, TSPN_LIB_PATH): per wave of one
workgroup, summed over its 192 phases: [retire reads + DMA issue | fragment-read issue | counted vmcnt wait | wait at the
first barrier | MFMA cluster | wait at the second barrier]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
os.environ["TSPN_CONV3_BF16_PP"] = "1"
N, T, D, videos = 64, 900, 1024, 4
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = tspn.ops.cast_bf16(torch.rand((videos * N, T, D), device=dev, generator=g))
w = (torch.rand((2 * D, 2 * D, 3), device=dev, generator=g) - 0.5) * 0.02
packed = tspn.ops.pack_conv3_bf16(w, split=D)
for _ in range(3):
    y = tspn.ops.conv3_tc_bf16(x, packed)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["TSPN_LIB_PATH"])
buf = (ctypes.c_ulonglong * 128)()
assert lib.tspn_debug_pp_stamps(buf) == 0
names = ["retire+DMA", "read issue", "vmcnt wait", "barrier 1", "cluster", "barrier 2"]
nph = 3 * D // 16
for wv in range(8):
    v = [buf[wv * 8 + i] for i in range(6)]
    tot = sum(v)
    print(f"wave {wv}: " + "  ".join(f"{n} {a / nph:6.0f}" for n, a in zip(names, v)) + f"   = {tot / nph:6.0f} cycles per phase")
