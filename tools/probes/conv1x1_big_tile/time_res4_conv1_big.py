#!/usr/bin/env python3
"""Scratch timing (needs tools/probes/conv1x1_big_tile/tspn_gemm_bf16.hip copied into csrc/ and a rebuild): res4's 1x1 convs
(1024 -> 256 on 45 x 80 maps) on the generic kernel (256 x 128 tiles, 2 per CU) against the 256 x 256 big-tile GEMM of round 4
(1 tile per CU), 9 / 18 / 36 frames."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tspn_mi355x as tspn
lib = ctypes.CDLL(tspn._abi.LIB_PATH)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.tspn_pack_conv1x1_rows_bf16.argtypes = [vp, i64, i64, vp, vp]
lib.tspn_conv1x1_big_bf16.argtypes = [vp, i64, i64, i64, i64, vp, i64, i64, vp, vp, ctypes.c_int, vp, vp]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


for frames in (9, 18, 36):
    for cin, cout in ((1024, 256),):
        x = torch.rand((frames, 45, 80, cin), device=dev, generator=g).to(torch.bfloat16)
        w = (torch.rand((cout, cin, 1, 1), device=dev, generator=g) - 0.5) * 0.05
        b = torch.rand(cout, device=dev, generator=g) - 0.5
        f = tspn.ops.pack_conv2d_frag_bf16(w)
        rows = torch.empty((cin // 8, cout, 8), dtype=torch.bfloat16, device=dev)
        assert lib.tspn_pack_conv1x1_rows_bf16(w.data_ptr(), cout, cin, rows.data_ptr(), st()) == 0
        out = torch.empty((frames, 45, 80, cout), dtype=torch.bfloat16, device=dev)
        gen = lambda: tspn.ops.conv2d_nhwc_bf16(x, f, (1, 1), 1, 0, bias=b, relu=True)
        def big():
            rc = lib.tspn_conv1x1_big_bf16(x.data_ptr(), frames, 45, 80, cin, rows.data_ptr(), cout, 1, b.data_ptr(), None, 1, out.data_ptr(), st())
            assert rc == 0, rc
        big(); torch.cuda.synchronize()
        same = torch.equal(gen(), out)
        ug, ub = timeit(gen), timeit(big)
        fl = 2.0 * frames * 3600 * cin * cout
        print(f"{frames} frames {cin}->{cout}: generic {ug:.1f} us ({fl / ug / 1e6:.0f} TFLOP/s)  big tile {ub:.1f} us ({fl / ub / 1e6:.0f} TFLOP/s)  equal {same}", flush=True)
