#!/usr/bin/env python3
"""What the host-input path is made of on this box: PCIe H2D / D2H rates from pinned and pageable memory, the host
copy pageable -> pinned (torch's copy_ and a thread pool of row copies), the price of a pinned allocation."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

dev = torch.device("cuda", 0)
MB = 1 << 20
n = 157 * MB // 4            # one chunk of 4 cfg2 videos, fp32
out = {"cores": len(os.sched_getaffinity(0)), "torch_threads": torch.get_num_threads()}


def tm(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


page = torch.rand(n)
t0 = time.perf_counter()
pin = torch.empty(n, pin_memory=True)
out["pinned_alloc_ms_157MB"] = (time.perf_counter() - t0) * 1e3
del pin
t0 = time.perf_counter()
pin = torch.empty(n, pin_memory=True)
out["pinned_realloc_ms_157MB"] = (time.perf_counter() - t0) * 1e3
d = torch.empty(n, device=dev)
gb = n * 4 / 1e9
out["h2d_pinned_GBps"] = gb / tm(lambda: d.copy_(pin, non_blocking=True))
out["h2d_pageable_GBps"] = gb / tm(lambda: d.copy_(page))
out["d2h_pinned_GBps"] = gb / tm(lambda: pin.copy_(d, non_blocking=True))
out["d2h_pageable_GBps"] = gb / tm(lambda: page.copy_(d))
out["host_copy_to_pinned_GBps"] = gb / tm(lambda: pin.copy_(page))
page2 = torch.rand(n)
out["host_copy_pageable_GBps"] = gb / tm(lambda: page2.copy_(page))
for nt in (4, 8, 16):
    pool = ThreadPoolExecutor(nt)
    pv, gv = pin.view(nt, -1), page.view(nt, -1)

    def par():
        list(pool.map(lambda k: pv[k].copy_(gv[k]), range(nt)))
    out[f"host_copy_to_pinned_{nt}threads_GBps"] = gb / tm(par)
    pool.shutdown()
# both directions at once
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pin2 = torch.empty(n, pin_memory=True)
d2 = torch.rand(n, device=dev)


def duplex():
    with torch.cuda.stream(s1):
        d.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        pin2.copy_(d2, non_blocking=True)
out["duplex_each_GBps"] = gb / tm(duplex)
print(json.dumps(out, indent=1))
