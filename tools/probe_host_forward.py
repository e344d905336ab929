#!/usr/bin/env python3
"""cProfile of bench.py's host-input step (where the host thread spends its time): python tools/probe_host_forward.py pageable|pinned"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import tspn_mi355x as tspn

mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
args = bench.parse(["--host-inputs", mode, "--steps", "8", "--warmup", "3", "--no-cpu-baseline"] + sys.argv[2:])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gather = bench.Gatherer(tspn, args, 1, False, rank=0, torch=torch, on_gpu=True)
wl = bench.ScoringWorkload(args, tspn, torch, np, dev, 1, 0, gather)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(3, 11):
    wl.step(i)
torch.cuda.synchronize()
pr.disable()
print("ms per step", (time.perf_counter() - t0) / 8 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
