#!/usr/bin/env python3
"""Per-call latency and host-side cost of HipCircuit at small batches:  python scripts/latency.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

plan = image_data((1, 28, 28), "quad-tree-2", num_input_units=32, num_sum_units=32)
tensors = init_plan_tensors(plan)
for B in [int(a) for a in sys.argv[1:]] or [1, 32, 256, 4096]:
    hc = HipCircuit(plan, tensors, device="cuda:0", cache_params=True)
    x = torch.randint(0, 256, (B, 784)).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20):
            hc(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            hc(x)
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t) / 200
        t = time.perf_counter()
        for _ in range(1000):
            hc(x)
        host = (time.perf_counter() - t) / 1000
        torch.cuda.synchronize()
        thr = (time.perf_counter() - t) / 1000
    print(f"B={B}: latency {lat * 1e6:.1f} us/call (sync each), host enqueue {host * 1e6:.1f} us/call, back-to-back {thr * 1e6:.1f} us/call "
          f"-> {B / thr:.3e} evals/s (derived parameters cached)")
