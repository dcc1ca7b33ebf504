#!/usr/bin/env python3
"""Forward time of the 784-var QuadTree circuit at other unit counts:  python scripts/bench_k.py K [B] [region_graph] [layer]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

K = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rg = sys.argv[3] if len(sys.argv) > 3 else "quad-tree-2"
sp = sys.argv[4] if len(sys.argv) > 4 else "cp"
plan = image_data((1, 28, 28), rg, input_layer="categorical", num_input_units=K, sum_product_layer=sp, num_sum_units=K)
hc = HipCircuit(plan, init_plan_tensors(plan), device="cuda:0")
x = torch.randint(0, 256, (B, 784)).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        hc(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(s)
        hc(x)
        b.record(s)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    rows = hc.profile_kernels(x, iters=3)
if os.environ.get("ROWS"):
    for r, l in ((r, plan.layers[r["layer"]]) for r in rows):
        print(f"  layer {r['layer']:3d} {l.type:10s} F={l.num_folds:5d} H={l.arity} {l.num_input_units}->{l.num_output_units}  {r['kernel']:40s} {r['ms']:.4f} ms")
agg = {}
for r in rows:
    agg[r["kernel"]] = agg.get(r["kernel"], 0.0) + r["ms"]
alg = plan.algorithmic_bytes(B)["total"]
flops = sum(2.0 * l.num_folds * B * l.num_output_units *
            (l.num_input_units ** l.arity if l.type == "tucker" else l.num_input_units * (l.arity if l.type == "sum" else 1))
            for l in plan.layers if l.type in ("sum", "cpt", "tucker"))
print(f"K={K} B={B} {rg} {sp}: {ms:.3f} ms  {B / ms * 1e3:.3e} evals/s  {alg / ms / 1e6:.0f} GB/s algorithmic  "
      f"{flops / ms / 1e9:.1f} TFLOP/s  kernels {dict(sorted(((k, round(v, 3)) for k, v in agg.items()), key=lambda kv: -kv[1])[:4])}")
