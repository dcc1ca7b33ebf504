#!/usr/bin/env python3
"""Time any committed fixture plan on the GPU box:  python scripts/bench_plan.py <fixture> <B> [steps]
Prints ms/forward (hipGraph replay), algorithmic GB/s and the per-kernel event breakdown."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import Plan  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("=")
    kw[k] = {"True": True, "False": False}.get(v, int(v) if v.isdigit() else v)
plan = Plan.load(os.path.join(ROOT, "tests", "golden", name))
tensors = init_plan_tensors(plan)
hc = HipCircuit(plan, tensors, device="cuda:0", **kw)
g = torch.Generator().manual_seed(0)
if plan.num_variables == 0:
    x = None
elif any(l.type == "gaussian" for l in plan.layers):
    x = torch.randn((B, plan.num_variables), generator=g).cuda()
else:
    x = torch.randint(0, 256, (B, plan.num_variables), generator=g).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        hc(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(s)
        hc(x)
        b.record(s)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    rows = hc.profile_kernels(x, iters=5)
alg = plan.algorithmic_bytes(B if x is not None else 1)
agg = {}
for r in rows:
    a = agg.setdefault(r["kernel"], [0.0, 0.0, 0])
    a[0] += r["ms"]; a[1] += r["algorithmic_bytes"]; a[2] += 1
print(json.dumps({
    "plan": name, "B": B, "ms": ms, "evals_per_s": (B / ms * 1e3) if x is not None else None,
    "algorithmic_GB": alg["total"] / 1e9, "algorithmic_GB_per_s": alg["total"] / ms / 1e6,
    "launches": hc.num_launches(B if x is not None else 1),
    "kernels": {k: {"ms": round(v[0], 4), "n": v[2], "GB/s": round(v[1] / max(v[0], 1e-9) / 1e6, 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])},
}))
