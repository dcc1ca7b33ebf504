#!/usr/bin/env python3
"""Per-layer kernel times of a committed fixture plan:  python scripts/layer_times.py <fixture> <B>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import Plan  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
plan = Plan.load(os.path.join(ROOT, "tests", "golden", name))
hc = HipCircuit(plan, init_plan_tensors(plan), device="cuda:0")
if any(l.type == "gaussian" for l in plan.layers):
    x = torch.randn((B, plan.num_variables)).cuda()
else:
    x = torch.randint(0, 256, (B, plan.num_variables)).cuda()
for r in hc.profile_kernels(x, iters=5):
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
