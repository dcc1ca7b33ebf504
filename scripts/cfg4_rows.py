import sys, os, json, torch, numpy as np
sys.path.insert(0, "/root/repo")
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.plan import Plan
plan = Plan.load("/root/repo/tests/golden/cfg4_pd784")
hc = HipCircuit(plan, init_plan_tensors(plan), device="cuda:0")
x = torch.randn((4096, 784), generator=torch.Generator().manual_seed(0)).cuda()
for _ in range(3): hc(x)
rows = hc.profile_kernels(x, iters=10)
for r in rows:
    print(f"{r['ms']*1000:8.1f} us  layer={r.get('layer')}  {r['kernel'][:60]:60s} flops={r.get('flops',0)/1e9:8.2f}G exec={r.get('exec',0)/1e9:8.2f}G bytes={r.get('algorithmic_bytes',0)/1e6:8.1f}MB  {r.get('note','')}")
for i, l in enumerate(hc.layers):
    print(i, type(l).__name__, getattr(l, "num_folds", None), getattr(l, "arity", None), getattr(l, "num_input_units", None), getattr(l, "num_output_units", None))
