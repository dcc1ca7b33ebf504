"""Experiment: the parameter prologue of every step on a low-priority side stream, beside the forward (leaf + 16-row tail) on the
main stream, no events between them in the steady state."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data

dev = torch.device("cuda:0")
plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32, sum_product_layer="cp", num_sum_units=32)
t = init_plan_tensors(plan)
g = torch.Generator().manual_seed(0)
xs = [torch.randint(0, 256, (4096, 784), generator=g).to(dev) for _ in range(12)]
print("priority range", torch.cuda.Stream.priority_range())
lo, hi = torch.cuda.Stream.priority_range()


def run(hc, side, steps=400, warm=2500):
    main = torch.cuda.Stream(dev, priority=int(os.environ.get("MAINPRIO", hi)))
    def step(k):
        if side is not None:
            hc._launch_param_batch(side.cuda_stream)
        hc.log_likelihood_sum(xs[k % 12])
    with torch.cuda.stream(main):
        for k in range(warm): step(k)
        torch.cuda.synchronize()
        res = []
        for r in range(5):
            t0 = time.perf_counter()
            for k in range(steps): step(k)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
    return res

base = HipCircuit(plan, t, device=dev)
print("default (params_at_end)", [round(v, 5) for v in run(base, None)])
cp = HipCircuit(plan, t, device=dev, cache_params=True)
print("cache_params alone      ", [round(v, 5) for v in run(cp, None)])
for pr in sorted({lo, 0}):
    side = torch.cuda.Stream(dev, priority=pr)
    print(f"cache_params + prologue on a side stream (priority {pr})", [round(v, 5) for v in run(cp, side)])
ref = base.log_likelihood_sum(xs[0]).clone(); got = cp.log_likelihood_sum(xs[0]).clone(); torch.cuda.synchronize()
print("same result", torch.equal(ref, got))
