"""Experiment: how much of the leaf launch is waiting for table rows?  The same forward on batches whose table rows are (a) 256
different rows per variable (random pixels: the bench), (b) ONE row per variable (every pixel 0: all gathers of a leaf hit one cached
row), (c) 4 / 16 distinct values per variable.  Step time with cached parameters (the leaf launch + the 16-row tail)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data

dev = torch.device("cuda:0")
plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32, sum_product_layer="cp", num_sum_units=32)
t = init_plan_tensors(plan)
hc = HipCircuit(plan, t, device=dev)
g = torch.Generator().manual_seed(0)


def run(xs, steps=400, warm=2500):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        for k in range(warm):
            hc.log_likelihood_sum(xs[k % len(xs)])
        torch.cuda.synchronize()
        res = []
        for r in range(5):
            t0 = time.perf_counter()
            for k in range(steps):
                hc.log_likelihood_sum(xs[k % len(xs)])
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(res)[2]


for name, hi in (("256 values", 256), ("16 values", 16), ("4 values", 4), ("1 value", 1)):
    xs = [torch.randint(0, hi, (4096, 784), generator=g).to(dev) for _ in range(12)]
    print(f"{name:12s} {run(xs):.5f} ms/step")
