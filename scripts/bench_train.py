#!/usr/bin/env python3
"""Training-step timing on the GPU box:
    python scripts/bench_train.py [B] [steps] [config] [fused|layerwise]
config 2 (default): 784-var QuadTree, Categorical, K = 32; config 4: Poon-Domingos, Gaussian, K = 64;
config 6: QuadGraph, Categorical, CP, K = 64 (the circuit of the reference's learning-a-circuit notebook, batch 256)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cirkit_amd import _capi  # noqa: E402

if os.environ.get("CK_LIB"):  # a lab build of the library (scripts/bwd_stamps.py leaves one and prints its path)
    _capi._LIB_PATH = os.environ["CK_LIB"]
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402
from cirkit_amd.training import HipTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if cfg == 4:
    plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64,
                      sum_product_layer="cp", num_sum_units=64)
    x = torch.randn(B, 784).cuda()
elif cfg == 6:
    plan = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64,
                      sum_product_layer="cp", num_sum_units=64)
    x = torch.randint(0, 256, (B, 784)).cuda()
else:
    plan = image_data((1, 28, 28), "quad-tree-2", num_input_units=32, num_sum_units=32)
    x = torch.randint(0, 256, (B, 784)).cuda()
mode = sys.argv[4] if len(sys.argv) > 4 else "auto"
tr = HipTrainer(plan, init_plan_tensors(plan), device="cuda:0", lr=0.01, fused={"auto": None, "fused": True, "layerwise": False, "jobs": None}[mode],
                jobs={"auto": None, "jobs": True}.get(mode, False))
print("fused" if tr.fused else ("job-list" if tr._jobs is not None else "layer-wise"), "training step",
      f"({tr._jobs.num_launches(B)} launches per step)" if tr._jobs is not None else "")
lls = []
for _ in range(3):
    ll = tr.step(x)
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(steps):
    ll = tr.step(x)
    if k in (0, steps - 1):
        lls.append(ll.clone())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"train step {dt * 1e3:.3f} ms  {B / dt:.3e} samples/s  mean LL first {float(lls[0][0] / lls[0][1]):.3f} last {float(lls[-1][0] / lls[-1][1]):.3f}")
