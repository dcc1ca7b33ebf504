#!/usr/bin/env python3
"""Fused vs layer-wise training step on the north-star circuit (GPU box):
    python scripts/cmp_fused_train.py [B] [steps]
prints the largest gradient difference per tensor (relative to the tensor's largest gradient) and both step times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402
from cirkit_amd.training import HipTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
plan = image_data((1, 28, 28), "quad-tree-2", num_input_units=32, num_sum_units=32)
tensors = init_plan_tensors(plan)
x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(B)).cuda()
a = HipTrainer(plan, tensors, device="cuda:0", lr=0.01, fused=False)
b = HipTrainer(plan, tensors, device="cuda:0", lr=0.01, fused=True)
la = a.loss_and_grads(x).clone()
lb = b.loss_and_grads(x).clone()
torch.cuda.synchronize()
print("LL sum layer-wise", float(la[0]), "fused", float(lb[0]), "rel diff", abs(float(la[0] - lb[0])) / abs(float(la[0])))
worst = 0.0
for k in plan.tensors:
    ga, gb = a.grads[k].double(), b.grads[k].double()
    sc = float(ga.abs().max())
    err = float((ga - gb).abs().max())
    worst = max(worst, err / max(sc, 1e-30))
    print(f"{k:12s} shape {tuple(ga.shape)!s:18s} max|g| {sc:.3e}  max diff {err:.3e}  rel {err / max(sc, 1e-30):.2e}  "
          f"norm a {float(ga.norm()):.6e} b {float(gb.norm()):.6e}")
print("worst relative difference", worst)
redo = b.circuit._bind(B).keep[b._fz["group"].root][1]
print("flagged tiles", int(redo.sum()))
for name, tr in (("layer-wise", a), ("fused", b)):
    for _ in range(3):
        tr.step(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        tr.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(f"{name}: train step {dt * 1e3:.3f} ms  {B / dt:.3e} samples/s")
