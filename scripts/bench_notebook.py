#!/usr/bin/env python3
"""The reference's compilation-options notebook configuration (QuadGraph 28x28, Categorical-256, Tucker, K = 64, batch 128):
forward time with the Tucker weights normalised online by the stream-K launch and with the prologue writing them.
python scripts/bench_notebook.py [B] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import _capi  # noqa: E402

if os.environ.get("CK_LIB"):  # a lab build of the library (scripts/exp_tucker_bf16.py)
    _capi._LIB_PATH = os.environ["CK_LIB"]
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
plan = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64, sum_product_layer="tucker",
                  num_sum_units=64)
tensors = init_plan_tensors(plan)
x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(0)).to(dev)
outs = {}
# (fused, contraction): the product, the prologue-normalised form, and the labelled bf16-split variants of the stream-K launch
for fused in ((True, False, "bf16x3", "bf16x6") if not os.environ.get("ONLY") else [{"True": True, "False": False}.get(v, v) for v in os.environ["ONLY"].split(",")]):
    hc = (HipCircuit(plan, tensors, device=dev, fused_weight_softmax=True, contraction=fused) if isinstance(fused, str)
          else HipCircuit(plan, tensors, device=dev, fused_weight_softmax=fused))
    for _ in range(5):
        y = hc(x)
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            y = hc(x)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / steps)
    outs[fused] = y.double().cpu()
    print(f"{'contraction' if isinstance(fused, str) else 'fused_weight_softmax'}={fused}: {sorted(times)[2]:.3f} ms / forward (median of 5 x {steps}), mean LL {float(y.mean()):.4f}", flush=True)
    if os.environ.get("KERNELS"):
        for r in sorted(hc.profile_kernels(x, 10), key=lambda r: -r["ms"])[:10]:
            print(f"    layer {r['layer']:3d} {r['kernel'][:70]:70s} {r['ms']:.4f} ms")
    del hc
if os.environ.get("ONLY"):
    sys.exit(0)
d = (outs[True] - outs[False]).abs().max() / outs[False].abs().max()
print(f"max relative difference between the two: {float(d):.2e}")
for c in ("bf16x3", "bf16x6"):
    print(f"max relative difference of contraction={c} to the exact launch: {float(((outs[c] - outs[True]).abs() / outs[True].abs()).max()):.2e}")
