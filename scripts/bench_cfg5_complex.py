#!/usr/bin/env python3
"""BASELINE config 5's c(x) (QuadTree-2, Embedding-256, CP-T, K = 32, complex-lse-sum, batch 4096) on the COMPLEX kernels
(`HipCircuit(signed_real=False)`: what circuits with complex-valued parameters take) next to the signed-tile path of real
parameters: forward time and the per-launch breakdown.  python scripts/bench_cfg5_complex.py [B] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
plan = image_data((1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t", num_sum_units=32,
                  sum_weight_activation="none", semiring="complex-lse-sum")
t = init_plan_tensors(plan)
x = torch.randint(0, 256, (B, 784), generator=torch.Generator().manual_seed(0)).cuda()
cases = [("linear (re, im) tiles depth %d" % d, dict(signed_real=False), d) for d in (2, 3, 4)]
cases += [("layer-wise complex kernels", dict(signed_real=False, complex_linear=False), None), ("signed tiles (real parameters)", {}, None)]
if os.environ.get("COMPLEX_W"):  # complex-valued sum weights and Embedding weights (random phases): four chains per contraction
    import numpy as np

    rng = np.random.default_rng(5)
    tc = {k: (np.asarray(v) * np.exp(1j * rng.uniform(-np.pi, np.pi, np.asarray(v).shape))).astype(np.complex64) for k, v in t.items()}
    cases = [("complex parameters, linear tiles depth %d" % d, dict(), d) for d in (2, 3)] + [("complex parameters, layer-wise kernels", dict(complex_linear=False), None)]
for label, kw, depth in cases:
    if os.environ.get("ONLY") and os.environ["ONLY"] not in label:
        continue
    if depth is not None:
        os.environ["CK_CLIN_DEPTH"] = str(depth)
    if os.environ.get("COMPLEX_W"):
        t_use = tc
    else:
        t_use = t
    hc = HipCircuit(plan, t_use, device="cuda:0", **kw)
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
    print(f"{label}: {sorted(times)[2]:.4f} ms per forward of {B} rows ({hc.num_launches(B)} launches)", flush=True)
    if os.environ.get("KERNELS"):
        for r in sorted(hc.profile_kernels(x, 10), key=lambda r: -r["ms"])[:12]:
            print(f"    layer {r['layer']:3d} {r['kernel'][:60]:60s} {1e3 * r['ms']:8.1f} us")
