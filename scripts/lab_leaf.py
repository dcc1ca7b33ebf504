#!/usr/bin/env python3
"""Leaf-kernel lab (GPU box): the north-star circuit under several HipCircuit configurations.

    python scripts/lab_leaf.py [B] [steps] [name=kw,kw ...]

For every configuration: bitwise comparison of the circuit output with the first configuration, and the time of a step
(forward + device-side LL sum, HIP events around `steps` back-to-back steps, median of 5 rounds).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.
"""

from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data

CONFIGS = {
    "classic": dict(persistent_leaf=False),
    "default": dict(),
    "persistent": dict(persistent_leaf=True),
}


def main() -> None:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    names = sys.argv[3:] or list(CONFIGS)
    dev = torch.device("cuda:0")
    plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32,
                      sum_product_layer="cp", num_sum_units=32)
    tensors = init_plan_tensors(plan)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (B, 784), generator=g).to(dev)
    x[::7, ::5] = -1  # some marginalised entries
    stream = torch.cuda.Stream(dev)
    ref = None
    for name in names:
        hc = HipCircuit(plan, tensors, device=dev, **CONFIGS[name])
        with torch.cuda.stream(stream):
            y = hc(x).clone()
            torch.cuda.synchronize(dev)
            if ref is None:
                ref = y
            same = bool(torch.equal(y, ref))
            maxdiff = float((y - ref).abs().max())
            ll = hc.log_likelihood_sum(x).cpu()
            ll_err = abs(float(ll[0]) - float(y.double().sum())) / abs(float(y.double().sum()))
            for _ in range(10):
                hc.log_likelihood_sum(x)
            rounds = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0.record(stream)
                for _ in range(steps):
                    hc.log_likelihood_sum(x)
                e1.record(stream)
                torch.cuda.synchronize(dev)
                rounds.append(e0.elapsed_time(e1) / steps)
        print(json.dumps({"config": name, "B": B, "bit_identical_to_first": same, "max_abs_diff": maxdiff,
                          "mean_ll": float(y.mean()), "ll_sum_rel_err": ll_err, "ll_count": float(ll[1]), "ms_per_step_median": float(np.median(rounds)),
                          "ms_per_step_min": float(min(rounds))}), flush=True)
        del hc


if __name__ == "__main__":
    main()
