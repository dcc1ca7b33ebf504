#!/usr/bin/env python3
"""The partition function Z = integral |c|^2 of BASELINE config 5's squared circuit (built natively from the plan of c,
cirkit_amd/functional.py) as one HipCircuit forward: python scripts/bench_partition.py [fuse: 0|1]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.functional import squared_partition_plan
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
plan5 = image_data((1, 28, 28), "quad-tree-2", input_layer="embedding", num_input_units=32, sum_product_layer="cp-t",
                   num_sum_units=32, sum_weight_activation="none", semiring="complex-lse-sum")
fuse = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
hz = HipCircuit(squared_partition_plan(plan5), init_plan_tensors(plan5), device="cuda:0", fuse=fuse)
for _ in range(3): y = hz(None)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); y = hz(None); b.record()
torch.cuda.synchronize()
print("fuse", fuse, "Z forward ms", float(np.median([a.elapsed_time(b) for a, b in ev])), "launches", hz.num_launches(1), "Z", complex(y.reshape(-1)[0]))
