#!/usr/bin/env python3
"""Ablation of the fused leaf kernel (GPU box): time the fused launch with parts switched off."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import _capi as capi  # noqa: E402
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.plan import Plan  # noqa: E402

plan = Plan.load(os.path.join(ROOT, "tests", "golden", "cfg2_qt784"))
tensors = init_plan_tensors(plan)
B = int(os.environ.get("B", 4096))
x = torch.randint(0, 256, (B, 784)).cuda()
names = {0: "full", 0x100: "full tpw1", 0x200: "full tpw2", 0x400: "full tpw4", 0x800: "full tpw8", 0x10b: "MFMA only tpw1", 0x40b: "MFMA only tpw4", 0x10f: "nothing tpw1", 0x40f: "nothing tpw4",
         0x403: "noW+noGather tpw4", 0x401: "noW tpw4", 0x402: "noGather tpw4", 1: "noW", 2: "noGather", 4: "noMFMA", 8: "noExpLog", 3: "noW+noGather", 12: "noMFMA+noExpLog",
         7: "noW+noGather+noMFMA", 11: "MFMA only", 15: "nothing"}
for depth in [int(a) for a in sys.argv[1:]] or [2, 3]:
    for mask, label in names.items():
        capi.call("ck_debug_ablate", mask)
        hc = HipCircuit(plan, tensors, device="cuda:0", use_graph=False, fuse=depth)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            rows = hc.profile_kernels(x, iters=10)
        ms = [r["ms"] for r in rows if r["kernel"].startswith("subtree")][0]
        print(f"depth {depth} {label:24s} {1e3*ms:8.1f} us", flush=True)
capi.call("ck_debug_ablate", 0)
