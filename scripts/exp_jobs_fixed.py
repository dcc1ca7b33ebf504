#!/usr/bin/env python3
"""Experiment: the fixed cost of a unit of the sum backward launch (cirkit_amd/csrc/ck_jobs.hip) -- everything but its row tiles.
A level of 512 jobs of the notebook circuit is launched with EMPTY row ranges (weights staged, accumulators reduced, epilogue
run), in mode 1 (d theta written) and mode 2 (optimizer in the epilogue), unsplit and with the partial-sum path (n_split = 2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cirkit_amd import _capi as capi  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402
from cirkit_amd.training import HipTrainer  # noqa: E402

B = 256
plan = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
x = torch.randint(0, 256, (B, 784)).cuda()
tr = HipTrainer(plan, init_plan_tensors(plan), device="cuda:0", lr=0.01, jobs=True)
js = tr._jobs
tr.step(x)
torch.cuda.synchronize()
st = js.bind(B)
pool = st["pool"].data_ptr()
opt = js._opt_state().data_ptr()
stream = torch.cuda.current_stream().cuda_stream
la = max((l for l in st["launches"] if l[0] == "sum_bwd"), key=lambda l: l[2])
dt = np.dtype(capi.SUM_JOB_DTYPE)


WAVES = int(os.environ.get("WAVES", "4"))


def timed(tab, n, o):
    for _ in range(3):
        capi.call("ck_jobs_sum64_bwd", tab.data_ptr(), n, pool, o, WAVES, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        capi.call("ck_jobs_sum64_bwd", tab.data_ptr(), n, pool, o, WAVES, stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20


for units in (64, 512, 1024):
    for mode in (1, 2):
        for empty in (False, True):
            for split in (1, 2):
                t = la[1][mode].cpu().numpy().view(dt).reshape(-1)[:units].copy()
                part = tick = None
                if split == 2:  # pairs of units become the two halves of one job
                    part = torch.zeros(units * 4096, dtype=torch.float32, device="cuda")
                    tick = torch.zeros(units, dtype=torch.int32, device="cuda")
                    for k in range(units):
                        j = k // 2
                        for f in ("w", "out", "gx", "dtheta", "theta", "m1", "m2", "w_out", "in_off", "n_in", "g_off", "n_g"):
                            t[k][f] = t[2 * j][f] if False else t[k - (k % 2)][f]
                        t[k]["row0"], t[k]["row1"] = (0, 128) if k % 2 == 0 else (128, 256)
                        t[k]["split"], t[k]["n_split"] = k % 2, 2
                        t[k]["part"], t[k]["ticket"] = part.data_ptr() + j * 2 * 4096 * 4, tick.data_ptr() + j * 4
                if empty:
                    t["row1"] = t["row0"]
                d = torch.from_numpy(t.view(np.uint8).reshape(units, -1)).cuda()
                us = timed(d, units, opt if mode == 2 else None)
                print(f"units {units:5d} mode {mode} {'empty rows' if empty else 'full rows '} n_split {split}: {us:7.1f} us")
