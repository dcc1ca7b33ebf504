#!/usr/bin/env python3
"""Per-launch table of the job form of a training step (cirkit_amd/train_jobs.py): kind, units, jobs, splits, list lengths and
the launch's time when it is issued alone (HIP events, 20 repetitions after the whole step has run once).
    python scripts/jobs_levels.py [B] [config]      config 6: QuadGraph CP K = 64 (the reference's learning notebook); 4: BASELINE config 4"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cirkit_amd import _capi as capi  # noqa: E402

if os.environ.get("CK_LIB"):  # a lab build of the library
    capi._LIB_PATH = os.environ["CK_LIB"]
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402
from cirkit_amd.training import HipTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if cfg == 4:
    plan = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
    x = torch.randn(B, 784).cuda()
else:
    plan = image_data((1, 28, 28), "quad-graph", input_layer="categorical", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
    x = torch.randint(0, 256, (B, 784)).cuda()
tr = HipTrainer(plan, init_plan_tensors(plan), device="cuda:0", lr=0.01, jobs=True)
js = tr._jobs
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
st = js.bind(B)
MODE = int(os.environ.get("MODE", "2"))  # 2: the optimizer inside the epilogues (what `step` runs on one rank); 1: d theta only
OPT = js._opt_state().data_ptr() if MODE == 2 else None
pool = st["pool"].data_ptr()
blk = B * 64
stream = torch.cuda.current_stream().cuda_stream
total = 0.0
print(f"{'launch':10s} {'units':>6s} {'jobs':>6s} {'split':>5s} {'n_in':>9s} {'n_g':>9s} {'us':>8s}")
for la in st["launches"]:
    what = la[0]
    if what == "mix_params":
        continue
    if what == "root":
        fn = lambda: capi.call("ck_jobs_root", C.byref(st["root"][MODE]), stream)
        desc = (1, 1, 1, "", "")
    else:
        tabs = la[2] if what in ("input_bwd", "gauss_bwd", "cat_bwd") else la[1]
        n = la[3] if what in ("input_bwd", "gauss_bwd", "cat_bwd") else la[2]
        tab = tabs.get(MODE, tabs[1])
        raw = tab.cpu().numpy()
        if what == "cat_bwd":
            t = raw.view(np.dtype(capi.CAT_JOB_DTYPE)).reshape(-1)
            desc = (n, n, 1, "", f"{t['n_g'].mean():.1f}/{t['n_g'].max()}")
            fn = (lambda tab=tab, n=n, Cn=tr.circuit.layers[la[1]].num_categories: capi.call("ck_jobs_cat_bwd", tab.data_ptr(), n, pool, B, Cn, OPT, stream))
        elif what == "gauss_bwd":
            t = raw.view(np.dtype(capi.GAUSS_JOB_DTYPE)).reshape(-1)
            desc = (n, n, 1, "", f"{t['n_g'].mean():.1f}/{t['n_g'].max()}")
            fn = (lambda tab=tab, n=n: capi.call("ck_jobs_gauss_bwd", tab.data_ptr(), n, pool, B, OPT, stream))
        elif what in ("nsum", "input_bwd"):
            t = raw.view(np.dtype(capi.NSUM_JOB_DTYPE)).reshape(-1)
            desc = (n, n, 1, f"{t['n_in'].mean():.1f}/{t['n_in'].max()}", "")
            fn = (lambda tab=tab, n=n: capi.call("ck_jobs_nsum", tab.data_ptr(), n, pool, blk, stream))
        elif what.startswith("sum"):
            t = raw.view(np.dtype(capi.SUM_JOB_DTYPE)).reshape(-1)
            ns = int(t["n_split"].max())
            desc = (n, int((t["split"] == 0).sum()), ns, f"{t['n_in'].mean():.1f}/{t['n_in'].max()}", f"{t['n_g'].mean():.1f}/{t['n_g'].max()}")
            name = "ck_jobs_sum64_fwd" if what == "sum_fwd" else "ck_jobs_sum64_bwd"
            fn = ((lambda tab=tab, n=n: capi.call("ck_jobs_sum64_fwd", tab.data_ptr(), n, pool, stream)) if what == "sum_fwd" else
                  (lambda tab=tab, n=n, wv=la[3]: capi.call("ck_jobs_sum64_bwd", tab.data_ptr(), n, pool, OPT, wv, stream)))
            what = what + (f"/{la[3]}w" if what == "sum_bwd" else "")
        else:
            t = raw.view(np.dtype(capi.MIX_JOB_DTYPE)).reshape(-1)
            ns = int(t["n_split"][0])
            desc = (n, n // ns, ns, f"H{t['H'].mean():.1f}/{t['H'].max()} S{t['S'].max()}", f"{t['n_g'].mean():.1f}/{t['n_g'].max()}")
            hm = la[3]
            fn = ((lambda tab=tab, n=n, hm=hm: capi.call("ck_jobs_mix_fwd", tab.data_ptr(), n, pool, hm, stream)) if what == "mix_fwd" else
                  (lambda tab=tab, n=n, hm=hm: capi.call("ck_jobs_mix_bwd", tab.data_ptr(), n, pool, hm, blk, OPT, stream)))
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    total += us
    print(f"{what:10s} {desc[0]:6d} {desc[1]:6d} {desc[2]:5d} {desc[3]:>9s} {desc[4]:>9s} {us:8.1f}")
print(f"sum of the job launches {total:.0f} us  ({len(st['launches'])} launches; the recorded step has {js.num_launches(B, MODE)})")
