#!/usr/bin/env python3
"""When does each block of the launch that ends a forward (tail of forward k + parameters of forward k + 1, ck_tailp.hip) run?
Builds a copy of the library with -DCK_TAILP_STAMPS (100 MHz wall-clock stamps per block: entry, each level of a tail block,
exit), runs the north-star step and prints the distribution per kind of block.  The product library is not touched.

    python scripts/tailp_stamps.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import build as B  # noqa: E402

tmp = tempfile.mkdtemp(prefix="ckstamps")
B.build(verbose=False)
objs = []
for name in ("ck_tailp", "ck_leaf", "ck_tail16"):  # (everything that includes ck_tailwalk.h sees the same header)
    obj = os.path.join(tmp, name + ".o")
    flags = ["-DCK_TAILP_STAMPS"] if name == "ck_tailp" else []
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-w", *flags, "-c", os.path.join(B.SRC, name + ".hip"), "-o", obj])
    objs.append(obj)
rest = [os.path.join(B.LIB_DIR, o) for o in sorted(os.listdir(B.LIB_DIR)) if o.endswith(".o") and o[:-2] not in ("ck_tailp", "ck_leaf", "ck_tail16")]
lib = os.path.join(tmp, "libcirkit_hip_stamps.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, *rest])

import torch  # noqa: E402
from cirkit_amd import _capi  # noqa: E402

_capi._LIB_PATH = lib
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

dev = torch.device("cuda:0")
NB = 1024
buf = torch.zeros(NB * 16, dtype=torch.int64, device=dev)
os.environ["CK_STAMP_PTR"] = str(buf.data_ptr())
plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32, sum_product_layer="cp", num_sum_units=32)
hc = HipCircuit(plan, init_plan_tensors(plan), device=dev)
g = torch.Generator().manual_seed(0)
xs = [torch.randint(0, 256, (4096, 784), generator=g).to(dev) for _ in range(12)]
for k in range(3000):
    hc.log_likelihood_sum(xs[k % 12])
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(NB, 16).astype(np.float64) / 100.0
used = s[:, 0] > 0
t0 = s[used, 0].min()
s = np.where(s > 0, s - t0, np.nan)
n_tail, n_pair = 256, 392
kinds = (("tail tiles", slice(0, n_tail)), ("table pairs", slice(n_tail, n_tail + n_pair)), ("32-wide softmaxes", slice(n_tail + n_pair, int(used.sum()))))
print(f"{int(used.sum())} blocks; us since the first block entered (min / median / max)")
def row(name, v):
    v = v[~np.isnan(v)]
    if len(v):
        print(f"  {name:34s} {v.min():6.2f} {np.median(v):6.2f} {v.max():6.2f}")
for name, sl in kinds:
    print(name)
    row("entry", s[sl, 0])
    if name == "tail tiles":
        row("descriptors in LDS", s[sl, 1])
        prev = s[sl, 1]
        for li in range(6):
            row(f"level {li + 1} done", s[sl, 2 + li])
            row(f"   (took)", s[sl, 2 + li] - prev)
            prev = s[sl, 2 + li]
    row("exit", s[sl, 15])
    row("lifetime", s[sl, 15] - s[sl, 0])
