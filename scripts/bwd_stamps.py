#!/usr/bin/env python3
"""Where is a wave of the fused backward launch when?  Builds a copy of the library with -DCK_BWD_STAMPS (ck_leaf_bwd.hip:
shader-clock stamps of the first 16 units of every wave of one workgroup) and runs fused training steps of the north-star
circuit:

    [CK_BWD_WAVES=4|8] [CK_BWD_STAMP_TOP=1] python scripts/bwd_stamps.py [workgroup]

Stamps per unit, 4-wave (software-pipelined) form: 0 iteration begins, 1 raw tiles consumed (= they had arrived), 2 next unit's
loads and the previous unit's stores issued, 3 node P done, 4 Q0 done, 5 Q1 done.  8-wave form: 0 begins, 1 loads issued,
2 gradient / kept tile of P arrived, 3 node P done, 4 Q0, 5 Q1, 6 stores issued.  The product library is not touched."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import build as B  # noqa: E402

wg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tmp = tempfile.mkdtemp(prefix="ckbstamps")
B.build(verbose=False)
obj = os.path.join(tmp, "ck_leaf_bwd.o")
subprocess.check_call([B.HIPCC, *B.FLAGS, "-w", "-DCK_BWD_STAMPS", "-c", os.path.join(B.SRC, "ck_leaf_bwd.hip"), "-o", obj])
objs = [o for o in sorted(os.listdir(B.LIB_DIR)) if o.endswith(".o") and o != "ck_leaf_bwd.o"]
lib = os.path.join(tmp, "libcirkit_hip_bstamps.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *[os.path.join(B.LIB_DIR, o) for o in objs]])

import numpy as np  # noqa: E402
import torch  # noqa: E402
from cirkit_amd import _capi  # noqa: E402

print("lab library:", lib)  # (CK_LIB=<this> python scripts/bench_train.py ... runs the timing experiments of scripts/exp_leaf_bwd.sh)

_capi._LIB_PATH = lib
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402
from cirkit_amd.training import HipTrainer  # noqa: E402

dev = torch.device("cuda:0")
waves = int(os.environ.get("CK_BWD_WAVES", "4"))
buf = torch.zeros(8 * 16 * 8, dtype=torch.int64, device=dev)
os.environ["CK_BWD_STAMP_PTR"] = str(buf.data_ptr())
os.environ["CK_BWD_STAMP_WG"] = str(wg)
plan = image_data((1, 28, 28), "quad-tree-2", num_input_units=32, num_sum_units=32)
tr = HipTrainer(plan, init_plan_tensors(plan), device=dev, lr=0.01, fused=True)
x = torch.randint(0, 256, (4096, 784), generator=torch.Generator().manual_seed(0)).to(dev)
for _ in range(200):
    tr.step(x)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(8, 16, 8)
n_id = 6 if waves == 4 else 7
print(f"workgroup {wg}, {'top' if os.environ.get('CK_BWD_STAMP_TOP') else 'leaf'} launch, {waves} waves; shader cycles (2.4 GHz: 1000 cycles = 0.42 us)")
for w in range(waves):
    units = [u for u in range(16) if s[w, u, 0] != 0]
    print(f"wave {w}: {len(units)} units stamped")
    for u in units:
        row = s[w, u]
        gaps = [int(row[i + 1] - row[i]) for i in range(n_id - 1)]
        nxt = int(s[w, u + 1, 0] - row[n_id - 1]) if u + 1 in units else -1
        print(f"   unit {u:2d}: begins at {int(row[0] - s[w, units[0], 0]):8d}  gaps {gaps}  -> next begins +{nxt}")
