#!/usr/bin/env python3
"""Where is a wave of the leaf launch when?  Builds a copy of the library with -DCK_LEAF_STAMPS (ck_leaf.hip: shader-clock
stamps of one tile of every wave of one workgroup, kept in LDS, written out at the end), runs the north-star step, prints per
wave the cycles between the stamps:

    python scripts/leaf_stamps.py [workgroup] [tile-of-the-wave]

per leaf i: `wait` = leaf begins -> its rows are in registers (gather wait + slot read), `req` = -> the request of leaf i + 3 is out
(row indices through ds_bpermute + 5 vector-memory instructions), `steps` = -> the next leaf begins (products + the chains of
the 0..4 contraction steps behind leaf i, as ISSUED).  The product library is not touched."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import build as B  # noqa: E402

wg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = tempfile.mkdtemp(prefix="ckstamps")
B.build(verbose=False)
obj = os.path.join(tmp, "ck_leaf.o")
subprocess.check_call([B.HIPCC, *B.FLAGS, "-w", "-DCK_LEAF_STAMPS", "-c", os.path.join(B.SRC, "ck_leaf.hip"), "-o", obj])
objs = [o for o in sorted(os.listdir(B.LIB_DIR)) if o.endswith(".o") and o != "ck_leaf.o"]
lib = os.path.join(tmp, "libcirkit_hip_stamps.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *[os.path.join(B.LIB_DIR, o) for o in objs]])

import torch  # noqa: E402
from cirkit_amd import _capi  # noqa: E402

_capi._LIB_PATH = lib
from cirkit_amd.circuit import HipCircuit  # noqa: E402
from cirkit_amd.initializers import init_plan_tensors  # noqa: E402
from cirkit_amd.templates import image_data  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.zeros(8 * 64 + 8 * 256, dtype=torch.int64, device=dev)
os.environ["CK_STAMP_PTR"] = str(buf.data_ptr())
os.environ["CK_STAMP_WG"] = str(wg)
os.environ["CK_STAMP_TILE"] = str(nth)
plan = image_data((1, 28, 28), region_graph="quad-tree-2", input_layer="categorical", num_input_units=32, sum_product_layer="cp", num_sum_units=32)
hc = HipCircuit(plan, init_plan_tensors(plan), device=dev)
g = torch.Generator().manual_seed(0)
xs = [torch.randint(0, 256, (4096, 784), generator=g).to(dev) for _ in range(12)]
for k in range(3000):
    hc.log_likelihood_sum(xs[k % 12])
torch.cuda.synchronize()
allb = buf.cpu().numpy()
s = allb[:512].reshape(8, 64)
print(f"workgroup {wg}, tile {nth} of every wave; shader cycles (2.4 GHz: 1000 cycles = 0.42 us)")
t0 = s[:, 0].min()
for w in range(8):
    row = s[w]
    print(f"wave {w}: starts at {row[0] - t0:6d}, tile takes {row[49] - row[0]:6d} cycles (chains issued after {row[48] - row[0]}, stored +{row[49] - row[48]})")
    parts = []
    for i in range(16):
        nxt = row[3 * (i + 1)] if i < 15 else row[48]
        parts.append(f"{i:2d}: wait {row[3 * i + 1] - row[3 * i]:5d} req {row[3 * i + 2] - row[3 * i + 1]:5d} steps {nxt - row[3 * i + 2]:5d}")
    for a in range(0, 16, 4):
        print("    " + " | ".join(parts[a:a + 4]))
tot = {"wait": 0, "req": 0, "steps": 0}
for w in range(8):
    row = s[w]
    for i in range(16):
        nxt = row[3 * (i + 1)] if i < 15 else row[48]
        tot["wait"] += row[3 * i + 1] - row[3 * i]
        tot["req"] += row[3 * i + 2] - row[3 * i + 1]
        tot["steps"] += nxt - row[3 * i + 2]
print("mean per wave:", {k: int(v / 8) for k, v in tot.items()}, "of", int((s[:, 49] - s[:, 0]).mean()), "(15 chains = 15360 cycles of the matrix pipe per wave)")

# per-workgroup wall clock (100 MHz ticks -> us), relative to the first workgroup's entry
import numpy as np  # noqa: E402
wgs = allb[512:512 + 1024].reshape(256, 4).astype(np.float64) / 100.0
t00 = wgs[:, 0].min()
wgs -= t00
pro = allb[512 + 1024:].reshape(256, 4).astype(np.float64) / 100.0 - t00
tiles = None
print("workgroups (us since the first one entered): entry, weights landed, segment walked, exit")
for name, col in (("entry", 0), ("weights landed", 1), ("walked", 2), ("exit", 3)):
    v = wgs[:, col]
    print(f"  {name:16s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
for name, col in (("root row here, weight DMAs out", 0), ("batch values requested", 1), ("... packed", 2), ("first rows requested", 3)):
    v = pro[:, col]
    print(f"  {name:30s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
walk = wgs[:, 2] - wgs[:, 1]
print(f"  walk (weights landed -> walked): min {walk.min():6.2f} median {np.median(walk):6.2f} max {walk.max():6.2f}")
order = np.argsort(wgs[:, 3])
print("  last five to exit:", [(int(w), round(float(wgs[w, 3]), 2), round(float(walk[w]), 2)) for w in order[-5:]])
print("  first five to exit:", [(int(w), round(float(wgs[w, 3]), 2), round(float(walk[w]), 2)) for w in order[:5]])
from cirkit_amd.fusion import leaf_segments  # noqa: E402
seg = leaf_segments(49, 128, 256)
nt = seg[:, 2] - seg[:, 1]
for n in sorted(set(nt.tolist())):
    m = nt == n
    print(f"  workgroups with {n} tiles: {int(m.sum()):3d}, walk median {np.median(walk[m]):6.2f} us, exit median {np.median(wgs[m, 3]):6.2f} max {wgs[m, 3].max():6.2f}")
