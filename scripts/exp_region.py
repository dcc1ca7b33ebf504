#!/usr/bin/env python3
"""Experiments inside the exact K = 64 region launch (VERDICT r3 #4 / LAB_NOTES 9.7 "left to try"): lab builds of ck_cp.hip
(-DCK_REGION_LABBITS=<bits>: 1 = two independent 16-deep MFMA chains per weight unit instead of one 32-deep chain, 2 = two
workgroups per CU with 256 registers instead of three with 168, 4 = a ring of two whole matrices and one barrier per slot
instead of two (needs 2), 8 = the operand reads of unit (t, 1) issued under the chain of unit (t, 0) (needs 4); bit 1 changes
the order of the adds, so results differ in the last bits) and config 4's forward
time and largest region launch under each.  The product library is not touched.

    python scripts/exp_region.py [bits ...]"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import build as B  # noqa: E402

bits = [int(b) for b in sys.argv[1:]] or [0, 1, 2, 3, 6, 14]
tmp = tempfile.mkdtemp(prefix="ckregion")
B.build(verbose=False)
objs = [os.path.join(B.LIB_DIR, o) for o in sorted(os.listdir(B.LIB_DIR)) if o.endswith(".o") and o != "ck_cp.o"]


def make(b):
    obj, lib = os.path.join(tmp, f"ck_cp_{b}.o"), os.path.join(tmp, f"libcirkit_hip_lab{b}.so")
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-w", f"-DCK_REGION_LABBITS={b}", "-c", os.path.join(B.SRC, "ck_cp.hip"), "-o", obj])
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
    return lib


RUN = r'''
import os, sys, torch
sys.path.insert(0, %r)
from cirkit_amd import _capi
_capi._LIB_PATH = os.environ["CK_LIB"]
from cirkit_amd.circuit import HipCircuit
from cirkit_amd.initializers import init_plan_tensors
from cirkit_amd.templates import image_data
dev = torch.device("cuda:0")
plan4 = image_data((1, 28, 28), "poon-domingos", input_layer="gaussian", num_input_units=64, sum_product_layer="cp", num_sum_units=64)
hc = HipCircuit(plan4, init_plan_tensors(plan4), device=dev)
x = torch.randn((4096, 784), generator=torch.Generator().manual_seed(4)).to(dev)
for _ in range(5): y = hc(x)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): y = hc(x)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
top = sorted(hc.profile_kernels(x, 5), key=lambda r: -r["ms"])[0]
print(f"{sorted(ts)[2]:.3f} ms / forward, mean LL {float(y.mean()):.4f}, largest launch {top['ms']:.4f} ms")
''' % ROOT

with ThreadPoolExecutor(max_workers=8) as ex:
    libs = list(ex.map(make, bits))
for b, lib in zip(bits, libs):
    out = subprocess.run([sys.executable, "-c", RUN], env=dict(os.environ, CK_LIB=lib), capture_output=True, text=True)
    print(f"bits {b}: {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
