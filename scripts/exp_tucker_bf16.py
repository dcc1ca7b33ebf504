#!/usr/bin/env python3
"""What the bf16-split stream-K Tucker launch spends its time on: lab builds of ck_gemm.hip with parts compiled out
(-DCK_TUCKER_LABBITS=<bits>, wrong results on purpose; the bits are listed beside `kLab` in ck_gemm.hip) and the time of the
largest layer of the notebook configuration (784 folds, K = 64, batch 128) under each.  The product library is not touched.

    [BATCH=128] python scripts/exp_tucker_bf16.py [bits ...]      (BATCH=1024: every weight an L2 hit, HBM out of the picture)"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import build as B  # noqa: E402

bits = [int(b) for b in sys.argv[1:]] or [0, 1, 2, 4, 6, 8, 16, 32, 48, 64, 14, 62, 126]
tmp = tempfile.mkdtemp(prefix="cktucker")
B.build(verbose=False)
objs = [os.path.join(B.LIB_DIR, o) for o in sorted(os.listdir(B.LIB_DIR)) if o.endswith(".o") and o != "ck_gemm.o"]


def make(b):
    obj, lib = os.path.join(tmp, f"ck_gemm_{b}.o"), os.path.join(tmp, f"libcirkit_hip_lab{b}.so")
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-w", f"-DCK_TUCKER_LABBITS={b}", "-c", os.path.join(B.SRC, "ck_gemm.hip"), "-o", obj])
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs])
    return lib


with ThreadPoolExecutor(max_workers=8) as ex:
    libs = list(ex.map(make, bits))
for b, lib in zip(bits, libs):
    env = dict(os.environ, CK_LIB=lib, ONLY="bf16x3,bf16x6", KERNELS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_notebook.py"), os.environ.get("BATCH", "128"), "10"], env=env, capture_output=True, text=True).stdout
    t = [l.split()[-2] for l in out.splitlines() if l.strip().startswith("layer   1 ")]
    tot = [l.split()[1] for l in out.splitlines() if l.startswith("contraction=")]
    print(f"bits {b:3d}: largest layer bf16x3 / bf16x6 = {' / '.join(t)} ms, forward {' / '.join(tot)} ms", flush=True)
