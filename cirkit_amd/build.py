"""Build the HIP extension in-tree: cirkit_amd/lib/libcirkit_hip.so (gfx950 only).

    python -m cirkit_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the working tree.
"""

from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libcirkit_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: gfx950 has a unified VGPR/AGPR file; keeping MFMA results in VGPRs removes
# the v_accvgpr_read/write pair around every accumulator (32 VALU instructions per 32x32x32 step,
# which matter because fp32-input MFMA shares the fp32 ALUs with the VALU -- DESIGN.md section 4.2).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(SRC, "*.h"))) + [
        os.path.join(os.path.dirname(HERE), "include", "cirkit_hip.h"),
        os.path.join(os.path.dirname(HERE), "include", "cirkit_hip_internal.h"),
    ]
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + hdrs):
            continue
        cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or not _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
