#!/usr/bin/env python3
"""Instruction mix of every basic block of one kernel in the assembly hipcc writes:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Icirkit_amd/csrc \
          -S --cuda-device-only -o leaf.s cirkit_amd/csrc/ck_leaf.hip
    python scripts/asm_blocks.py leaf.s _ZN12_GLOBAL__N_122leaf_persistent_kernelILi4ELi8ELb0ELb1ELb1ELb0EEEvNS_8LeafArgsE [--all]

Per block (by default those with MFMAs or at least 40 instructions): label, first line, instruction count, counts by class
(mfma, valu, salu, lds, vmem = vector memory, smem, lane = v_readlane / v_writelane i.e. spilled scalar registers, scr =
scratch, wait, br, bar) and branch targets.  What the hot loop of a kernel is made of, before and after a change.
"""
import re
import sys


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_readlane", "v_writelane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_barrier"):
        return "bar"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scr"
    return "other"


def main() -> None:
    src, kernel = sys.argv[1], sys.argv[2]
    everything = "--all" in sys.argv[3:]
    text = open(src, encoding="utf-8").read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(kernel + ":"))
    lines = []
    for l in text[start:]:
        lines.append(l)
        if "s_endpgm" in l:
            break
    blocks, cur = [], ["entry", 0, {}, []]
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), i, {}, []]
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        c = classify(t.split()[0])
        cur[2][c] = cur[2].get(c, 0) + 1
        if c == "br":
            cur[3].append(t.split()[-1])
    blocks.append(cur)
    print(f"{len(lines)} lines")
    for name, line, counts, targets in blocks:
        total = sum(counts.values())
        if everything or total >= 40 or counts.get("mfma"):
            print(name, line, total, counts, targets)


if __name__ == "__main__":
    main()
