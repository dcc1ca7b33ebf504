"""Registers, spills and LDS of every kernel of one translation unit: python scripts/kernel_regs.py ck_leaf [name filter]"""
import re, subprocess, sys
src = f"/root/repo/cirkit_amd/csrc/{sys.argv[1]}.hip"
out = f"/tmp/{sys.argv[1]}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form",
                "-I/root/repo/include", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for b in s.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if len(sys.argv) > 2 and sys.argv[2] not in dn:
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", b).group(1)
    print(f"{dn[:100]:100s} vgpr {g('vgpr_count')} spill {g('vgpr_spill_count')} sspill {g('sgpr_spill_count')} lds {g('group_segment_fixed_size')}")
