#!/bin/bash
# FETCH_SIZE / WRITE-free calibration per access kind (scripts/ubench/fetch_calib.hip): prints bytes requested / (FETCH_SIZE KiB x 1024)
# and the achieved GB/s of every kernel.  Run on the GPU box from the repository root; writes gpurun_out/<round>_fetch_calib.txt.
set -u
ROUND=${1:-r05}
export TMPDIR=/tmp
OUT=gpurun_out/fetch_calib
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc" -o p -- scripts/ubench/fetch_calib.bin > "$OUT/pmc.log" 2>&1
rocprofv3 --kernel-trace -d "$OUT/trace" -o p -- scripts/ubench/fetch_calib.bin > "$OUT/trace.log" 2>&1
python - "$OUT" <<'PY' | tee "gpurun_out/${ROUND}_fetch_calib.txt"
import glob, sqlite3, sys
out = sys.argv[1]
pmc = sqlite3.connect(glob.glob(out + "/pmc/**/*.db", recursive=True)[0])
tr = sqlite3.connect(glob.glob(out + "/trace/**/*.db", recursive=True)[0])
dur = {n: d for n, d in tr.execute("select name, min(duration) from kernels group by name")}
print("# every kernel reads each byte of a 2 GiB buffer exactly once (scripts/ubench/fetch_calib.hip); factor = bytes requested / FETCH_SIZE bytes")
print(f"{'kernel':44s} {'FETCH_SIZE MiB':>15s} {'factor':>8s} {'GB/s':>8s}")
for n, v in pmc.execute("select kernel_name, avg(value) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name order by kernel_name"):
    if n.startswith("__amd"):
        continue
    short = n.split("(")[0].replace("void ", "")
    d = next((x for k, x in dur.items() if k.split("(")[0].replace("void ", "") == short), None)
    print(f"{short:44s} {v / 1024:15.1f} {2048.0 * 1024 / v:8.3f} {(2**31 / (d * 1e-9) / 1e9) if d else float('nan'):8.0f}")
PY
find "$OUT" -name '*.db' -delete
