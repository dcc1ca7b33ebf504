#!/bin/bash
# rocprofv3 kernel traces of the K = 64 training steps: run on the GPU box from the repository root.
#     bash scripts/profile_train_k64.sh r05 <tag> [mode]
# config 6 = the circuit of the reference's learning-a-circuit notebook (QuadGraph, CP, K = 64, batch 256); config 4 at 1024 rows.
set -u
ROUND=${1:-r05}; TAG=${2:-a}; MODE=${3:-auto}
OUT=gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
for spec in "256 40 6" "1024 20 4"; do
  set -- $spec
  name=cfg$3_b$1
  CMD="python scripts/bench_train.py $1 $2 $3 $MODE"
  $CMD > "$OUT/train_${TAG}_$name.line" 2> "$OUT/train_${TAG}_$name.err"
  rocprofv3 --kernel-trace --stats -d "$OUT/ttrace_${TAG}_$name" -o trace -- $CMD > "$OUT/ttrace_${TAG}_$name.log" 2>&1
  db=$(find "$OUT/ttrace_${TAG}_$name" -name '*.db' | head -1)
  { echo "# $CMD"; tail -1 "$OUT/train_${TAG}_$name.line"; echo; python scripts/rocprof_summary.py "$db"; } > "$OUT/${ROUND}_${TAG}_train_$name.txt"
  find "$OUT" -name '*.db' -delete
  head -45 "$OUT/${ROUND}_${TAG}_train_$name.txt"
done
# counter passes (one counter per pass, kernel trace only) of the notebook configuration: HBM bytes and MFMA-busy cycles per launch
if [ "${PMC:-0}" = "1" ]; then
  CMD="python scripts/bench_train.py 256 40 6 $MODE"
  dbs=()
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --kernel-trace --pmc $c -d "$OUT/tpmc_${TAG}_$c" -o pmc -- $CMD > "$OUT/tpmc_${TAG}_$c.log" 2>&1
    db=$(find "$OUT/tpmc_${TAG}_$c" -name '*.db' | head -1)
    [ -n "$db" ] && dbs+=("$db")
  done
  python scripts/rocprof_summary.py "${dbs[0]}" --pmc "${dbs[@]}" | sed -n '/# PMC/,$p' >> "$OUT/${ROUND}_${TAG}_train_cfg6_b256.txt"
  find "$OUT" -name '*.db' -delete
fi
