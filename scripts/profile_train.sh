#!/bin/bash
# rocprofv3 evidence for the fused training step of config 2: run on the GPU box from the repository root.
#
#     bash scripts/profile_train.sh r04 [tag] [fused|layerwise]
#
# Writes gpurun_out/<round>/<round>_<tag>_train.txt: the line scripts/bench_train.py prints, the kernel trace of
# `scripts/bench_train.py 4096 40 2 <mode>` (BASELINE config 2, 4096 rows, forward + backward + Adam per step) and one PMC
# pass per counter (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) -- counters in their own runs with --kernel-trace only.
set -u
ROUND=${1:-r04}
TAG=${2:-a}
MODE=${3:-fused}
OUT=gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python scripts/bench_train.py 4096 40 2 $MODE"
$CMD > "$OUT/train_$TAG.line" 2> "$OUT/train_$TAG.err"
rocprofv3 --kernel-trace --stats -d "$OUT/ttrace_$TAG" -o trace -- $CMD > "$OUT/ttrace_$TAG.log" 2>&1
dbs=()
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/tpmc_${TAG}_$c" -o pmc -- $CMD > "$OUT/tpmc_${TAG}_$c.log" 2>&1
  db=$(find "$OUT/tpmc_${TAG}_$c" -name '*.db' | head -1)
  [ -n "$db" ] && dbs+=("$db")
done
trace_db=$(find "$OUT/ttrace_$TAG" -name '*.db' | head -1)
{
  echo "# $CMD"
  tail -1 "$OUT/train_$TAG.line"
  echo
  python scripts/rocprof_summary.py "$trace_db" --pmc "${dbs[@]}"
} > "$OUT/${ROUND}_${TAG}_train.txt"
find "$OUT" -name '*.db' -delete
head -40 "$OUT/${ROUND}_${TAG}_train.txt"
