#!/bin/bash
# The rocprofv3 recipe behind profiles/<round>_*: run on the GPU box (gpurun) from the repository root.
#
#     bash scripts/profile_round.sh r02 [tag]
#
# Writes into gpurun_out/<round>/ (scratch; copy what is to be judged into profiles/):
#   <round>_<tag>_bench.json      the line printed by `python bench.py` (all defaults)
#   <round>_<tag>_bench.txt       kernel trace of `bench.py --no-cpu-baseline --no-other-configs --no-variants --no-live-pmc`
#                                 + one PMC pass per counter (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES)
#   traffic_<tag>.json            scripts/make_traffic_json.py entry of the FETCH/WRITE passes
# Counters are collected in their own runs, with --kernel-trace only (never with a sys/hip/hsa trace).
set -u
ROUND=${1:-r02}
TAG=${2:-a}
OUT=gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
PY=python
QUIET="--no-cpu-baseline --no-other-configs --no-variants --no-live-pmc"
PMCARGS="--steps 10 --warmup 2 --rounds 1 --no-cpu-baseline --no-kernel-breakdown --no-variants --no-other-configs --no-live-pmc"

$PY bench.py > "$OUT/${ROUND}_${TAG}_bench.json" 2> "$OUT/${ROUND}_${TAG}_bench.err"
tail -1 "$OUT/${ROUND}_${TAG}_bench.json"

rocprofv3 --kernel-trace --stats -d "$OUT/trace_$TAG" -o trace -- $PY bench.py $QUIET > "$OUT/trace_$TAG.log" 2>&1
dbs=()
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_${TAG}_$c" -o pmc -- $PY bench.py $PMCARGS > "$OUT/pmc_${TAG}_$c.log" 2>&1
  db=$(find "$OUT/pmc_${TAG}_$c" -name '*.db' | head -1)
  [ -n "$db" ] && dbs+=("$db")
done
trace_db=$(find "$OUT/trace_$TAG" -name '*.db' | head -1)
{
  echo "# python bench.py $QUIET  (kernel trace);  PMC passes: python bench.py $PMCARGS"
  $PY scripts/rocprof_summary.py "$trace_db" --pmc "${dbs[@]}"
  echo
  echo "# steady-state timeline"
  $PY scripts/timeline.py "$trace_db" 2>/dev/null | tail -40
} > "$OUT/${ROUND}_${TAG}_bench.txt"
f=$(find "$OUT/pmc_${TAG}_FETCH_SIZE" -name '*.db' | head -1)
w=$(find "$OUT/pmc_${TAG}_WRITE_SIZE" -name '*.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && TRAFFIC_OUT="$OUT/traffic_$TAG.json" $PY scripts/make_traffic_json.py "$ROUND,$TAG" "$f" "$w" > /dev/null
# the databases are large: only the summaries travel back
find "$OUT" -name '*.db' -delete
head -30 "$OUT/${ROUND}_${TAG}_bench.txt"
