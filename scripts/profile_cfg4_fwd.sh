#!/bin/bash
# rocprofv3 evidence for BASELINE config 4's forward (Poon-Domingos 28x28, Gaussian leaves, CP, K = 64, 4096 rows): a kernel trace
# and one counter pass each for FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES of
# `python scripts/bench_plan.py cfg4_pd784 4096 20`.   bash scripts/profile_cfg4_fwd.sh <round> <tag>
#   -> gpurun_out/<round>_<tag>_cfg4_fwd.txt   (PMC passes carry --kernel-trace only, one counter per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r06}; TAG=${2:-a}
OUT=$R/gpurun_out/${ROUND}_${TAG}_cfg4_fwd.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4f && mkdir -p /tmp/c4f
CMD="python $R/scripts/bench_plan.py cfg4_pd784 4096 20"
{
  echo "# $CMD   (first line: the script's JSON summary -- ms per forward, launches, event-timed per-kernel table)"
  (cd $R && $CMD 2>&1 | grep -v amdgpu.ids)
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/c4f/trace -o t -- $CMD > /dev/null 2>&1)
  dbs=()
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do
    (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/c4f/pmc_$c -o p -- $CMD > /dev/null 2>&1)
    db=$(find /tmp/c4f/pmc_$c -name '*results.db' | head -1)
    [ -n "$db" ] && dbs+=("$db")
  done
  python $R/scripts/rocprof_summary.py $(find /tmp/c4f/trace -name "*results.db" | head -1) --pmc "${dbs[@]}"
} > $OUT 2>&1
echo $OUT
