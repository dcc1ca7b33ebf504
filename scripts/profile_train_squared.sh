#!/bin/bash
# Kernel trace of the squared-circuit training step (config 5's plans): per-kernel totals -> gpurun_out/train_sq/stats.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/train_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in ${BATCHES:-256 4096}; do
  python $ROOT/scripts/bench_train_squared.py $B > $OUT/plain_$B.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$B -o sq -- python $ROOT/scripts/bench_train_squared.py $B > $OUT/run_$B.log 2>&1
  db=$(find $OUT/prof_$B -name '*.db' | head -1)
  { echo "# python scripts/bench_train_squared.py $B   (3 + 10 steps; the first two run eagerly and size the scratch)"; grep "training step" $OUT/plain_$B.log; echo;
    python $ROOT/scripts/rocprof_summary.py "$db"; } > $OUT/stats_$B.txt
  python $ROOT/scripts/rocprof_timeline.py "$db" opt_range_kernel > $OUT/timeline_$B.txt 2>&1
  find $OUT -name '*.db' -delete
  head -50 $OUT/stats_$B.txt
done
