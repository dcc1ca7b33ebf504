#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl_k64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "256 40 6" "1024 20 4"; do
  set -- $spec
  rocprofv3 --kernel-trace -d $OUT/p_$3 -o t -- python $ROOT/scripts/bench_train.py $1 $2 $3 auto > $OUT/run_$3.log 2>&1
  db=$(find $OUT/p_$3 -name '*.db' | head -1)
  python $ROOT/scripts/rocprof_timeline.py "$db" opt_tick_kernel > $OUT/timeline_cfg$3.txt 2>&1
  find $OUT -name '*.db' -delete
done
