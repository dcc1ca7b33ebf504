#!/bin/bash
# Timing experiments on the fused backward launches (wrong results on purpose): CK_BWD_EXP bits -- 1 no dW contraction,
# 2 no W^T contraction, 4 every tile reads rows 0..31 (cache hits), 8 no gradient-tile stores; CK_BWD_WAVES 4 / 8.
# Needs the lab build: python scripts/bwd_stamps.py builds it and prints its path; export CK_LIB=<that path> first.
# Prints the two launch times (bottom = leaf_bwd_kernel<true, .>, top = <false, .>).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in 4 8; do for e in ${EXPS:-0 3 4 8 15}; do
  CK_BWD_WAVES=$w CK_BWD_EXP=$e rocprofv3 --kernel-trace --stats -d gpurun_out/r04/exp_$e -o trace -- python scripts/bench_train.py 4096 10 2 fused > /dev/null 2>&1
  db=$(find gpurun_out/r04/exp_$e -name "*.db" | head -1)
  echo "waves $w exp $e: $(python scripts/rocprof_summary.py $db | grep '^leaf_bwd_kernel' | head -2 | awk '{print $1, $2, $5}' | tr '\n' ' ')"
  rm -rf gpurun_out/r04/exp_$e
done; done
