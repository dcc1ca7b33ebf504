#!/bin/bash
# rocprofv3 evidence for BASELINE config 5 evaluated as complex-valued parameters require (HipCircuit(signed_real=False): the
# linear (re, im) tile path of csrc/ck_clin.hip): a kernel trace and SQ instruction counters (VALU per MFMA) of
# `ONLY="depth 3" python scripts/bench_cfg5_complex.py 4096 50`.   bash scripts/profile_cfg5_complex.sh <round> <tag>
#   -> profiles/<round>_<tag>_cfg5_complex.txt   (PMC passes carry --kernel-trace only, one counter group per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r06}; TAG=${2:-a}
OUT=$R/gpurun_out/${ROUND}_${TAG}_cfg5_complex.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5c && mkdir -p /tmp/c5c
{
  echo "# ONLY='depth 3' python scripts/bench_cfg5_complex.py 4096 50   (config 5, real parameters forced onto the complex path)"
  (cd $R && ONLY="depth 3" python scripts/bench_cfg5_complex.py 4096 50 2>&1 | grep -v amdgpu.ids)
  echo "# COMPLEX_W=1 ONLY='depth 3' ...   (complex-valued Embedding and sum weights: four MFMA chains per contraction)"
  (cd $R && COMPLEX_W=1 ONLY="depth 3" python scripts/bench_cfg5_complex.py 4096 50 2>&1 | grep -v amdgpu.ids)
  echo "# ONLY='layer-wise' ...   (the layer-wise complex kernels this path replaces)"
  (cd $R && ONLY="layer-wise" python scripts/bench_cfg5_complex.py 4096 50 2>&1 | grep -v amdgpu.ids)
  (cd $R && ONLY="depth 3" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/c5c/trace -o t -- python scripts/bench_cfg5_complex.py 4096 50 > /dev/null 2>&1)
  python $R/scripts/rocprof_summary.py $(find /tmp/c5c/trace -name "*results.db" | head -1)
  echo "# SQ counters per dispatch (scripts/pmc_cmd.sh: one rocprofv3 --kernel-trace --pmc pass per group)"
  (cd $R && ONLY="depth 3" bash scripts/pmc_cmd.sh python scripts/bench_cfg5_complex.py 4096 20 2>&1 | grep -A17 "^clin_leaf\|^clin_layer\|^clin_tail\|^clin_table")
} > $OUT 2>&1
echo $OUT
