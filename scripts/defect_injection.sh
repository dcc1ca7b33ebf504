#!/bin/bash
# prove that the gradient tests of the job form see a wrong <W, dW> in a sum job's epilogue.  Builds a lab copy of
# the library whose epilogue scales <W, dW> by (1 + DEFECT) (default 1e-3) and runs the job-form gradient tests against it:
# they must FAIL.  Usage (GPU box):  bash scripts/defect_injection.sh [defect]   -> exit 0 when the defect was caught.
set -u
cd "$(dirname "$0")/.."
DEFECT=${1:-1e-3f}
bash scripts/lab_build.sh defect cirkit_amd/csrc/ck_jobs.hip -DCK_JOBS_LAB_DEFECT=$DEFECT > /dev/null || exit 2
CIRKIT_HIP_LIB=$PWD/build/lab/lib_defect.so python -m pytest tests/test_training_jobs.py -q -m gpu -x \
  -k "shallow_circuit or gradients_match_the_layerwise" > build/lab/defect.log 2>&1
rc=$?
grep -E "^E  .*(assert|Error)" build/lab/defect.log | head -6
tail -3 build/lab/defect.log
if [ $rc -ne 0 ]; then echo "defect $DEFECT in <W, dW>: CAUGHT (pytest rc $rc)"; exit 0; fi
echo "defect $DEFECT in <W, dW>: NOT caught"; exit 1
