#!/bin/bash
# Usage (on the GPU box, from the repo root): scripts/gpu_quick.sh "<pytest -k expression or empty>" [bench args...]
# Runs the selected -m gpu tests (log in gpurun_out/pytest.log), then a short bench and prints its per-kernel table.
mkdir -p gpurun_out
K="$1"; shift
if [ -n "$K" ]; then
  python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
  grep -v "^  File\|^Extension modules" gpurun_out/pytest.log | head -60
  tail -5 gpurun_out/pytest.log
fi
python bench.py --no-variants --no-other-configs --no-cpu-baseline "$@" > gpurun_out/bq.json 2> gpurun_out/bq.err
tail -c 400 gpurun_out/bq.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bq.json"))
print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 5), "launches", d["config"]["launches_per_step"],
      "raw", d["config"].get("leaf_reads_raw_batch"))
print({k: round(1e3 * v["ms_per_step"], 2) for k, v in d["roofline"].get("kernels", {}).items()})
print("by round", [round(x, 4) for x in d["timing"]["ms_per_step_by_round"]])
PY
