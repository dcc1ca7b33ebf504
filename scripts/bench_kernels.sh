#!/bin/bash
# prints the per-kernel table of a short bench run (instrumented pass) -- for A/B experiments on the GPU box
python bench.py --no-variants --no-other-configs --no-cpu-baseline --no-live-pmc --rounds 4 --steps 50 "$@" > gpurun_out/bk.json 2> gpurun_out/bk.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bk.json"))
print("ms_per_step", round(d["ms_per_step"], 5), {k[:40]: round(1e3 * v["ms_per_step"], 2) for k, v in d["roofline"].get("kernels", {}).items()})
PY
