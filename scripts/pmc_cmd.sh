#!/bin/bash
# usage: scripts/pmc_cmd.sh <command ...>  -- SQ occupancy / issue counters per kernel of any command (one rocprofv3 pass per group)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp CIRKIT_BENCH_NO_PMC=1
rm -rf /tmp/pmcsq && mkdir -p /tmp/pmcsq
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcsq/g$i -o p -- "$@" > /dev/null 2>&1)
done
python - <<'PY'
import os, sqlite3
rows = {}
for r, _, fs in os.walk("/tmp/pmcsq"):
    for f in fs:
        if f.endswith("results.db"):
            con = sqlite3.connect(os.path.join(r, f))
            for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
                if k.startswith(("__amd", "at::")): continue
                rows.setdefault(k, {})[c] = (v, n)
for k, d in rows.items():
    print(k, "dispatches", max(n for _, n in d.values()))
    for c in sorted(d): print(f"   {c:32s} {d[c][0]:14.0f}")
PY
