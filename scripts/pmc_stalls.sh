cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
ARGS="--steps 10 --warmup 2 --rounds 1 --no-cpu-baseline --no-kernel-breakdown --no-variants --no-other-configs --no-live-pmc"
dbs=""
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pl$i -o pmc -- python bench.py $ARGS > /tmp/pl$i.log 2>&1
  db=$(find /tmp/pl$i -name '*.db' | head -1)
  [ -n "$db" ] && dbs="$dbs $db" || { echo "group '$grp' failed"; tail -3 /tmp/pl$i.log; }
done
python scripts/pmc_table.py $dbs > gpurun_out/r02/pmc_leaf_stalls.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/pmc_leaf_stalls.json'))
for k,v in d.items():
    if 'leaf_persistent' in k or 'tail16' in k or 'softmax_batch' in k:
        print(k)
        for c,x in sorted(v.items()): print(f"   {c:36s} {x:16.1f}")
PY
