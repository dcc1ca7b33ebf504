cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent or cfg2" 2>&1 | tail -2
for i in 1 2; do
rocprofv3 --kernel-trace --stats -d /tmp/lb$i -o lb -- python scripts/lab_leaf.py 4096 200 default > /tmp/lb.log 2>&1
python scripts/rocprof_summary.py $(find /tmp/lb$i -name "*results.db" | head -1) 2>&1 | grep -E "leaf_persistent" | head -1 | cut -c1-140
done
