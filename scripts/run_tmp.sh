mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -x -q -m gpu > gpurun_out/pt_tr.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/pt_tr.log | tail -5
python scripts/bench_train.py 4096 100 2 fused | tail -1
python scripts/bench_train.py 4096 100 2 fused | tail -1
