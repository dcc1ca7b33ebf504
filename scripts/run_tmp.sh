mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_training_squared.py -x -q -m gpu > gpurun_out/pt_sq.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/pt_sq.log | tail -30
BATCHES=4096 bash scripts/profile_train_squared.sh > gpurun_out/prof_sq.log 2>&1; head -45 gpurun_out/train_sq/stats_4096.txt
