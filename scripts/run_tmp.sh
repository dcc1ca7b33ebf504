mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training_squared.py tests/test_gpu_param_einsum.py tests/test_gpu_squared_kernels.py -x -q -m gpu > gpurun_out/pt_sq.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/pt_sq.log | tail -5
CK_SQ_STEPS=50 timeout 300 python scripts/bench_train_squared.py 256 4096 2>&1 | grep "training step"
CK_SQ_STEPS=50 timeout 300 python scripts/bench_train_squared.py 256 4096 2>&1 | grep "training step"
BATCHES=4096 bash scripts/profile_train_squared.sh > gpurun_out/prof_sq.log 2>&1; grep "bmm\|copyBuffer\|at::native" gpurun_out/train_sq/stats_4096.txt | head
