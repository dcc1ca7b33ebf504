python scripts/bench_plan.py cfg5_sos_c_k32 4096 30 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "5 or sos or signed or complex" 2>&1 | tail -3
