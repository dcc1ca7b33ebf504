mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pt_all.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" gpurun_out/pt_all.log | tail -8
