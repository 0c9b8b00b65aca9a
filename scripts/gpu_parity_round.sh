mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu.log | head -40
