mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_update.py -m gpu -x -q > gpurun_out/pytest_update.log 2>&1 < /dev/null; tail -30 gpurun_out/pytest_update.log
