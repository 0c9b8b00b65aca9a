mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or optimality or interior" > gpurun_out/pytest_parity.log 2>&1 < /dev/null; tail -15 gpurun_out/pytest_parity.log
