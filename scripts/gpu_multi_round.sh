mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/pytest_multi.log 2>&1 < /dev/null; tail -25 gpurun_out/pytest_multi.log
