mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/pytest_x.log 2>&1 < /dev/null; tail -40 gpurun_out/pytest_x.log
