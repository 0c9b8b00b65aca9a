mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_examples_gpu.py -m gpu -x -q > gpurun_out/pytest_x.log 2>&1 < /dev/null; tail -25 gpurun_out/pytest_x.log
