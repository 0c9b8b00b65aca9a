mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "peer_gather" > gpurun_out/pytest_x.log 2>&1 < /dev/null; tail -30 gpurun_out/pytest_x.log
