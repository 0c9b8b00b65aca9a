mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fold.py -m gpu -x -q -k "single_process or determinism" > gpurun_out/pytest_x.log 2>&1 < /dev/null; tail -30 gpurun_out/pytest_x.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-3500
