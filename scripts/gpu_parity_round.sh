mkdir -p gpurun_out
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$i.log 2>&1 < /dev/null; tail -3 gpurun_out/pytest_gpu_$i.log; done
