mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python scripts/exp.py config3 "" "geomean_exact=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_geo.txt
