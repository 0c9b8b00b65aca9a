mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_fold.py -m gpu -x -q > gpurun_out/pytest_fold.log 2>&1 < /dev/null; tail -5 gpurun_out/pytest_fold.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -5 gpurun_out/pytest_gpu.log
for w in config3 config2 product1m config5 config4shard; do
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu --workload $w > gpurun_out/bench_${w}.log 2>&1 < /dev/null
  tail -1 gpurun_out/bench_${w}.log | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); r=l['roofline']
print('$w', 'ms_per_step',l['ms_per_step'],'kernel_ms',r['kernel_ms'],'reduce',r['reduce_kernel_ms'],'cold',r['cold'] and r['cold']['kernel_ms'], 'host',l.get('host_boundary'), 'route', l.get('route'))
"
done
