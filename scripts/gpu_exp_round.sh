mkdir -p gpurun_out
for w in config3 config4shard config5 config2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --no-cold --workload $w > gpurun_out/bench_${w}.log 2>&1 < /dev/null
  tail -1 gpurun_out/bench_${w}.log | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('$w', 'step us %.2f'%(1e3*l['ms_per_step']), 'route', l.get('route'))
"
done
timeout 600 python -m pytest tests -m gpu -x -q -k "route or lbfgs or golden" 2>&1 | tail -3
