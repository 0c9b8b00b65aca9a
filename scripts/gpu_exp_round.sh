mkdir -p gpurun_out; rm -f gpurun_out/exp_ct.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fold.py -m gpu -x -q 2>&1 | tail -3
for w in config3 product1m config4shard config5; do
for ct in 0 1 1; do timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu --cold-only --workload $w --opt compact_trades=$ct 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); r=l['roofline']; print('$w cold compact=$ct step %.2f sweep %.2f frac %.3f'%(1e3*l['ms_per_step'],1e3*r['kernel_ms'],r['frac']))" | tee -a gpurun_out/exp_ct.txt; done
done
