mkdir -p gpurun_out; rm -f gpurun_out/exp_pack.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in config5; do
timeout 300 python scripts/exp.py $w "pack=0" "pack=1" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /" | tee -a gpurun_out/exp_pack.txt
for pk in 0 1; do timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu --cold-only --workload $w --opt pack=$pk 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); r=l['roofline']; print('$w cold pack=$pk step %.2f sweep %.2f frac %.3f'%(1e3*l['ms_per_step'],1e3*r['kernel_ms'],r['frac']))" | tee -a gpurun_out/exp_pack.txt; done
done
