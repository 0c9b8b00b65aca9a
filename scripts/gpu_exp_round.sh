mkdir -p gpurun_out; rm -f gpurun_out/exp_blk.txt
timeout 300 python scripts/exp.py config3 "" "block=1024" "" "block=1024" "block=1024,cost_geomean=14" "block=1024,cost_geomean=18" "block=1024,max_grid=512" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/config3 /" | tee -a gpurun_out/exp_blk.txt
for o in "" "--opt block=1024" "" "--opt block=1024"; do timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu --cold-only --workload config3 $o 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); r=l['roofline']; print('config3 cold [$o] step %.2f sweep %.2f frac %.3f'%(1e3*l['ms_per_step'],1e3*r['kernel_ms'],r['frac']))" | tee -a gpurun_out/exp_blk.txt; done
