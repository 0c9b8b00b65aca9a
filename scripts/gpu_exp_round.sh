mkdir -p gpurun_out; rm -f gpurun_out/exp_nt.txt
run() { lib=$1; w=$2; mode=$3; 
CFMM_AMD_LIB=$PWD/cfmmrouter.jl_amd/$lib timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu $mode --workload $w 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); r=l['roofline']; print('$w $lib $mode step %.2f sweep %.2f frac %.3f layout %.3f'%(1e3*l['ms_per_step'],1e3*r['kernel_ms'],r['frac'],r['layout']['frac']))" | tee -a gpurun_out/exp_nt.txt; }
for w in config3 product1m; do
for lib in libcfmm_amd.so libcfmm_amd_nt.so libcfmm_amd.so libcfmm_amd_nt.so; do
run $lib $w --cold-only
done
for lib in libcfmm_amd.so libcfmm_amd_nt.so; do
run $lib $w --no-cold
done
done
