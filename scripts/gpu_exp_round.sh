# scratch A/B session on the GPU box (rewritten per experiment; results go to gpurun_out/, conclusions to DESIGN.md)
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu --maxfail=25 -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 < /dev/null; tail -2 gpurun_out/smoke.log
{
for w in product1m config3 config4shard config5 config2; do
  timeout 300 python scripts/exp.py $w "" "fast_math=0" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /"
done
echo "== prefetch build"
for w in product1m config3 config4shard config5; do
  CFMM_AMD_LIB=$R/cfmmrouter.jl_amd/libcfmm_amd_pf.so timeout 300 python scripts/exp.py $w "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/pf $w /"
done
echo "== geometry"
timeout 300 python scripts/exp.py product1m "max_grid=512" "max_grid=384" "block=512" "block=512,max_grid=1024" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/product1m /"
timeout 300 python scripts/exp.py config4shard "max_grid=512" "block=512" "block=512,max_grid=1024" "bin_copies=1" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/config4shard /"
timeout 300 python scripts/exp.py config3 "max_grid=1024" "block=1024" "block=1024,max_grid=512" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/config3 /"
} > gpurun_out/exp1.txt 2>&1
cat gpurun_out/exp1.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_config3.log 2>&1 < /dev/null; tail -c 3000 gpurun_out/bench_config3.log
