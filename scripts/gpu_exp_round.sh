mkdir -p gpurun_out; rm -f gpurun_out/exp_pf2.txt
for rep in 1 2; do
for lib in libcfmm_amd.so libcfmm_pf.so; do
for w in config3 product1m config5; do
CFMM_AMD_LIB=$PWD/cfmmrouter.jl_amd/$lib timeout 300 python scripts/exp.py $w "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$lib $w /" | tee -a gpurun_out/exp_pf2.txt
done
done
done
