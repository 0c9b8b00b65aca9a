mkdir -p gpurun_out; rm -f gpurun_out/exp_dbg2.txt
for lib in libcfmm_amd.so libcfmm_dbg1.so libcfmm_dbg2.so libcfmm_dbg3.so libcfmm_dbg4.so libcfmm_dbg7.so; do
CFMM_AMD_LIB=$PWD/cfmmrouter.jl_amd/$lib timeout 300 python scripts/exp.py config3 "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$lib /" | tee -a gpurun_out/exp_dbg2.txt
done
