mkdir -p gpurun_out; rm -f gpurun_out/exp_alt4.txt
for w in config3 product1m config4shard config5 config2; do
timeout 300 python scripts/exp.py $w "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /" | tee -a gpurun_out/exp_alt4.txt
done
timeout 300 python scripts/size_scaling.py 2>&1 | grep -v amdgpu | tee gpurun_out/size_scaling.txt
