# scratch A/B session on the GPU box (rewritten per experiment; results go to gpurun_out/, conclusions to DESIGN.md)
mkdir -p gpurun_out
for w in config3 product1m; do
timeout 300 python scripts/exp.py $w "" "alternate=0" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /"
done
