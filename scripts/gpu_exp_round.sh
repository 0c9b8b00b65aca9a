mkdir -p gpurun_out; rm -f gpurun_out/exp_warm.txt
for ws in 0 0.5 3; do
for w in config3 product1m config2; do
WARM_S=$ws timeout 300 python scripts/exp.py $w "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/warm=$ws $w /" | tee -a gpurun_out/exp_warm.txt
done
done
WARM_S=3 K=2000 timeout 300 python scripts/exp.py config3 "" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/warm=3 K=2000 config3 /" | tee -a gpurun_out/exp_warm.txt
(python scripts/exp.py config3 "" > /dev/null 2>&1 &) ; sleep 6; rocm-smi --showclocks 2>/dev/null | grep -i sclk
