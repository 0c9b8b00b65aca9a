mkdir -p gpurun_out; rm -f gpurun_out/exp_tail.txt
timeout 500 python -m pytest tests/test_gpu_fold.py tests/test_gpu_armed.py -m gpu -x -q 2>&1 | tail -2
for w in config3 product1m; do
timeout 300 python scripts/exp.py $w "inline_fold=0" "inline_fold=3" "inline_fold=0" "inline_fold=3" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /" | tee -a gpurun_out/exp_tail.txt
done
