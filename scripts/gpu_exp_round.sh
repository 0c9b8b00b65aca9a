# scratch session on the GPU box (rewritten per experiment; results go to gpurun_out/, conclusions to DESIGN.md)
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu --maxfail=25 -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -8 gpurun_out/pytest_gpu.log
{
for w in product1m config3 config4shard config5 config2 univ3_ticks; do
  timeout 300 python scripts/exp.py $w "" "fast_math=0" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /"
done
} > gpurun_out/exp2.txt 2>&1
cat gpurun_out/exp2.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_config3.log 2>&1 < /dev/null; python scripts/show_bench.py gpurun_out/bench_config3.log config3
