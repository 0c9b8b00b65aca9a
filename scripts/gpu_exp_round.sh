# scratch session on the GPU box (rewritten per experiment; results go to gpurun_out/, conclusions to DESIGN.md)
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu --maxfail=25 -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -8 gpurun_out/pytest_gpu.log
{
for w in univ3_ticks config5 config3 product1m; do
  timeout 300 python scripts/exp.py $w "" "fast_math=0" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /"
done
} > gpurun_out/exp3.txt 2>&1
cat gpurun_out/exp3.txt
timeout 400 python bench.py --steps 100 --warmup 10 --workload univ3_ticks > gpurun_out/bench_univ3_ticks.log 2>&1 < /dev/null; python scripts/show_bench.py gpurun_out/bench_univ3_ticks.log univ3_ticks
