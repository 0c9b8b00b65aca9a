#!/bin/bash
# round-4 session 1: GPU tests of the FASTK build + geometry A/B (occupancy now possible) + one bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -15 gpurun_out/pytest_gpu.log
{
for w in product1m config4shard config5; do
  timeout 200 python scripts/exp.py $w "" "max_grid=512" "max_grid=384" "fast_math=0" "block=512,max_grid=512" "block=512,max_grid=1024"
  COLD=1 timeout 300 python scripts/exp.py $w "" "max_grid=512" "fast_math=0" "block=512,max_grid=1024"
done
timeout 200 python scripts/exp.py config3 "" "max_grid=768" "max_grid=1024" "fast_math=0" "block=1024,max_grid=512"
COLD=1 timeout 300 python scripts/exp.py config3 "" "max_grid=768" "max_grid=1024" "fast_math=0"
timeout 200 python scripts/exp.py univ3_ticks "" "max_grid=512" "fast_math=0"
COLD=1 timeout 300 python scripts/exp.py univ3_ticks "" "max_grid=512"
timeout 100 python scripts/exp.py config2 "" "fast_math=0"
} > gpurun_out/ab_session1.txt 2>&1
cat gpurun_out/ab_session1.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_s1.log 2>&1 < /dev/null; python scripts/show_bench.py gpurun_out/bench_s1.log config3
