#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprof kernel stats.  Every step is bounded.
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 < /dev/null; tail -4 gpurun_out/pytest_gpu.log
for w in ${WORKLOADS:-config3 product1m config2 config5}; do
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu --workload $w > gpurun_out/bench_$w.log 2>&1 < /dev/null
  timeout 20 python scripts/show_bench.py gpurun_out/bench_$w.log $w < /dev/null
done
if [ -n "$TUNE" ]; then
  for t in $TUNE; do timeout 250 python scripts/tune.py $t > gpurun_out/tune_$t.txt 2>&1 < /dev/null; done
fi
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  for w in $PROF; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-cold --workload $w > $GRAFT_REPO_ROOT/gpurun_out/prof_$w.log 2>&1 < /dev/null
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && head -8 "$f"
  done
fi
if [ -n "$PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  for w in $PMC; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${w}_$c -o $w -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --workload $w > $GRAFT_REPO_ROOT/gpurun_out/pmc_${w}_$c.log 2>&1 < /dev/null
    done
  done
  cd $GRAFT_REPO_ROOT
  timeout 60 python scripts/pmc_summary.py gpurun_out $PMC < /dev/null
fi
if [ -n "$FULL" ]; then
  cd $GRAFT_REPO_ROOT
  for w in $FULL; do
    timeout 500 python bench.py --workload $w > gpurun_out/benchfull_$w.log 2>&1 < /dev/null
    tail -1 gpurun_out/benchfull_$w.log | cut -c1-3000
  done
fi
