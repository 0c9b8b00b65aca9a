#!/bin/bash
# One GPU-box session for the judged artefacts: parity tests, bench lines, rocprofv3 kernel stats, PMC traffic, SQ counters,
# option A/Bs, route! kernel gaps, the reference's benchmark grid.  Every step is bounded.  Outputs go to gpurun_out/
# (scratch); scripts/collect_profiles.py copies the summaries into profiles/ (tracked).
#   usage: [TESTS=1] [FULL="config3 ..."] [PROF="config3 ..."] [PMC="config3"] [PMCX="config3"] [GAPS="config3 config5"]
#          [AB="workload:spec;spec ..."] [GRID=1] bash scripts/gpu_round.sh
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
if [ -n "$TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 < /dev/null | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|amdgpu.ids" > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
  g++ -O2 -std=c++17 -I include -o /tmp/solver_bench.bin scripts/native/solver_bench.cpp -L cfmmrouter.jl_amd -lcfmm_amd -Wl,-rpath,$R/cfmmrouter.jl_amd -Wl,-rpath,/opt/rocm/lib && { for n in 64 256 512; do /tmp/solver_bench.bin $n 20000 40; done; } > gpurun_out/solver_bench.txt 2>&1; cat gpurun_out/solver_bench.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 < /dev/null; tail -1 gpurun_out/smoke.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "parity_by_convergence" 2>&1 < /dev/null | grep -E "default|passed|failed" > gpurun_out/route_convergence.txt; cat gpurun_out/route_convergence.txt
  timeout 900 python -m pytest tests/test_route_fortran_pin.py -m gpu -q -s 2>&1 < /dev/null | grep -E "Fortran|passed|failed" > gpurun_out/route_fortran_pin.txt; cat gpurun_out/route_fortran_pin.txt
fi
for w in $FULL; do
  steps=100; [ "$w" = "config3" ] && steps=20      # config3 exactly as the driver runs it (--steps 20 --warmup 5)
  warm=10; [ "$w" = "config3" ] && warm=5
  timeout 500 python bench.py --steps $steps --warmup $warm --workload $w > gpurun_out/benchfull_$w.log 2>&1 < /dev/null
  timeout 20 python scripts/show_bench.py gpurun_out/benchfull_$w.log $w < /dev/null
done
if [ -n "$GRID" ]; then
  timeout 900 python bench.py --workload scaling > gpurun_out/benchfull_scaling.log 2>&1 < /dev/null; tail -c 600 gpurun_out/benchfull_scaling.log
fi
if [ -n "$AB" ]; then   # AB="product1m:;max_grid=512 config3:;max_grid=768"  (specs separated by ';', '' = defaults), warm then HBM-resident
  for item in $AB; do
    w=${item%%:*}; specs=${item#*:}
    IFS=';' read -ra S <<< "$specs"
    timeout 300 python scripts/exp.py $w "${S[@]}" >> gpurun_out/ab_options.txt 2>&1 < /dev/null
    COLD=1 timeout 400 python scripts/exp.py $w "${S[@]}" >> gpurun_out/ab_options.txt 2>&1 < /dev/null
  done
  cat gpurun_out/ab_options.txt
fi
cd /tmp && export TMPDIR=/tmp
for w in $PROF; do
  for mode in warm cold; do
    flag="--no-cold"; [ "$mode" = "cold" ] && flag="--cold-only"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${w}_$mode -o $w -- python $R/bench.py --steps 100 --warmup 10 --no-cpu $flag --workload $w > $R/gpurun_out/prof_${w}_$mode.log 2>&1 < /dev/null
    f=$(find $R/gpurun_out/prof_${w}_$mode -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && echo "== $w $mode" && head -3 "$f" | cut -c1-200
  done
done
if [ -n "$PROF" ]; then   # the row fold launch by launch: span against the idle gap in front of it
  for w in $PROF; do python $R/scripts/fold_trace.py $R/gpurun_out/prof_${w}_warm; python $R/scripts/fold_trace.py $R/gpurun_out/prof_${w}_cold; done > $R/gpurun_out/fold_trace.txt 2>&1
  cat $R/gpurun_out/fold_trace.txt
fi
for w in $PMC; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${w}_$c -o $w -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-cold --workload $w > $R/gpurun_out/pmc_${w}_$c.log 2>&1 < /dev/null
  done
done
for w in $PMCX; do   # occupancy / stall mix / LDS conflicts / L2 hit rate of the dominant kernels
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmcx_${w}_sq -o $w -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-cold --workload $w > $R/gpurun_out/pmcx_${w}_sq.log 2>&1 < /dev/null
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcx_${w}_tcc -o $w -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-cold --workload $w > $R/gpurun_out/pmcx_${w}_tcc.log 2>&1 < /dev/null
done
if [ -n "$GAPS" ]; then   # one warm cfmm_route under the kernel trace, pre-armed and launch-when-ready
  {
  echo "# rocprofv3 --kernel-trace of scripts/route_gaps.py once (one warm cfmm_route), summarised by scripts/route_gaps.py gaps"
  for w in $GAPS; do for a in 1 0; do
    rm -rf $R/gpurun_out/rt_${w}_$a
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rt_${w}_$a -o t -- python $R/scripts/route_gaps.py once $w $a > $R/gpurun_out/rt_${w}_$a.log 2>&1 < /dev/null
    python $R/scripts/route_gaps.py gaps $R/gpurun_out/rt_${w}_$a "$w a=$a"
  done; done
  } > $R/gpurun_out/route_kernel_gaps.txt 2>&1
  cat $R/gpurun_out/route_kernel_gaps.txt
fi
cd $R
[ -n "$PMC" ] && timeout 60 python scripts/pmc_summary.py gpurun_out $PMC < /dev/null
[ -n "$PMCX" ] && timeout 60 python scripts/pmc_summary.py --extra < /dev/null
true
