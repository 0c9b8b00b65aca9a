python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp" | tail -15 > gpurun_out/r06_t3.log; tail -5 gpurun_out/r06_t3.log
rm -f gpurun_out/ab_options.txt
for w in univ3_ticks; do
  timeout 300 python scripts/exp.py $w "" "coop_records=0" "dev_prices_in_window=1" "dev_prices_in_window=1,coop_records=0" >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 400 python scripts/exp.py $w "" "coop_records=0" "dev_prices_in_window=1" "dev_prices_in_window=1,coop_records=0" >> gpurun_out/ab_options.txt 2>&1
done
for w in config3 product1m config5 config2; do
  timeout 300 python scripts/exp.py $w "" "dev_prices_in_window=1" >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 400 python scripts/exp.py $w "" "dev_prices_in_window=1" >> gpurun_out/ab_options.txt 2>&1
done
cat gpurun_out/ab_options.txt
