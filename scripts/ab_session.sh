# A/B of the trade-store policy (option stream_stores: 1 = write-through, 2 = non-temporal), warm then HBM-resident
rm -f gpurun_out/ab_options.txt
for w in config3 product1m config4shard config5 univ3_ticks config2; do
  timeout 300 python scripts/exp.py $w "stream_stores=1" "stream_stores=2" "stream_stores=1" "stream_stores=2" >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 400 python scripts/exp.py $w "stream_stores=1" "stream_stores=2" "stream_stores=1" "stream_stores=2" >> gpurun_out/ab_options.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/ab_options.txt
