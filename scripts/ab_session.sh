# A/B of launch geometry x arithmetic for the fused config3 launch under true HBM residency
rm -f gpurun_out/ab_options.txt
for w in config3; do
  timeout 400 python scripts/exp.py $w "" "max_grid=768" "max_grid=1024" "dev_prices_in_window=1" "dev_prices_in_window=1,max_grid=768" "dev_prices_in_window=1,max_grid=1024" "block=1024" "fuse_segments=0" "" >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 600 python scripts/exp.py $w "" "max_grid=768" "max_grid=1024" "dev_prices_in_window=1" "dev_prices_in_window=1,max_grid=768" "dev_prices_in_window=1,max_grid=1024" "block=1024" "fuse_segments=0" "" >> gpurun_out/ab_options.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/ab_options.txt
