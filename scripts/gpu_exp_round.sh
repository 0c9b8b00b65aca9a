mkdir -p gpurun_out; rm -f gpurun_out/exp_xcd.txt
timeout 600 python -m pytest tests/test_gpu_fold.py -m gpu -x -q -k "block_maps" 2>&1 | tail -5
timeout 300 python scripts/exp.py config3 "xcd_map=0" "xcd_map=2" "cost_geomean=14" "cost_geomean=18" "cost_geomean=22" "cost_geomean=26" "cost_geomean=30" "cost_geomean=40" "cost_geomean=22,max_grid=768" "cost_geomean=22,max_grid=1024" 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/exp_xcd.txt
