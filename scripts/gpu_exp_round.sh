mkdir -p gpurun_out
for w in config3 product1m; do
timeout 300 python scripts/exp.py $w "inline_fold=0" "inline_fold=0,nt_stores=1" "inline_fold=0,nt_stores=2" "inline_fold=1,nt_stores=2" "inline_fold=0,nt_stores=2,max_grid=1024" "inline_fold=0,nt_stores=2,max_grid=768" "inline_fold=0,nt_stores=2,block=256,max_grid=2048" "inline_fold=0,nt_stores=2,block=1024,max_grid=512" 2>&1 | tee gpurun_out/exp_$w.txt
done
