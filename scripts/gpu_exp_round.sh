mkdir -p gpurun_out; rm -rf gpurun_out/rt_*; R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in config3 config5; do for a in 1 0; do
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rt_${w}_$a -o rt -- python $R/scripts/route_once.py $w $a > $R/gpurun_out/rt_${w}_$a.log 2>&1 < /dev/null
tail -1 $R/gpurun_out/rt_${w}_$a.log
python $R/scripts/route_kernel_gaps.py $R/gpurun_out/rt_${w}_$a "$w a=$a" | tee -a $R/gpurun_out/route_kernel_gaps.txt
done; done
