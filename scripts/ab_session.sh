# One A/B session on the GPU box (gpurun -- 'bash scripts/ab_session.sh'): options of the shipped library, two interleaved repetitions,
# cache-warm then HBM-resident (COLD=1: ring of market copies >= 2 x the Infinity Cache).  Edit WORKLOADS / SPECS as needed; this is the
# session behind profiles/r06_ab_stream_stores.txt.
WORKLOADS=${WORKLOADS:-"config3 product1m config4shard config5 univ3_ticks config2"}
SPECS=${SPECS:-"stream_stores=1 stream_stores=2 stream_stores=1 stream_stores=2"}
rm -f gpurun_out/ab_options.txt
for w in $WORKLOADS; do
  timeout 300 python scripts/exp.py $w $SPECS >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 400 python scripts/exp.py $w $SPECS >> gpurun_out/ab_options.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/ab_options.txt
