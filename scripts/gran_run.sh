R=$PWD; mkdir -p gpurun_out
timeout 120 scripts/native/gran.bin > gpurun_out/gran_times.txt 2>&1; cat gpurun_out/gran_times.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/gran_fetch -o g -- $R/scripts/native/gran.bin > $R/gpurun_out/gran_fetch.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/gpurun_out/gran_req -o g -- $R/scripts/native/gran.bin > $R/gpurun_out/gran_req.log 2>&1
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $R/gpurun_out/gran_tcc -o g -- $R/scripts/native/gran.bin > $R/gpurun_out/gran_tcc.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gran_fetch","gran_req","gran_tcc"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==",d)
        for k,v in acc.items():
            print(k[:60].ljust(60), {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
tail -3 gpurun_out/gran_req.log
