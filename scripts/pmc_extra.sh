#!/bin/bash
# Extra PMC passes for the dominant kernels (occupancy / stall mix / LDS conflicts / L2 hit rate).
cd /tmp && export TMPDIR=/tmp
for w in ${1:-config3 product1m}; do
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcx_${w}_sq -o $w -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-cold --workload $w > $GRAFT_REPO_ROOT/gpurun_out/pmcx_${w}_sq.log 2>&1 < /dev/null
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcx_${w}_tcc -o $w -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-cold --workload $w > $GRAFT_REPO_ROOT/gpurun_out/pmcx_${w}_tcc.log 2>&1 < /dev/null
done
cd $GRAFT_REPO_ROOT
timeout 60 python - <<'PY'
import csv, glob, json, os
out = {}
for d in sorted(glob.glob("gpurun_out/pmcx_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "cfmm" in row["Kernel_Name"]:
                acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(os.path.basename(d), {}).setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/pmc_extra.json", "w"), indent=1)
for run, ks in out.items():
    for k, cs in ks.items():
        if "sweep" in k and ("true" in k):
            print(run, k[:70], {c: round(x, 1) for c, x in cs.items()})
PY
