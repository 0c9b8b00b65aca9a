"""Summarise the SQ / TCC counter passes of scripts/gpu_round.sh (PMCX=...) into gpurun_out/pmc_extra.json:
per run and kernel the mean counter value per launch."""
import csv
import glob
import json
import os

out = {}
for d in sorted(glob.glob("gpurun_out/pmcx_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "cfmm" in row["Kernel_Name"]:
                acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(os.path.basename(d), {}).setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/pmc_extra.json", "w"), indent=1)
for run, ks in out.items():
    for k, cs in ks.items():
        if "sweep" in k and "true" in k:
            print(run, k[:70], {c: round(x, 1) for c, x in cs.items()})
