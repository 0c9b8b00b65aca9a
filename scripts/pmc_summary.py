"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into per-kernel HBM traffic per launch.

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports 1/2 of a wide coalesced read
stream -> doubled here.  WRITE_SIZE is taken as reported (uncalibrated per the guide).

    python scripts/pmc_summary.py gpurun_out WORKLOAD ...     -> gpurun_out/traffic.json
    python scripts/pmc_summary.py --extra                      the SQ / TCC passes of gpu_round.sh (PMCX=...) ->
                                                               gpurun_out/pmc_extra.json: mean counter value per launch"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

if len(sys.argv) > 1 and sys.argv[1] == "--extra":
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
    sys.exit(0)

root, workloads = sys.argv[1], sys.argv[2:]
out = {}
for w in workloads:
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(root, f"pmc_{w}_{c}", "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(list)
        for f in files:
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == c:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        per[c] = {k: sum(v) / len(v) for k, v in acc.items() if v}
    kernels = sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]))
    for variant, tags in (("materialising", (" true,", "<true,")), ("fused", (" false,", "<false,"))):
        tot = 0.0
        detail = {}
        for k in kernels:
            if "sweep" not in k or not any(t in k for t in tags):
                continue
            f, wr = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
            b = (2.0 * f + wr) * 1024.0
            detail[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": b}
            tot += b
        key = w if variant == "materialising" else w + "_fused"
        out[key] = tot if detail else None
        out[key + "_detail"] = detail
        print(key, "traffic bytes/step =", tot)
        for k, d in detail.items():
            print("   ", k[:100], d)
json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)
