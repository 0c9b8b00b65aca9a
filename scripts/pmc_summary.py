"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into per-kernel HBM traffic per launch.

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports 1/2 of a wide coalesced read
stream -> doubled here.  WRITE_SIZE is taken as reported (uncalibrated per the guide)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

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
