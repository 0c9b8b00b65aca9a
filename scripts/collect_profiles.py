"""Copies the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked), named per round."""
import csv
import glob
import json
import os
import shutil
import sys

import subprocess

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
# The artefacts must belong to the tree they are committed with: gpu_round.sh records `git rev-parse HEAD`... which a GPU
# box cannot know (the snapshot carries no .git), so the SESSION's head is written here, next to the artefacts, by
# `python scripts/collect_profiles.py --stamp` BEFORE the gpurun call; collecting refuses when it differs from HEAD or
# when the work tree has uncommitted changes to product sources (override: --force).
head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
stamp = os.path.join(src, "session_head.txt")
if "--stamp" in sys.argv:
    dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "cfmmrouter.jl_amd", "include", "bench.py", "benchlib"],
                           capture_output=True, text=True).stdout.strip()
    os.makedirs(src, exist_ok=True)
    open(stamp, "w").write(head + ("\nDIRTY\n" + dirty if dirty else "") + "\n")
    print("session stamped at", head[:12], "(dirty tree: commit first)" if dirty else "")
    sys.exit(1 if dirty else 0)
if "--force" not in sys.argv:
    got = open(stamp).read().split() if os.path.exists(stamp) else []
    if not got or got[0] != head or "DIRTY" in got:
        sys.exit(f"refusing to collect: gpurun_out/ was produced at {got[0][:12] if got else '(no stamp)'}"
                 f"{' with uncommitted changes' if 'DIRTY' in got else ''}, HEAD is {head[:12]} -- re-run the session on this tree "
                 f"(python scripts/collect_profiles.py --stamp; gpurun ...) or pass --force and say so in the commit")
rnd = [a for a in sys.argv[1:] if not a.startswith("--")][0] if [a for a in sys.argv[1:] if not a.startswith("--")] else rnd
for d in glob.glob(os.path.join(src, "prof_*")):
    if not os.path.isdir(d):
        continue
    w = os.path.basename(d)[5:]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{rnd}_{w}_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "tune_*.txt")):
    shutil.copy(f, os.path.join(dst, f"{rnd}_{os.path.basename(f)}"))
if os.path.exists(os.path.join(src, "traffic.json")):
    shutil.copy(os.path.join(src, "traffic.json"), os.path.join(dst, "traffic.json"))
for name in ("floor.txt", "dist_world1.json", "bench_torchrun_world1_peer.json", "bench_torchrun_world1_rccl.json",
             "bench_single_process_3shards.json", "bench_rehearsal_world2.json", "pytest_gpu.log", "smoke.log",
             "route_convergence.txt", "route_kernel_gaps.txt", "ab_options.txt", "ipc_ranks_world2.json", "ipc_ranks_world4.json",
             "fold_trace.txt", "cpu_baseline_probe.txt", "route_fortran_pin.txt", "kernel_resources.txt",
             "ipc_ranks_config4_world8.json", "bench_scaling_grid.json", "solver_bench.txt"):
    if os.path.exists(os.path.join(src, name)):
        out = {"floor.txt": "launch_floor.txt"}.get(name, name)
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{rnd}_{out}"))
if os.path.exists(os.path.join(src, "pmc_extra.json")):
    shutil.copy(os.path.join(src, "pmc_extra.json"), os.path.join(dst, f"{rnd}_pmc_extra.json"))
for f in glob.glob(os.path.join(src, "benchfull_*.log")):
    lines = [x for x in open(f) if x.startswith("{")]
    if lines:
        w = os.path.basename(f)[10:-4]
        json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{rnd}_bench_{w}.json"), "w"), indent=1)
for w in {os.path.basename(d)[4:].rsplit("_", 2)[0] for d in glob.glob(os.path.join(src, "pmc_*_SIZE")) if os.path.isdir(d)}:
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, f"pmc_{w}_{c}", "**", "*counter_collection.csv"), recursive=True)
        acc = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c and "cfmm" in row["Kernel_Name"]:
                    acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        out[c] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
                  for k, v in acc.items()}
    json.dump(out, open(os.path.join(dst, f"{rnd}_{w}_pmc_summary.json"), "w"), indent=1)
print(sorted(os.listdir(dst)))
