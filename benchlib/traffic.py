"""roofline.traffic measured live: two bounded rocprofv3 PMC child runs of the bench's own command."""
import os
import sys


def live_traffic(bench_script, workload, fused, opts):
    """roofline.traffic measured NOW: HBM bytes per sweep launch from two bounded rocprofv3 PMC passes of this very command
    (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs with --kernel-trace only, MI355X_MICROARCH.md §HBM: both
    in KiB, FETCH_SIZE doubled on gfx950).  Returns (bytes, detail) or (None, reason).  The child runs are this script with
    --no-cpu --no-cold --no-live-traffic; each is bounded, and on a time-out exactly the process group started here is
    killed."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if os.environ.get("CFMM_BENCH_CHILD") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "already inside a profiled run"
    tmp = tempfile.mkdtemp(prefix="cfmm_pmc_", dir="/tmp")
    env = dict(os.environ, CFMM_BENCH_CHILD="1", TMPDIR="/tmp")
    counters = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, c), "-o", "w", "--",
                   sys.executable, os.path.abspath(bench_script), "--steps", "20", "--warmup", "3", "--no-cpu", "--no-cold",
                   "--no-live-traffic", "--workload", workload] + (["--fused"] if fused else [])
            for o in opts:
                cmd += ["--opt", o]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=90)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)     # the session started above, nothing else
                p.wait()
                return None, f"rocprofv3 --pmc {c} timed out"
            acc = {}
            for f in glob.glob(os.path.join(tmp, c, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == c and "cfmm::sweep" in row["Kernel_Name"]:
                        acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not acc:
                return None, f"rocprofv3 --pmc {c}: no counter rows (exit code {p.returncode})"
            counters[c] = {k: sum(v) / len(v) for k, v in acc.items()}
        tag = ("<false,", " false,") if fused else ("<true,", " true,")
        total, detail = 0.0, {}
        for k in sorted(set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"])):
            if not any(t in k for t in tag):
                continue
            f, w = counters["FETCH_SIZE"].get(k, 0.0), counters["WRITE_SIZE"].get(k, 0.0)
            detail[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
            total += detail[k]["hbm_bytes_per_launch"]
        return (total, detail) if detail else (None, "no sweep kernel of this variant in the counter rows")
    except Exception as e:      # a profiler problem must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


