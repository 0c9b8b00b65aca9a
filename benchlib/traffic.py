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




def live_kernel_stats(bench_script, workload, fused, opts):
    """The HBM-resident sweep kernel's average span as rocprofv3 itself reports it, from a bounded child run of this very
    command with `--cold-only` under `rocprofv3 --kernel-trace --stats` (no counters: PMC passes serialise and perturb
    kernels) -- the number profiles/rNN_*_cold_kernel_stats.csv holds, so that the line and the committed summaries cannot
    disagree.  Returns ({"kernel": name, "avg_ms": ..., "calls": ...}, None) or (None, reason)."""
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
    tmp = tempfile.mkdtemp(prefix="cfmm_kt_", dir="/tmp")
    env = dict(os.environ, CFMM_BENCH_CHILD="1", TMPDIR="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "w", "--",
               sys.executable, os.path.abspath(bench_script), "--steps", "100", "--warmup", "10", "--no-cpu", "--cold-only",
               "--no-live-traffic", "--workload", workload] + (["--fused"] if fused else [])
        for o in opts:
            cmd += ["--opt", o]
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            p.wait(timeout=150)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)     # the session started above, nothing else
            p.wait()
            return None, "rocprofv3 --kernel-trace --stats timed out"
        tag = ("<false,", " false,") if fused else ("<true,", " true,")
        best = None
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Name", "")
                if "cfmm::sweep" in name and any(t in name for t in tag):
                    rec = {"kernel": name, "avg_ms": float(row["AverageNs"]) * 1e-6, "calls": int(row["Calls"]),
                           "min_ms": float(row["MinNs"]) * 1e-6, "max_ms": float(row["MaxNs"]) * 1e-6}
                    if best is None or rec["calls"] * rec["avg_ms"] > best["calls"] * best["avg_ms"]:
                        best = rec
        return (best, None) if best else (None, f"no sweep kernel in the kernel stats (exit code {p.returncode})")
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
