"""Where does the CPU restatement's time go, evaluation by evaluation?  (cpu_baseline moved 1.35e7 ..
2.39e7 pools/s over four rounds.)  No torch, no HIP in this process: the oracle's OpenMP sweep + the two serial reductions on
the config-3 market, timed per evaluation under several thread counts / OpenMP placements (each in its own subprocess:
placement is fixed when the OpenMP runtime starts).  python scripts/cpu_baseline_probe.py [workload]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from benchlib.workloads import WORKLOADS, build_market, sweep_prices_for
    from helpers import oracle_poolset
    from oracle import cfmm_oracle as orc
    name, threads, reps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    n = WORKLOADS[name][1]
    ps = oracle_poolset(build_market(name, 0, 1, "weak"), n)
    v = sweep_prices_for(name, n)
    D, L, G = np.empty((ps.m, 2)), np.empty((ps.m, 2)), np.zeros(n)
    ps.sweep_into(v, D, L, threads)
    ts, tr = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        ps.sweep_into(v, D, L, threads)
        t1 = time.perf_counter()
        orc.dual_acc(D, L, ps.Ai, v)
        G[:] = 0.0
        orc.grad_scatter(G, D, L, ps.Ai)
        t2 = time.perf_counter()
        ts.append(1e3 * (t1 - t0))
        tr.append(1e3 * (t2 - t1))
    q = lambda a: [round(float(np.quantile(a, p)), 3) for p in (0.05, 0.25, 0.5, 0.75, 0.95, 1.0)]
    tot = np.array(ts) + np.array(tr)
    print(json.dumps({"threads": threads, "reps": reps, "sweep_ms_q": q(ts), "reductions_ms_q": q(tr), "evaluation_ms_q": q(tot),
                      "pools_per_s_mean": ps.m * reps / (tot.sum() * 1e-3), "pools_per_s_median": ps.m / (np.median(tot) * 1e-3),
                      "slow_evaluations": int(np.sum(tot > 3 * np.median(tot)))}))
    sys.exit(0)

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
ncpu = len(os.sched_getaffinity(0))
print(f"# {name}; logical CPUs visible {ncpu}; quantiles = 5 / 25 / 50 / 75 / 95 / 100 %, ms per evaluation")
variants = [("default placement", {}), ("OMP_PLACES=cores OMP_PROC_BIND=spread", {"OMP_PLACES": "cores", "OMP_PROC_BIND": "spread"}),
            ("OMP_PLACES=cores OMP_PROC_BIND=close", {"OMP_PLACES": "cores", "OMP_PROC_BIND": "close"}),
            ("spread + OMP_WAIT_POLICY=active", {"OMP_PLACES": "cores", "OMP_PROC_BIND": "spread", "OMP_WAIT_POLICY": "active"}),
            ("spread + OMP_WAIT_POLICY=passive", {"OMP_PLACES": "cores", "OMP_PROC_BIND": "spread", "OMP_WAIT_POLICY": "passive"})]
for label, env in variants:
    for threads in sorted({1, 16, 64, min(128, max(1, ncpu // 2)), ncpu}):
        if threads == 1 and env:
            continue
        e = dict(os.environ)
        for k in ("OMP_PLACES", "OMP_PROC_BIND", "OMP_WAIT_POLICY", "OMP_NUM_THREADS"):
            e.pop(k, None)
        e.update(env)
        reps = 20 if threads == 1 else 150
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, str(threads), str(reps)], env=e,
                           capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        print(f"{label:44s} {line[-1] if line else 'FAILED ' + p.stderr[-300:]}", flush=True)
