mkdir -p gpurun_out; rm -f gpurun_out/exp_pre.txt
timeout 300 python -m pytest tests/test_gpu_armed.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/exp_pre.txt
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cfmmrouter_amd as cr
import bench
for name in ("config3", "product1m", "config2", "config4shard", "config5"):
    desc, n, build = bench.WORKLOADS[name]
    batches = build(0)
    obj = bench.objective_for(name, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n)
    res = {}
    for armed, pre in ((1, 1), (1, 0), (0, 0), (1, 1), (1, 0)):
        r._backend.ctx.set_option("armed", armed); r._backend.ctx.set_option("armed_prefetch", pre)
        cr.route_(r, v=v0, solver="native")
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); cr.route_(r, v=v0, solver="native"); ts.append(time.perf_counter() - t0)
        res.setdefault((armed, pre), []).append((1e3 * min(ts), r.info["funcalls"], 1e3 * r.info["sweep_seconds"], cr.netflows(r).copy(), r.v.copy()))
    base = res[(0, 0)][0]
    line = name
    for key in ((1, 1), (1, 0), (0, 0)):
        ms = min(x[0] for x in res[key]); sw = min(x[2] for x in res[key])
        same = all(np.array_equal(x[3], base[3]) and np.array_equal(x[4], base[4]) for x in res[key])
        line += f" | armed={key[0]} prefetch={key[1]}: {ms:.4f} ms (sweeps {sw:.4f}) evals {res[key][0][1]} bit-identical {same}"
    print(line)
    r.close()
PY
