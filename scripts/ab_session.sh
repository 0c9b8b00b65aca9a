# A/B of the last-arriving-block fold (option tail_fold) on the launches it applies to
rm -f gpurun_out/ab_options.txt
for w in config2; do
  timeout 300 python scripts/exp.py $w "tail_fold=1" "tail_fold=0" "tail_fold=1" "tail_fold=0" >> gpurun_out/ab_options.txt 2>&1
  COLD=1 timeout 400 python scripts/exp.py $w "tail_fold=1" "tail_fold=0" "tail_fold=1" "tail_fold=0" >> gpurun_out/ab_options.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/ab_options.txt
python - <<'PY'
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import OBJ_LINEAR_NONNEGATIVE
for m, n in ((3594, 60), (10000, 100), (10000, 400), (30000, 64), (100000, 64)):
    b = [synth.product_pools(m, n, seed=1234)]; c = synth.linear_prices(n, seed=1234)
    for tf in (1, 0, 1, 0):
        be = cr.DeviceBackend(n, b); be.ctx.set_option("tail_fold", tf)
        for _ in range(3): be.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=np.ones(n))
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); v, psi, info = be.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=np.ones(n)); ts.append(time.perf_counter() - t0)
        print(m, n, "tail_fold", tf, "route ms %.4f" % (1e3 * np.median(ts)), info["evaluations"], be.ctx.segments()[0]["grid"], flush=True)
        be.close()
PY
