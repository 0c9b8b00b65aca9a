mkdir -p gpurun_out; rm -f gpurun_out/exp_tail.txt
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/exp_tail.txt
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
for name, n, batches in (("mixed", 256, [synth.product_pools(300_000, 256, seed=1), synth.geomean_pools(200_000, 256, seed=2)]),
                         ("product", 64, [synth.product_pools(100_000, 64, seed=3)]),
                         ("wide", 512, [synth.product_pools(400_000, 512, seed=4)])):
    v = synth.sweep_prices(n, seed=5)
    out = {}
    for mode in (0, 2, 1):
        be = cr.DeviceBackend(n, batches)
        be.ctx.set_option("inline_fold", mode)
        be.ctx.set_option("alternate", 0)
        r = [be.find_arb(v) for _ in range(5)]
        e = [be.eval(v) for _ in range(5)]
        out[mode] = (r[-1][0], r[-1][1], e[-1][0], e[-1][1])
        be.close()
    for mode in (2, 1):
        same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(out[mode], out[0]))
        print(name, "inline_fold", mode, "bit-identical to separate fold:", same)
PY
for w in config3 product1m config2 config4shard config5; do
timeout 300 python scripts/exp.py $w "inline_fold=0" "inline_fold=2" "inline_fold=0" "inline_fold=2" 2>&1 | grep -v "amdgpu.ids\|^#" | sed "s/^/$w /" | tee -a gpurun_out/exp_tail.txt
done
