"""One-line summary of a bench.py log (the last JSON line): python scripts/show_bench.py LOG NAME"""
import json
import sys

path, name = sys.argv[1], sys.argv[2]
lines = [x for x in open(path) if x.startswith("{")]
if not lines:
    print(name, "FAILED")
    print(open(path).read()[-1500:])
else:
    d = json.loads(lines[-1])
    r = d["roofline"]
    warm = r.get("warm") or {}
    print(name, "%.3e pools/s" % d["value"], "ms/step %.4f" % d["ms_per_step"],
          "sweep us warm %.2f / hbm %.2f, fold %.2f, frac hbm %.3f warm %.3f step %.3f layout %.3f" % (
              1e3 * warm.get("kernel_ms", 0), 1e3 * r["kernel_ms"], 1e3 * r["reduce_kernel_ms"], r["frac"], warm.get("frac", 0),
              r["step_frac"], r["layout"]["frac"]),
          [(s["block"], s["grid"]) for s in d["config"]["segments"]])
    for key in ("route", "route_sharded", "parity", "host_boundary", "strong_scaling"):
        if key in d:
            print("   ", key, json.dumps({k: (round(v, 4) if isinstance(v, float) and abs(v) > 1e-3 else v) for k, v in d[key].items()
                                          if not isinstance(v, str) or len(v) < 60}))
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("    cpu_baseline", {k: c[k] for k in c if k != "sample"})
