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
    print(name, "%.3e pools/s" % d["value"], "ms/step %.4f" % d["ms_per_step"],
          "sweep_ms %.4f reduce_ms %.4f frac %.3f" % (r["kernel_ms"], r["reduce_kernel_ms"], r["frac"]),
          [(s["block"], s["grid"]) for s in d["config"]["segments"]])
