"""Evidence for DESIGN 3.6: where the time of one route! evaluation goes on the GPU's own clock, with and without
pre-armed launches.
    python scripts/route_gaps.py once WORKLOAD ARMED        one warm cfmm_route on a bench workload (run it under
                                                            rocprofv3 --kernel-trace --output-format csv -d DIR)
    python scripts/route_gaps.py gaps DIR [label]           reads DIR's *_kernel_trace.csv and prints, for the fused evaluations
                                                            of the LAST cfmm_route call: the period of an evaluation (sweep start
                                                            -> next sweep start), the idle gap between a fold's end and the next
                                                            sweep's start, and the sweep's span (armed: includes the wait for v)"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def once(name, armed):
    import numpy as np
    import cfmmrouter_amd as cr
    from benchlib.workloads import WORKLOADS, build_market, objective_for
    desc, n, _ = WORKLOADS[name]
    batches = build_market(name, 0, 1, "weak")
    obj = objective_for(name, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n)
    r._backend.ctx.set_option("armed", armed)
    for _ in range(3):
        cr.route_(r, v=v0, solver="native")
    print(name, "armed", armed, "evaluations", r.info["funcalls"], "route ms", 1e3 * r.info["total_seconds"])
    r.close()


def gaps(d, label=""):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cfmm::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()

    def fused(n):   # sweep_multi<false, ... / sweep_kernel<Ops, false, ...
        return ("sweep_multi<false" in n) or re.search(r"sweep_kernel<[^,]+, false,", n) is not None

    i = len(rows) - 1
    while i > 0 and not fused(rows[i - 1][2]):
        i -= 1
    pairs, j = [], i        # walk back over (sweep, fold) pairs
    while j >= 1 and "reduce_partials" in rows[j][2] and fused(rows[j - 1][2]):
        pairs.append((rows[j - 1], rows[j]))
        j -= 2
    pairs.reverse()
    if len(pairs) < 3:
        print(label, "too few evaluations found:", len(pairs))
        return
    period = [(pairs[k + 1][0][0] - pairs[k][0][0]) / 1e3 for k in range(len(pairs) - 1)]
    gap = [(pairs[k + 1][0][0] - pairs[k][1][1]) / 1e3 for k in range(len(pairs) - 1)]
    sweep = [(p[0][1] - p[0][0]) / 1e3 for p in pairs]
    fold = [(p[1][1] - p[1][0]) / 1e3 for p in pairs]
    s2f = [(p[1][0] - p[0][1]) / 1e3 for p in pairs]
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"{label:10s} evaluations {len(pairs):3d} | period {med(period):6.2f} us | fold end -> next sweep start {med(gap):6.2f} us | "
          f"sweep span {med(sweep):6.2f} us | sweep end -> fold start {med(s2f):5.2f} us | fold span {med(fold):5.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "once":
        once(sys.argv[2], int(sys.argv[3]))
    else:
        gaps(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
