"""Evidence for DESIGN 3.6: where the time of one route! evaluation goes on the GPU's own clock, with and without
pre-armed launches.  Reads a rocprofv3 --kernel-trace CSV of `scripts/route_once.py` and prints, for the fused
evaluations of the LAST cfmm_route call: the period of an evaluation (sweep start -> next sweep start), the idle gap
between a fold's end and the next sweep's start, and the sweep's span (armed: includes the wait for the host's v).
usage: python scripts/route_kernel_gaps.py <dir with *_kernel_trace.csv> [label]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "cfmm::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last run of consecutive fused sweeps (sweep_multi<false / sweep_kernel<..., false, ...) each followed by a fold
def fused(n):
    import re
    return ("sweep_multi<false" in n) or re.search(r"sweep_kernel<[^,]+, false,", n) is not None
evals = []
i = len(rows) - 1
while i > 0 and not fused(rows[i - 1][2]):
    i -= 1
# walk back over (sweep, fold) pairs
pairs = []
j = i
while j >= 1 and "reduce_partials" in rows[j][2] and fused(rows[j - 1][2]):
    pairs.append((rows[j - 1], rows[j]))
    j -= 2
pairs.reverse()
if len(pairs) < 3:
    print(label, "too few evaluations found:", len(pairs))
    sys.exit(0)
period = [(pairs[k + 1][0][0] - pairs[k][0][0]) / 1e3 for k in range(len(pairs) - 1)]
gap = [(pairs[k + 1][0][0] - pairs[k][1][1]) / 1e3 for k in range(len(pairs) - 1)]
sweep = [(p[0][1] - p[0][0]) / 1e3 for p in pairs]
fold = [(p[1][1] - p[1][0]) / 1e3 for p in pairs]
s2f = [(p[1][0] - p[0][1]) / 1e3 for p in pairs]
med = lambda x: sorted(x)[len(x) // 2]
print(f"{label:10s} evaluations {len(pairs):3d} | period {med(period):6.2f} us | fold end -> next sweep start {med(gap):6.2f} us | "
      f"sweep span {med(sweep):6.2f} us | sweep end -> fold start {med(s2f):5.2f} us | fold span {med(fold):5.2f} us")
