"""Per-launch view of the row fold (reduce_partials) from a rocprofv3 --kernel-trace CSV: its span against the idle gap in
front of it.  python scripts/fold_trace.py gpurun_out/prof_config3_warm [label]

Why: the --stats summary shows MinNs 2.0-2.5 us against an average of 4.3-4.7 us for the fold.  This
groups the fold launches by how long the GPU had been idle when they started, and prints span, gap + span (what the step
pays from the end of the sweep to the end of the fold) for each group."""
import csv
import glob
import os
import statistics as st
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end, prev_name, folds, sweeps = None, None, [], []
for r in rows:
    s, e, k = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
    if "reduce_partials" in k and prev_name and "sweep" in prev_name:
        folds.append((s - prev_end, e - s))
    if "cfmm::sweep" in k:
        sweeps.append(e - s)
    prev_end, prev_name = e, k
print(f"# {label}: {len(folds)} fold launches behind a sweep; sweep span mean {st.mean(sweeps) / 1e3:.2f} us")
print("# idle gap in front of the fold (sweep end -> fold start) | launches | fold span us: mean (min..max) | gap + span us: mean")
for lo, hi, name in ((-10**9, 1000, "< 1 us   (back to back: the timed region)"), (1000, 3000, "1-3 us"), (3000, 6000, "3-6 us   (the host was late)"),
                     (6000, 10**12, "> 6 us   (kernel events attached / idle stream)")):
    g = [(a, b) for a, b in folds if lo <= a < hi]
    if not g:
        continue
    sp = [b for _, b in g]
    print(f"{name:50s} {len(g):4d}   {st.mean(sp) / 1e3:5.2f} ({min(sp) / 1e3:.2f}..{max(sp) / 1e3:.2f})   {st.mean(a + b for a, b in g) / 1e3:6.2f}")
short = [(a, b) for a, b in folds if b < 3000]
print(f"# folds with a span below 3 us: {len(short)}; the smallest idle gap in front of any of them: "
      f"{min((a for a, _ in short), default=0) / 1e3:.2f} us; smallest gap + span among them {min((a + b for a, b in short), default=0) / 1e3:.2f} us")
b2b = [b for a, b in folds if a < 1000]
if b2b:
    print(f"# back-to-back folds: span min {min(b2b) / 1e3:.2f} us, 5 % quantile {sorted(b2b)[len(b2b) // 20] / 1e3:.2f} us")
