"""Prints the measured table of DESIGN.md §3.4 from the committed profiles (profiles/rNN_*): usage python scripts/design_table.py [r06]"""
import csv, json, os, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def stats(w, mode):
    rows = list(csv.DictReader(open(os.path.join(P, f"{rnd}_{w}_{mode}_kernel_stats.csv"))))
    sw = [r for r in rows if "sweep" in r["Name"]][0]
    fo = [r for r in rows if "reduce" in r["Name"]]
    return float(sw["AverageNs"]) / 1e3, (float(fo[0]["AverageNs"]) / 1e3 if fo else float("nan"))
def sci(x):
    e = int(f"{x:e}".split("e")[1]); m = x / 10 ** e
    return f"{m:.2f}×10" + "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(c)] for c in str(e))
tr = json.load(open(os.path.join(P, "traffic.json")))
names = {"config3": "**config3**: 500k Product + 500k GeoMean, 256 tokens", "product1m": "product1m: 1M Product, 256 tokens",
         "config4shard": "config4shard: 500k Product, 512 tokens (one GPU's share of config 4)", "config5": "config5: 1M BoundedProduct, BasketLiquidation",
         "config2": "config2: 100k Product, 64 tokens", "univ3_ticks": "univ3_ticks: 1M UniV3, ragged 2..64 ticks (17M ticks)"}
print("| workload | step µs warm (HBM-resident) | pool-evals/s warm (HBM-resident) | sweep kernel µs warm / **HBM-resident** | fold µs | reference MB → PMC MB | **frac** warm / HBM-resident | **bus** HBM-resident |")
print("|---|---|---|---|---|---|---|---|")
for w, label in names.items():
    b = json.load(open(os.path.join(P, f"{rnd}_bench_{w}.json"))); r = b["roofline"]
    warm, fw = stats(w, "warm"); cold, _ = stats(w, "cold"); ab = r["alg_bytes_per_launch"]; t = tr[w]
    print(f"| {label} | {1e3*b['ms_per_step']:.1f} ({1e3*b['ms_per_step_hbm_resident']:.1f}) | {sci(b['value'])} ({sci(b['value_hbm_resident'])}) | "
          f"{warm:.2f} / **{cold:.2f}** | {fw:.2f} | {ab/1e6:.1f} → {t/1e6:.1f} | {ab/warm/8e6:.3f} / {ab/cold/8e6:.3f} | {t/cold/8e6:.3f} |")
b = json.load(open(os.path.join(P, f"{rnd}_bench_config3.json")))
a = b["roofline"].get("at_scale", {})
if "kernel_ms" in a:
    print(f"| ProductTwoCoin at 8M pools (`roofline.at_scale` on the default line) | {1e3*a['ms_per_step']:.0f} | {sci(a['value'])} | — / {1e3*a['kernel_ms']:.1f} | | 512 → 320 | — / {a['frac']:.3f} | {a['bus_frac']:.2f} |")
print()
r = b["roofline"]; e = r["expanded"]
print(f"line: kernel_ms {1e3*r['kernel_ms']:.2f} frac {r['frac']:.3f} frac_bus {r['frac_bus']:.3f}; expanded step {1e3*e['expanded']['ms_per_step']:.1f} vs {1e3*e['expanded']['ms_per_step_compact_only']:.1f} "
      f"(+{1e3*e['expanded']['expand_ms']:.1f}); plain_rows frac {e['plain_rows']['frac']:.3f} kernel {1e3*e['plain_rows']['kernel_ms_hbm_resident']:.2f}")
for w in names:
    b = json.load(open(os.path.join(P, f"{rnd}_bench_{w}.json"))); rt = b["route"]; p = b["parity"]; c = b["cpu_baseline"]
    print(f"{w}: route {rt['gpu_native_solver_ms']:.3f} ms, {rt['native_evaluations']} evaluations, host solver {rt['native_host_solver_ms']:.3f}; vs Fortran {p.get('route_native_vs_fortran_netflow_rel_err'):.1e}; "
          f"cpu {c['value']:.3g} pools/s on {c['threads']} threads (1 thread {c['value_1thread']:.3g})")
print(b["config"].get("cfg_config4_full")) if False else None
b3 = json.load(open(os.path.join(P, f"{rnd}_bench_config3.json")))
print(b3["config"].get("cfg_config4_full"))
g = json.load(open(os.path.join(P, f"{rnd}_bench_scaling.json")))["grid"]
print("grid:", [(x["m"], x["n_tokens"], round(x["native_ms"], 3)) for x in g if x["m"] in (100, 1292, 10000)])
for t in ("peer", "rccl"):
    w = json.load(open(os.path.join(P, f"{rnd}_bench_torchrun_world1_{t}.json")))
    print(t, "N=1 ratio", round(w["plain_n1"]["ratio_torchrun_over_plain"], 3), {k: round(v.get("ms_per_step", 0) * 1e3, 2) for k, v in w["collectives"].items()})
print(open(os.path.join(P, f"{rnd}_route_kernel_gaps.txt")).read().strip().splitlines()[-2:])
print(open(os.path.join(P, f"{rnd}_solver_bench.txt")).read())
