"""Parts of bench.py (repo root): workloads, the per-rank timed machinery, the GPU-side legs, live PMC traffic.
bench.py itself keeps main() -- the timed region -- and the one leg that touches oracle/ (cpu_baseline_leg)."""
