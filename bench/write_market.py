"""Writes bench.py's synthetic workloads as market files for bench/reference.jl.

    python bench/write_market.py config3 market_config3.bin
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd import poolfile

import bench as B  # noqa: E402  (bench.py at the repo root)

name, path = sys.argv[1], sys.argv[2]
desc, n, build = B.WORKLOADS[name]
batches = build(0)
obj = B.objective_for(name, n)
v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
poolfile.save_market(path, batches, n, obj, v0)
print(f"wrote {path}: {desc}")
