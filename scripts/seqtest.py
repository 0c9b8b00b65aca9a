"""cfmm_route on the six BASELINE-size workloads against the Fortran L-BFGS-B fixture (tests/golden/route_fortran.npz): evaluations and
max|dPsi| / max|Psi|.  Round 6 used it to find which of the solver's sums may be reassociated without moving the stopping point of the
interior-optimum markets inside the rounding noise (csrc/lbfgsb.cpp, note above dotn).  usage: python scripts/seqtest.py"""
import sys, os, json, numpy as np
sys.path.insert(0, os.getcwd())
import cfmmrouter_amd as cr
from benchlib.workloads import build_market, objective_for, WORKLOADS
g = np.load("tests/golden/route_fortran.npz")
for name in ("config5", "univ3_ticks", "config4shard", "product1m", "config3", "config2"):
    n = WORKLOADS[name][1]
    obj = objective_for(name, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, build_market(name, 0, 1, "weak"), n)
    cr.route_(r, v=v0, solver="native")
    psi_f = g["full_" + name + "_psi"]
    print(os.environ.get("CFMM_SOLVER_SEQ", "0"), name, r.info["funcalls"], "%.3e" % (np.max(np.abs(cr.netflows(r) - psi_f)) / np.max(np.abs(psi_f))), flush=True)
    r.close()
