// lbfgsb.h -- a from-scratch C++ implementation of L-BFGS-B (limited-memory BFGS with box
// constraints), the outer solver route! hands its dual problem to (src/router.jl:60,105 call
// LBFGSB.jl, a wrapper of the Fortran L-BFGS-B 3.0 which is NOT part of the reference tree).
//
// Written from the published algorithm, not from the Fortran:
//   Byrd, Lu, Nocedal, Zhu, "A limited memory algorithm for bound constrained optimization",
//   SIAM J. Sci. Comput. 16 (1995)   -- generalized Cauchy point (Alg. CP), compact L-BFGS
//   representation B = θI − W M Wᵀ, direct primal subspace minimization (§5.1);
//   Morales, Nocedal, "Remark on Algorithm 778" (2011) -- projected subspace step of v3.0;
//   Moré, Thuente, "Line search algorithms with guaranteed sufficient decrease" (1994).
// Same knobs as the reference's call: m, factr, pgtol, maxfun, maxiter (src/router.jl:58).
//
// Host-only, O(n·m) per iteration with n = n_tokens: it exists so that route! can run with no
// Python/Julia interpreter between two device sweeps.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace cfmm {

struct LbfgsbOptions {
    int m = 5;              // history pairs (route! default, src/router.jl:58)
    double factr = 1e1;     // stop when (f_k − f_{k+1}) / max(|f_k|,|f_{k+1}|,1) <= factr·eps
    double pgtol = 1e-5;    // stop when max_i |proj g_i| <= pgtol
    int maxfun = 15000;
    int maxiter = 15000;
    int max_linesearch = 20;
    // The Fortran code calls a problem "boxed" when every nbd[i] == 2, whatever the bound VALUES
    // are, and then takes a unit first step instead of min(1/|d|, stpmx).  The reference passes
    // nbd = 2 with an infinite upper bound (src/router.jl:67-70), i.e. it runs boxed; SciPy maps
    // an infinite bound to "no bound" and therefore does not.  true = the reference's behaviour.
    bool boxed_from_nbd = false;
    // At convergence the objective sits on its rounding-noise floor (route!'s factr = 1e1 asks for a relative
    // decrease of 10 eps, while the dual value is a sum over 10^6 pools): a trial point then returns f one ulp
    // ABOVE the current one, Moré-Thuente rejects it, and the line search burns up to max_linesearch
    // evaluations shrinking the step among values that differ in the last bit (observed: 14 vs 24
    // evaluations of config3 depending on nothing but the summation order of Ψ).  true: a trial point that
    // does not decrease f, but differs from it by no more than the factr tolerance itself, ends the run with
    // the current iterate.  This is NOT what L-BFGS-B 3.0 (the reference's solver) does, so it is off unless asked
    // for (cfmm_set_option "stop_in_noise"): fewer evaluations, at the price of stopping up to a decade further from
    // the reference's v* (measured: config3 netflows 8e-9 -> 7e-8 of max|Psi| from the CPU restatement).
    bool stop_in_noise = false;
};

struct LbfgsbResult {
    double f = 0.0;
    int iterations = 0;
    int evaluations = 0;
    int status = 0;         // 0 pgtol, 1 factr, 2 maxiter, 3 maxfun, 4 line search failed, 5 callback error
    double proj_grad = 0.0;
    std::string message;
};

// nbd[i]: 0 unbounded, 1 lower only, 2 both, 3 upper only (the Fortran convention the
// reference's `bounds[1,:] .= 2` refers to, src/router.jl:67-70; infinite bounds are ignored).
// fg(x, g) returns f and fills g; returning a non-finite f aborts with status 5.
using LbfgsbFn = std::function<double(const double* x, double* g)>;

LbfgsbResult lbfgsb_minimize(int n, double* x, const double* lower, const double* upper, const int* nbd,
                             const LbfgsbFn& fg, const LbfgsbOptions& opt);

} // namespace cfmm
