// abi_route.cpp -- route!(r) in ONE call (src/router.jl:58-108): the library's own L-BFGS-B (lbfgsb.cpp) drives the
// device evaluations, with the reference's objectives (src/objectives.jl:51-146), bounds (src/router.jl:67-70) and
// v-cache rule (:74, :92).  Also the bare solver for host-side callers.
#include "ctx.h"

#include <new>
#include "lbfgsb.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

using namespace cfmm;

extern "C" {

int cfmm_lbfgsb_minimize(int32_t n, double* x, const double* lower, const double* upper, const int32_t* nbd,
                         cfmm_fg_callback fg, void* user, int32_t m, double factr, double pgtol, int32_t maxfun,
                         int32_t maxiter, int32_t boxed_from_nbd, cfmm_route_info* info)
{   // (the solver's defaults: reference behaviour, no noise-floor stop)
    if (n < 1 || !x || !nbd || !fg) return fail(nullptr, CFMM_ERR_INVALID_ARG, "bad argument to cfmm_lbfgsb_minimize");
    LbfgsbOptions opt;
    opt.boxed_from_nbd = boxed_from_nbd != 0;
    opt.m = m;
    opt.factr = factr;
    opt.pgtol = pgtol;
    opt.maxfun = maxfun;
    opt.maxiter = maxiter;
    std::vector<int> nb(nbd, nbd + n);
    LbfgsbResult r = lbfgsb_minimize(n, x, lower, upper, nb.data(),
                                     [&](const double* xx, double* gg) { return fg(user, xx, gg); }, opt);
    if (info) {
        info->f = r.f;
        info->proj_grad = r.proj_grad;
        info->iterations = r.iterations;
        info->evaluations = r.evaluations;
        info->sweeps = r.evaluations;
        info->status = r.status;
        info->sweep_seconds = 0.0;
        info->total_seconds = 0.0;
    }
    return CFMM_OK;
}

int cfmm_route(cfmm_ctx* c, int32_t objective_kind, const double* objective_vec, int32_t objective_index,
               const double* v0, int32_t m, double factr, double pgtol, int32_t maxfun, int32_t maxiter,
               double* v_out, double* psi_out, cfmm_route_info* info)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    const int n = c->n;
    if (!objective_vec) return fail(c, CFMM_ERR_INVALID_ARG, "objective vector is null");
    if (objective_kind != CFMM_OBJ_LINEAR_NONNEGATIVE && objective_kind != CFMM_OBJ_BASKET_LIQUIDATION)
        return fail(c, CFMM_ERR_INVALID_ARG, "unknown objective kind %d", objective_kind);
    const bool linear = objective_kind == CFMM_OBJ_LINEAR_NONNEGATIVE;
    if (linear) {
        for (int j = 0; j < n; ++j)
            if (!(objective_vec[j] > 0.0)) // src/objectives.jl:54
                return fail(c, CFMM_ERR_INVALID_ARG, "all elements must be strictly positive");
    } else if (objective_index < 0 || objective_index >= n) {
        return fail(c, CFMM_ERR_INVALID_ARG, "Invalid index i"); // src/objectives.jl:97
    }
    const double* ov = objective_vec;
    const int oi = objective_index;

    // bounds: src/router.jl:67-70 with lower_limit/upper_limit of src/objectives.jl:78-79, :123-129
    std::vector<double> lo(n), up(n, INFINITY), v(n), rv(n);
    std::vector<int> nbd(n, 2);
    const double sqrt_eps = std::sqrt(2.220446049250313e-16);
    for (int j = 0; j < n; ++j) lo[j] = linear ? ov[j] + 1e-8 : sqrt_eps;
    if (!linear) lo[oi] = 1.0 + sqrt_eps;
    for (int j = 0; j < n; ++j) rv[j] = v0 ? v0[j] : 1.0 / n; // src/router.jl:61-65

    int sweeps = 0, rc_inner = CFMM_OK;
    double sweep_s = 0.0;
    const auto t_begin = std::chrono::steady_clock::now();
    c->sweep_count = 0;        // tile directions (option "alternate") are a function of this call alone: evaluation k
    for (cfmm_ctx* s : c->shards) s->sweep_count = 0;   // walks in direction k & 1, the final find_arb! forwards
    bool armed = can_arm(c);   // (switched off for the rest of the call after a lost hand-over)
    struct ArmGuard {   // whatever path leaves this function: no launch stays behind waiting for a price vector
        cfmm_ctx* c;
        ~ArmGuard() { armed_cancel(c); }
    } arm_guard{c};
    auto timed_sweep = [&](const double* x, bool mat) {
        const auto t0 = std::chrono::steady_clock::now();
        if (mat) armed_cancel(c);
        int rc;
        if (armed && !mat) {
            bool lost = false;
            rc = armed_eval(c, x, &lost);
            if (lost) armed = false;
        } else {
            rc = host_sweep(c, x, mat);
        }
        sweep_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++sweeps;
        return rc;
    };
    auto sweep = [&](const double* x) { // fused evaluation: Ψ and acc into c->last_out
        rc_inner = timed_sweep(x, false);
        return rc_inner == CFMM_OK;
    };
    // f(objective, v) and grad!(G, objective, v): src/objectives.jl:62-76, :106-121
    auto obj_f = [&](const double* x) -> double {
        if (linear) {
            for (int j = 0; j < n; ++j)
                if (!(ov[j] <= x[j])) return INFINITY;
            return 0.0;
        }
        if (!(x[oi] >= 1.0)) return INFINITY;
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += (j == oi) ? 0.0 : ov[j] * x[j];
        return s;
    };
    auto fg = [&](const double* x, double* G) -> double {
        bool same = true; // src/router.jl:74 / :92: one sweep per evaluation
        for (int j = 0; j < n && same; ++j) same = x[j] == rv[j];
        if (!same) {
            if (!sweep(x)) return NAN;
            std::copy(x, x + n, rv.begin());
        }
        const double fo = obj_f(x);
        const bool feasible = std::isfinite(fo);
        for (int j = 0; j < n; ++j) {
            const double gobj = !feasible ? INFINITY : (linear ? 0.0 : (j == oi ? 0.0 : ov[j]));
            G[j] = gobj + c->last_out[(size_t)j]; // src/router.jl:96-100
        }
        return fo + c->last_out[(size_t)n];        // src/router.jl:85
    };

    if (!sweep(rv.data())) return rc_inner; // src/router.jl:104
    std::copy(rv.begin(), rv.end(), v.begin());
    LbfgsbOptions opt;
    opt.m = m;
    opt.factr = factr;
    opt.pgtol = pgtol;
    opt.maxfun = maxfun;
    opt.maxiter = maxiter;
    opt.boxed_from_nbd = true; // bounds[1,:] .= 2 with an infinite upper limit: the Fortran's "boxed" path
    opt.stop_in_noise = c->opt_stop_in_noise != 0;   // default off: the stopping rules of L-BFGS-B 3.0, nothing else
    LbfgsbResult r = lbfgsb_minimize(n, v.data(), lo.data(), up.data(), nbd.data(), fg, opt); // :105
    if (rc_inner != CFMM_OK) return rc_inner;
    int rc = timed_sweep(v.data(), true); // src/router.jl:106-107: r.v = v*, find_arb!(r, v*)
    if (rc != CFMM_OK) return rc;
    if (v_out) std::copy(v.begin(), v.end(), v_out);
    if (psi_out) std::memcpy(psi_out, c->last_out.data(), (size_t)n * sizeof(double));
    if (info) {
        info->f = r.f;
        info->proj_grad = r.proj_grad;
        info->iterations = r.iterations;
        info->evaluations = r.evaluations;
        info->sweeps = sweeps;
        info->status = r.status;
        info->sweep_seconds = sweep_s;
        info->total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return CFMM_OK;
}

static int polish_body(cfmm_ctx* c, int32_t objective_kind, const double* objective_vec, int32_t objective_index, double* v,
                       int32_t max_iters, double rel_step, double* psi_out, cfmm_polish_info* info);

// EXPERIMENTAL (not one of the reference's verbs; include/cfmm_amd.h, experimental section).  Dense n x n Jacobian on the
// host + n + 1 sweeps: bounded to LDS-path markets (n_tokens <= 8192: 512 MiB of Jacobian at most), no exception crosses
// the C ABI.
int cfmm_polish(cfmm_ctx* c, int32_t objective_kind, const double* objective_vec, int32_t objective_index, double* v,
                int32_t max_iters, double rel_step, double* psi_out, cfmm_polish_info* info)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (global_bins(c))
        return fail(c, CFMM_ERR_UNSUPPORTED, "cfmm_polish builds a dense n_tokens x n_tokens Jacobian: markets with more than %d "
                                             "tokens are not supported", kMaxLdsTokens);
    try {
        return polish_body(c, objective_kind, objective_vec, objective_index, v, max_iters, rel_step, psi_out, info);
    } catch (const std::bad_alloc&) {
        armed_cancel(c);
        return fail(c, CFMM_ERR_HIP, "cfmm_polish: out of host memory (dense %d x %d Jacobian)", c->n, c->n);
    } catch (...) {
        armed_cancel(c);
        return fail(c, CFMM_ERR_STATE, "cfmm_polish: unexpected exception");
    }
}

static int polish_body(cfmm_ctx* c, int32_t objective_kind, const double* objective_vec, int32_t objective_index, double* v,
                       int32_t max_iters, double rel_step, double* psi_out, cfmm_polish_info* info)
{
    const int n = c->n;
    if (!objective_vec || !v) return fail(c, CFMM_ERR_INVALID_ARG, "objective vector / v is null");
    if (objective_kind != CFMM_OBJ_LINEAR_NONNEGATIVE && objective_kind != CFMM_OBJ_BASKET_LIQUIDATION)
        return fail(c, CFMM_ERR_INVALID_ARG, "unknown objective kind %d", objective_kind);
    const bool linear = objective_kind == CFMM_OBJ_LINEAR_NONNEGATIVE;
    if (!linear && (objective_index < 0 || objective_index >= n)) return fail(c, CFMM_ERR_INVALID_ARG, "Invalid index i");
    if (!(rel_step > 0.0)) rel_step = 1e-7;
    const auto t_begin = std::chrono::steady_clock::now();
    armed_cancel(c);
    // bounds and the objective's (constant) gradient: src/objectives.jl:69-79, :113-129
    std::vector<double> lo(n), gobj(n, 0.0), x(n), G(n), xt(n), Gt(n), step(n);
    const double sqrt_eps = std::sqrt(2.220446049250313e-16);
    for (int j = 0; j < n; ++j) {
        lo[j] = linear ? objective_vec[j] + 1e-8 : sqrt_eps;
        if (!linear) gobj[j] = j == objective_index ? 0.0 : objective_vec[j];
    }
    if (!linear) lo[objective_index] = 1.0 + sqrt_eps;
    for (int j = 0; j < n; ++j) x[j] = std::max(v[j], lo[j]);
    int sweeps = 0;
    auto gradient = [&](const std::vector<double>& at, std::vector<double>& out) {   // G = grad f + psi: one fused sweep
        int rc = host_sweep(c, at.data(), false);
        ++sweeps;
        if (rc != CFMM_OK) return rc;
        for (int j = 0; j < n; ++j) out[j] = gobj[j] + c->last_out[(size_t)j];
        return (int)CFMM_OK;
    };
    std::vector<char> free_(n), freet(n);
    auto residual = [&](const std::vector<double>& at, const std::vector<double>& g, std::vector<char>& fr) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) {
            fr[j] = !(at[j] <= lo[j] && g[j] > 0.0);
            if (fr[j]) r = std::max(r, std::fabs(g[j]));
        }
        return r;
    };
    int rc = gradient(x, G);
    if (rc != CFMM_OK) return rc;
    // forward-difference Jacobian (column j = dG / dv_j), row-major J[i*n + j]
    std::vector<double> J(max_iters > 0 ? (size_t)n * n : 0);   // (max_iters <= 0: the residual at v alone, no Jacobian sweeps)
    xt = x;
    for (int j = 0; j < n && max_iters > 0; ++j) {
        xt[j] = x[j] * (1.0 + rel_step);
        rc = gradient(xt, Gt);
        if (rc != CFMM_OK) return rc;
        const double h = xt[j] - x[j];
        for (int i = 0; i < n; ++i) J[(size_t)i * n + j] = (Gt[i] - G[i]) / h;
        xt[j] = x[j];
    }
    double res = residual(x, G, free_);
    const double res0 = res;
    int done = 0;
    std::vector<int> idx;
    std::vector<double> A, b;
    for (int it = 0; it < max_iters && res > 0.0; ++it) {
        idx.clear();
        for (int j = 0; j < n; ++j)
            if (free_[j]) idx.push_back(j);
        const int nf = (int)idx.size();
        A.assign((size_t)nf * nf, 0.0);
        b.assign((size_t)nf, 0.0);
        double amax = 0.0;
        for (int a = 0; a < nf; ++a) {
            for (int q = 0; q < nf; ++q) {
                A[(size_t)a * nf + q] = J[(size_t)idx[a] * n + idx[q]];
                amax = std::max(amax, std::fabs(A[(size_t)a * nf + q]));
            }
            b[a] = -G[idx[a]];
        }
        // Gaussian elimination with partial pivoting; a vanishing pivot (the dual is homogeneous of degree 0 in v: J v = 0,
        // singular when no variable sits on a bound) is replaced by a tiny one -- the step is a search direction, the
        // residual test below decides whether it is taken
        const double tiny = 1e-13 * std::max(amax, 1e-300);
        for (int k = 0; k < nf; ++k) {
            int piv = k;
            for (int r = k + 1; r < nf; ++r)
                if (std::fabs(A[(size_t)r * nf + k]) > std::fabs(A[(size_t)piv * nf + k])) piv = r;
            if (piv != k) {
                for (int q = k; q < nf; ++q) std::swap(A[(size_t)k * nf + q], A[(size_t)piv * nf + q]);
                std::swap(b[k], b[piv]);
            }
            double& d = A[(size_t)k * nf + k];
            if (std::fabs(d) < tiny) d = d < 0 ? -tiny : tiny;
            const double inv = 1.0 / d;
            for (int r = k + 1; r < nf; ++r) {
                const double f = A[(size_t)r * nf + k] * inv;
                if (f == 0.0) continue;
                double* ar = &A[(size_t)r * nf];
                const double* ak = &A[(size_t)k * nf];
                for (int q = k + 1; q < nf; ++q) ar[q] -= f * ak[q];
                b[r] -= f * b[k];
            }
        }
        for (int k = nf - 1; k >= 0; --k) {
            double sum = b[k];
            for (int q = k + 1; q < nf; ++q) sum -= A[(size_t)k * nf + q] * b[q];
            b[k] = sum / A[(size_t)k * nf + k];
        }
        std::fill(step.begin(), step.end(), 0.0);
        for (int a = 0; a < nf; ++a) step[idx[a]] = b[a];
        bool improved = false;
        for (double t = 1.0; t >= 1.0 / 64; t *= 0.5) {
            for (int j = 0; j < n; ++j) xt[j] = std::max(x[j] + t * step[j], lo[j]);
            rc = gradient(xt, Gt);
            if (rc != CFMM_OK) return rc;
            const double rt = residual(xt, Gt, freet);
            if (rt < res) {
                x = xt; G = Gt; free_ = freet; res = rt;
                improved = true;
                break;
            }
        }
        ++done;
        if (!improved) break;   // the rounding-noise floor of psi (or a kink the chord matrix cannot cross): keep the best iterate
    }
    rc = host_sweep(c, x.data(), true);   // find_arb!(r, v) at the polished point, as route! ends (src/router.jl:106-107)
    ++sweeps;
    if (rc != CFMM_OK) return rc;
    std::copy(x.begin(), x.end(), v);
    if (psi_out) std::memcpy(psi_out, c->last_out.data(), (size_t)n * sizeof(double));
    if (info) {
        info->residual0 = res0;
        info->residual = res;
        info->iterations = done;
        info->sweeps = sweeps;
        info->total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return CFMM_OK;
}

} // extern "C"
