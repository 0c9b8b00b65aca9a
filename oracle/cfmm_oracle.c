/*
 * cfmm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See cfmm_oracle.h for scope, pinning status and conventions.
 *
 * Build with -ffp-contract=off: the reference (Julia, no @fastmath) never fuses
 * a*b+c, so neither may this file.  sqrt and / are IEEE correctly rounded on
 * both sides; pow is libm's (Julia ships its own pow, which may differ from
 * glibc's in the last ulp -- geometric-mean results are therefore compared
 * with a tolerance, product / UniV3 results bit-for-bit).
 */
#include "cfmm_oracle.h"

#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Julia's max(x::Float64, y::Float64): NaN if either is NaN, +0.0 for (-0.0, 0.0). */
static inline double jl_max(double x, double y)
{
    if (isnan(x) || isnan(y)) return x - y; /* NaN, as Base.max does */
    double diff = x - y;
    return signbit(diff) ? y : x;
}

/* ------------------------------------------------------------------------- */
/* ProductTwoCoin                                                            */
/* ------------------------------------------------------------------------- */

/* src/cfmms.jl:125  prod_arb_δ(m, r, k, γ) = max(sqrt(γ*m*k) - r, 0)/γ   (γ*m*k == (γ*m)*k) */
static inline double prod_arb_delta(double m, double r, double k, double g)
{
    return jl_max(sqrt((g * m) * k) - r, 0.0) / g;
}
/* src/cfmms.jl:126  prod_arb_λ(m, r, k, γ) = max(r - sqrt(k/(m*γ)), 0) */
static inline double prod_arb_lambda(double m, double r, double k, double g)
{
    return jl_max(r - sqrt(k / (m * g)), 0.0);
}

/* src/cfmms.jl:130-140 */
void oracle_product_find_arb(const double R[2], double gamma, const double v[2],
                             double Delta[2], double Lambda[2])
{
    double k = R[0] * R[1];                                   /* :132 */
    Delta[0] = prod_arb_delta(v[1] / v[0], R[0], k, gamma);   /* :134 */
    Delta[1] = prod_arb_delta(v[0] / v[1], R[1], k, gamma);   /* :135 */
    Lambda[0] = prod_arb_lambda(v[0] / v[1], R[0], k, gamma); /* :137 */
    Lambda[1] = prod_arb_lambda(v[1] / v[0], R[1], k, gamma); /* :138 */
}

double oracle_product_phi(const double R[2]) { return R[0] * R[1]; } /* :113-116 */
void oracle_product_grad_phi(const double R[2], double out[2])       /* :117-122 */
{
    out[0] = R[1];
    out[1] = R[0];
}

/* ------------------------------------------------------------------------- */
/* GeometricMeanTwoCoin                                                      */
/* ------------------------------------------------------------------------- */

/* src/cfmms.jl:180  max((γ*m*η*r1*r2^η)^(1/(η+1)) - r2, 0)/γ */
static inline double geom_arb_delta(double m, double r1, double r2, double eta, double g)
{
    double inner = (((g * m) * eta) * r1) * pow(r2, eta);
    return jl_max(pow(inner, 1.0 / (eta + 1.0)) - r2, 0.0) / g;
}
/* src/cfmms.jl:181  max(r1 - ((r2*r1^(1/η))/(η*γ*m))^(η/(1+η)), 0) */
static inline double geom_arb_lambda(double m, double r1, double r2, double eta, double g)
{
    double base = (r2 * pow(r1, 1.0 / eta)) / ((eta * g) * m);
    return jl_max(r1 - pow(base, eta / (1.0 + eta)), 0.0);
}

/* src/cfmms.jl:185-196 (note the swapped argument order at the call sites) */
void oracle_geomean_find_arb(const double R[2], const double w[2], double gamma,
                             const double v[2], double Delta[2], double Lambda[2])
{
    double eta = w[0] / w[1];                                                  /* :188 */
    Delta[0] = geom_arb_delta(v[1] / v[0], R[1], R[0], eta, gamma);            /* :190 */
    Delta[1] = geom_arb_delta(v[0] / v[1], R[0], R[1], 1.0 / eta, gamma);      /* :191 */
    Lambda[0] = geom_arb_lambda(v[0] / v[1], R[0], R[1], 1.0 / eta, gamma);    /* :193 */
    Lambda[1] = geom_arb_lambda(v[1] / v[0], R[1], R[0], eta, gamma);          /* :194 */
}

double oracle_geomean_phi(const double R[2], const double w[2]) /* :167-171 */
{
    return pow(R[0], w[0]) * pow(R[1], w[1]);
}
void oracle_geomean_grad_phi(const double R[2], const double w[2], double out[2]) /* :172-178 */
{
    out[0] = w[0] * pow(R[1] / R[0], w[1]);
    out[1] = w[1] * pow(R[0] / R[1], w[0]);
}

/* ------------------------------------------------------------------------- */
/* UniV3 / BoundedProduct                                                    */
/* ------------------------------------------------------------------------- */

/* src/cfmms.jl:235: last index i (1-based) with lower_ticks[i] >= current_price in a
 * descending vector; 0 when current_price > lower_ticks[1]. */
int64_t oracle_univ3_current_tick(const double* lower_ticks, int64_t n_ticks, double current_price)
{
    int64_t lo = 0, hi = n_ticks + 1; /* Base.searchsortedlast bisection, Reverse ordering */
    while (lo < hi - 1) {
        int64_t mid = lo + ((hi - lo) >> 1);
        /* lt(Reverse, x, a[mid]) == isless(a[mid], x) */
        if (lower_ticks[mid - 1] < current_price) hi = mid;
        else lo = mid;
    }
    return lo;
}

/* src/cfmms.jl:251 / :254-259 */
static inline double tick_high_price(const double* lower_ticks, int64_t idx) { return lower_ticks[idx - 1]; }
static inline double tick_low_price(const double* lower_ticks, int64_t n_ticks, int64_t idx)
{
    if (idx < n_ticks) return lower_ticks[idx];
    return 0.0;
}

/* src/cfmms.jl:294-313 */
oracle_bounded_product oracle_univ3_compute_at_tick(double current_price, int64_t current_tick,
                                                    const double* lower_ticks, const double* liquidity,
                                                    int64_t n_ticks, int64_t idx)
{
    double k = liquidity[idx - 1];
    double pminus = tick_low_price(lower_ticks, n_ticks, idx);
    double pplus = tick_high_price(lower_ticks, idx);
    double alpha = sqrt(k / pplus);
    double beta = sqrt(k * pminus);
    double p;
    if (idx > current_tick) p = pplus;
    else if (idx < current_tick) p = pminus;
    else p = current_price;
    oracle_bounded_product t;
    t.k = k;
    t.alpha = alpha;
    t.beta = beta;
    t.R_1 = sqrt(k / p) - alpha;
    t.R_2 = sqrt(k * p) - beta;
    return t;
}

/* src/cfmms.jl:289 */
oracle_bounded_product oracle_flip_sides(oracle_bounded_product t)
{
    oracle_bounded_product f;
    f.k = t.k;
    f.alpha = t.beta;
    f.beta = t.alpha;
    f.R_1 = t.R_2;
    f.R_2 = t.R_1;
    return f;
}

/* src/cfmms.jl:321-337 */
void oracle_find_arb_pos(oracle_bounded_product t, double price, double* delta, double* lambda)
{
    double d = sqrt(t.k / price) - (t.R_1 + t.alpha); /* :323 */
    if (d <= 0) { *delta = 0.0; *lambda = 0.0; return; } /* :325-327 */
    double d_max = t.k / t.beta - (t.R_1 + t.alpha);  /* :329 (beta==0 -> Inf) */
    if (d >= d_max) { *delta = d_max; *lambda = t.R_2; return; } /* :330-332 */
    *lambda = (t.R_2 + t.beta) - sqrt(price * t.k);   /* :334 */
    *delta = d;
}

/* src/cfmms.jl:339-395 */
void oracle_univ3_find_arb(double current_price, int64_t current_tick, const double* lower_ticks,
                           const double* liquidity, int64_t n_ticks, double gamma, const double v[2],
                           double Delta[2], double Lambda[2])
{
    double p = v[0] / v[1]; /* :340 */
    double g = gamma;
    Delta[0] = Delta[1] = 0.0; /* :343-344 */
    Lambda[0] = Lambda[1] = 0.0;

    if (g * current_price <= p && p <= current_price / g) return; /* :347-349 */

    if (p < g * current_price) { /* :351 */
        int initial = 1;
        for (int64_t idx = current_tick; idx <= n_ticks; ++idx) { /* :316 get_upper_pools */
            oracle_bounded_product pool = oracle_univ3_compute_at_tick(
                current_price, current_tick, lower_ticks, liquidity, n_ticks, idx);
            if (pool.k == 0) { initial = 0; continue; } /* :355-358 */
            double d, l;
            oracle_find_arb_pos(pool, p / g, &d, &l);   /* :361 */
            if (!initial && (d == 0 || l == 0)) break;  /* :363-365 */
            Delta[0] += d;                              /* :366 */
            Lambda[1] += l;                             /* :367 */
            initial = 0;
        }
        Delta[0] /= g; /* :372 */
    } else {
        int initial = 1;
        for (int64_t idx = current_tick; idx >= 1; --idx) { /* :317 get_lower_pools, :375 flip_sides */
            oracle_bounded_product pool = oracle_flip_sides(oracle_univ3_compute_at_tick(
                current_price, current_tick, lower_ticks, liquidity, n_ticks, idx));
            if (pool.k == 0) { initial = 0; continue; } /* :376-379 */
            double d, l;
            oracle_find_arb_pos(pool, 1.0 / (g * p), &d, &l); /* :381 */
            if (!initial && (d == 0 || l == 0)) break;        /* :383-385 */
            Delta[1] += d;                                    /* :386 */
            Lambda[0] += l;                                   /* :387 */
            initial = 0;
        }
        Delta[1] /= g; /* :391 */
    }
}

/* src/cfmms.jl:401-408 */
static double max_amount_pos(oracle_bounded_product t)
{
    if (t.beta > 0) return t.k / t.beta - (t.R_1 + t.alpha);
    else if (t.alpha > 0) return INFINITY; /* typemax(Float64) */
    return 0.0;
}
/* src/cfmms.jl:410-413 */
static double forward_amount(oracle_bounded_product t, double d)
{
    double l = (t.R_2 + t.beta) - t.k / (t.R_1 + t.alpha + d);
    return l < t.R_2 ? l : t.R_2; /* min(t.R_2, λ) */
}

/* src/cfmms.jl:416-449 */
double oracle_univ3_forward_trade(const double Delta[2], double current_price, int64_t current_tick,
                                  const double* lower_ticks, const double* liquidity, int64_t n_ticks,
                                  double gamma)
{
    if (Delta[0] == 0 && Delta[1] == 0) return 0.0; /* :440-442 */
    int up = Delta[0] > 0;                          /* :444 */
    double d = up ? gamma * Delta[0] : gamma * Delta[1];
    double l = 0.0;
    int64_t idx = current_tick;
    for (;;) {
        if (up ? (idx > n_ticks) : (idx < 1)) break;
        oracle_bounded_product pool = oracle_univ3_compute_at_tick(
            current_price, current_tick, lower_ticks, liquidity, n_ticks, idx);
        if (!up) pool = oracle_flip_sides(pool);
        double max_amount = max_amount_pos(pool);   /* :420 */
        if (max_amount > d) {                       /* :422-425 */
            l += forward_amount(pool, d);
            return l;
        }
        l += pool.R_2;                              /* :427 */
        d -= max_amount;                            /* :429 */
        idx += up ? 1 : -1;
    }
    return l; /* :433 */
}

/* ------------------------------------------------------------------------- */
/* Router-level sweep: src/router.jl:38-42                                   */
/* ------------------------------------------------------------------------- */

void oracle_sweep_product(int64_t m, const double* R, const double* gamma, const int32_t* Ai,
                          const double* v, double* Delta, double* Lambda, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < m; ++i) {
        double vi[2] = { v[Ai[2 * i]], v[Ai[2 * i + 1]] }; /* v[r.cfmms[i].Ai] */
        oracle_product_find_arb(R + 2 * i, gamma[i], vi, Delta + 2 * i, Lambda + 2 * i);
    }
}

void oracle_sweep_geomean(int64_t m, const double* R, const double* w, const double* gamma,
                          const int32_t* Ai, const double* v, double* Delta, double* Lambda,
                          int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < m; ++i) {
        double vi[2] = { v[Ai[2 * i]], v[Ai[2 * i + 1]] };
        oracle_geomean_find_arb(R + 2 * i, w + 2 * i, gamma[i], vi, Delta + 2 * i, Lambda + 2 * i);
    }
}

void oracle_sweep_univ3(int64_t m, const double* current_price, const int64_t* current_tick,
                        const double* gamma, const int32_t* Ai, const int64_t* tick_off,
                        const double* lower_ticks, const double* liquidity, const double* v,
                        double* Delta, double* Lambda, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < m; ++i) {
        double vi[2] = { v[Ai[2 * i]], v[Ai[2 * i + 1]] };
        int64_t o = tick_off[i];
        oracle_univ3_find_arb(current_price[i], current_tick[i], lower_ticks + o, liquidity + o,
                              tick_off[i + 1] - o, gamma[i], vi, Delta + 2 * i, Lambda + 2 * i);
    }
}

/* ------------------------------------------------------------------------- */
/* Serial reductions                                                         */
/* ------------------------------------------------------------------------- */

/* src/router.jl:79-83.  dot() of a Vector with an index-vector view is LinearAlgebra's
 * generic loop: s = x1*y1; s += x2*y2 (no FMA). */
double oracle_dual_acc(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                       const double* v)
{
    double acc = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        double v1 = v[Ai[2 * i]], v2 = v[Ai[2 * i + 1]];
        double dl = Lambda[2 * i] * v1 + Lambda[2 * i + 1] * v2;
        double dd = Delta[2 * i] * v1 + Delta[2 * i + 1] * v2;
        acc += dl - dd;
    }
    return acc;
}

/* src/router.jl:98-100 */
void oracle_grad_scatter(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                         double* G)
{
    for (int64_t i = 0; i < m; ++i) {
        G[Ai[2 * i]] += Lambda[2 * i] - Delta[2 * i];
        G[Ai[2 * i + 1]] += Lambda[2 * i + 1] - Delta[2 * i + 1];
    }
}

/* src/router.jl:111-119 */
void oracle_netflows(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                     int64_t n_tokens, double* psi)
{
    for (int64_t j = 0; j < n_tokens; ++j) psi[j] = 0.0;
    oracle_grad_scatter(m, Delta, Lambda, Ai, psi);
}

/* ------------------------------------------------------------------------- */
/* Objectives: src/objectives.jl                                             */
/* ------------------------------------------------------------------------- */

static int all_c_le_v(const double* c, const double* v, int64_t n)
{
    for (int64_t j = 0; j < n; ++j)
        if (!(c[j] <= v[j])) return 0;
    return 1;
}

double oracle_linear_nonneg_f(const double* c, const double* v, int64_t n) /* :62-67 */
{
    return all_c_le_v(c, v, n) ? 0.0 : INFINITY;
}
void oracle_linear_nonneg_grad(double* g, const double* c, const double* v, int64_t n) /* :69-76 */
{
    double fill = all_c_le_v(c, v, n) ? 0.0 : INFINITY;
    for (int64_t j = 0; j < n; ++j) g[j] = fill;
}
void oracle_linear_nonneg_lower(double* lo, const double* c, int64_t n) /* :78 */
{
    for (int64_t j = 0; j < n; ++j) lo[j] = c[j] + 1e-8;
}

double oracle_basket_liq_f(int64_t i, const double* Din, const double* v, int64_t n) /* :106-111 */
{
    if (v[i] >= 1.0) {
        double s = 0.0; /* left-to-right, as Base.sum does below its pairwise block size */
        for (int64_t j = 0; j < n; ++j) s += (j == i) ? 0.0 : Din[j] * v[j];
        return s;
    }
    return INFINITY;
}
void oracle_basket_liq_grad(double* g, int64_t i, const double* Din, const double* v, int64_t n) /* :113-121 */
{
    if (v[i] >= 1.0) {
        for (int64_t j = 0; j < n; ++j) g[j] = Din[j];
        g[i] = 0.0;
    } else {
        for (int64_t j = 0; j < n; ++j) g[j] = INFINITY;
    }
}
void oracle_basket_liq_lower(double* lo, int64_t i, int64_t n) /* :123-128 */
{
    double se = sqrt(2.220446049250313e-16); /* sqrt(eps()) */
    for (int64_t j = 0; j < n; ++j) lo[j] = se;
    lo[i] = 1.0 + se;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
