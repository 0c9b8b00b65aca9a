/*
 * cfmm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the arithmetic on CFMMRouter.jl's per-CFMM arbitrage
 * path (find_arb! sweep + the dual-value / gradient / netflow reductions that
 * route!'s L-BFGS-B callbacks evaluate).  Every function cites the reference
 * file:line (relative to the reference repo root) that it follows, keeps the
 * reference's floating-point OPERATION ORDER, and sums in serial pool-index
 * order exactly as the reference does.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The shipped path (cfmmrouter.jl_amd/) never does.
 *
 * Pinning status: the reference is Julia and Julia is not installed in the
 * build container, so the reference itself cannot be executed.  The oracle is
 * pinned on every known-answer test and predicate the reference's own tests
 * hold for this path (tests/test_oracle_*.py restate test/cfmms.jl,
 * test/objectives.jl, test/arb.jl, test/swap.jl).  The L-BFGS-B outer loop is
 * a third-party dependency (LBFGSB.jl 0.4.x, Fortran L-BFGS-B 3.0) that the
 * reference's tests check for feasibility only: route!-level parity is
 * therefore "parity unpinned" (see DESIGN.md).
 *
 * Conventions: token indices are 0-based int32 here (the reference's Ai is
 * 1-based Int64; the harness subtracts 1).  Pair arrays are [m][2] row-major.
 * UniV3 tick indices are 1-based inside the functions, as in the reference,
 * because the `idx > current_tick` comparisons are written that way.
 */
#ifndef CFMM_ORACLE_H
#define CFMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-pool closed forms (src/cfmms.jl) -------------------------------- */

/* src/cfmms.jl:125-126,130-140 */
void oracle_product_find_arb(const double R[2], double gamma, const double v[2],
                             double Delta[2], double Lambda[2]);
/* src/cfmms.jl:113-116 / :117-122 */
double oracle_product_phi(const double R[2]);
void oracle_product_grad_phi(const double R[2], double out[2]);

/* src/cfmms.jl:180-181,185-196 */
void oracle_geomean_find_arb(const double R[2], const double w[2], double gamma,
                             const double v[2], double Delta[2], double Lambda[2]);
/* src/cfmms.jl:167-171 / :172-178 */
double oracle_geomean_phi(const double R[2], const double w[2]);
void oracle_geomean_grad_phi(const double R[2], const double w[2], double out[2]);

/* UniV3 / BoundedProduct: src/cfmms.jl:226-449 */
typedef struct {
    double k, alpha, beta, R_1, R_2; /* src/cfmms.jl:272-278 */
} oracle_bounded_product;

/* src/cfmms.jl:235 searchsortedlast(lower_ticks, current_price, rev=true); 1-based, 0 if none */
int64_t oracle_univ3_current_tick(const double* lower_ticks, int64_t n_ticks, double current_price);
/* src/cfmms.jl:294-313 (idx is 1-based) */
oracle_bounded_product oracle_univ3_compute_at_tick(double current_price, int64_t current_tick,
                                                    const double* lower_ticks, const double* liquidity,
                                                    int64_t n_ticks, int64_t idx);
/* src/cfmms.jl:289 */
oracle_bounded_product oracle_flip_sides(oracle_bounded_product t);
/* src/cfmms.jl:321-337 */
void oracle_find_arb_pos(oracle_bounded_product t, double price, double* delta, double* lambda);
/* src/cfmms.jl:339-395 */
void oracle_univ3_find_arb(double current_price, int64_t current_tick, const double* lower_ticks,
                           const double* liquidity, int64_t n_ticks, double gamma, const double v[2],
                           double Delta[2], double Lambda[2]);
/* src/cfmms.jl:436-449 (with helpers :401-434) -- test helper the reference's UniV3 tests use */
double oracle_univ3_forward_trade(const double Delta[2], double current_price, int64_t current_tick,
                                  const double* lower_ticks, const double* liquidity, int64_t n_ticks,
                                  double gamma);

/* ---- router-level sweep (src/router.jl:38-42), one call per pool family --- */
/* nthreads mirrors Threads.@threads; 1 = serial.  Ai is [m][2], 0-based. */
void oracle_sweep_product(int64_t m, const double* R, const double* gamma, const int32_t* Ai,
                          const double* v, double* Delta, double* Lambda, int nthreads);
void oracle_sweep_geomean(int64_t m, const double* R, const double* w, const double* gamma,
                          const int32_t* Ai, const double* v, double* Delta, double* Lambda,
                          int nthreads);
/* tick_off has m+1 entries (CSR); current_tick is 1-based per pool */
void oracle_sweep_univ3(int64_t m, const double* current_price, const int64_t* current_tick,
                        const double* gamma, const int32_t* Ai, const int64_t* tick_off,
                        const double* lower_ticks, const double* liquidity, const double* v,
                        double* Delta, double* Lambda, int nthreads);

/* ---- serial reductions route!'s callbacks perform ------------------------ */
/* src/router.jl:79-83 : acc += dot(L, v[Ai]) - dot(D, v[Ai]), pool order */
double oracle_dual_acc(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                       const double* v);
/* src/router.jl:98-100 : G[Ai] .+= L .- D, pool order (G is NOT zeroed here) */
void oracle_grad_scatter(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                         double* G);
/* src/router.jl:111-119 : psi = 0; psi[Ai] += L - D */
void oracle_netflows(int64_t m, const double* Delta, const double* Lambda, const int32_t* Ai,
                     int64_t n_tokens, double* psi);

/* ---- objectives (src/objectives.jl), i is 0-based here ------------------- */
double oracle_linear_nonneg_f(const double* c, const double* v, int64_t n);          /* :62-67 */
void oracle_linear_nonneg_grad(double* g, const double* c, const double* v, int64_t n); /* :69-76 */
void oracle_linear_nonneg_lower(double* lo, const double* c, int64_t n);             /* :78 */
double oracle_basket_liq_f(int64_t i, const double* Din, const double* v, int64_t n);   /* :106-111 */
void oracle_basket_liq_grad(double* g, int64_t i, const double* Din, const double* v, int64_t n); /* :113-121 */
void oracle_basket_liq_lower(double* lo, int64_t i, int64_t n);                      /* :123-128 */

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
