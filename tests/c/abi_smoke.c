/* Plain-C client of libcfmm_amd.so (include/cfmm_amd.h): what a `ccall` binding does, without any
 * Python or torch in the process.  Built and run by tests/test_c_abi_gpu.py on the MI355X box.
 * README.md:27-38 of the reference: two ProductTwoCoin pools, LinearNonnegative([1,1]). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cfmm_amd.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != CFMM_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cfmm_last_error(ctx));        \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(void)
{
    cfmm_ctx* ctx = NULL;
    if (cfmm_ctx_create(0, 2, &ctx) != CFMM_OK) {
        fprintf(stderr, "cfmm_ctx_create: %s\n", cfmm_last_error(NULL));
        return 2;
    }
    const double R[4] = {1e6, 1e6, 1e3, 2e3};
    const double gamma[2] = {1.0, 1.0};
    const int32_t Ai[4] = {0, 1, 0, 1};
    CHECK(cfmm_pools_add_product(ctx, 2, R, gamma, Ai));

    /* find_arb!(r, v) at v = [2, 1] and the trades of pool 1 (pool 2 sits exactly at that price) */
    const double v[2] = {2.0, 1.0};
    double D[4], L[4], psi[2], acc;
    CHECK(cfmm_find_arb(ctx, v));
    CHECK(cfmm_get_trades(ctx, D, L));
    CHECK(cfmm_netflows(ctx, psi));
    CHECK(cfmm_dual_value(ctx, &acc));
    printf("trades pool1: D=[%.17g, %.17g] L=[%.17g, %.17g]\n", D[0], D[1], L[0], L[1]);
    printf("netflows at v=[2,1]: [%.17g, %.17g] acc=%.17g\n", psi[0], psi[1], acc);

    /* the same evaluation without trade write-back */
    double psi2[2], acc2;
    CHECK(cfmm_eval(ctx, v, psi2, &acc2));
    if (psi2[0] != psi[0] || psi2[1] != psi[1] || acc2 != acc) {
        fprintf(stderr, "cfmm_eval disagrees with cfmm_find_arb\n");
        return 3;
    }

    /* route!(router) entirely inside the library */
    const double c[2] = {1.0, 1.0};
    double vopt[2], psiopt[2];
    cfmm_route_info info;
    CHECK(cfmm_route(ctx, CFMM_OBJ_LINEAR_NONNEGATIVE, c, 0, NULL, 5, 1e1, 1e-5, 15000, 15000, vopt, psiopt, &info));
    printf("route: v=[%.12g, %.12g] psi=[%.12g, %.12g] evaluations=%d status=%d\n", vopt[0], vopt[1], psiopt[0],
           psiopt[1], info.evaluations, info.status);

    /* cfmm_polish: the gradient-only polish from the route!'s v* (not part of the reference); a converged corner solution stays put */
    double vpol[2] = {vopt[0], vopt[1]}, psipol[2];
    cfmm_polish_info pinfo;
    CHECK(cfmm_polish(ctx, CFMM_OBJ_LINEAR_NONNEGATIVE, c, 0, vpol, 8, 0.0, psipol, &pinfo));
    printf("polish: psi=[%.12g, %.12g] residual %.3g -> %.3g in %d iterations, %d sweeps\n", psipol[0], psipol[1],
           pinfo.residual0, pinfo.residual, pinfo.iterations, pinfo.sweeps);
    if (!(fabs(psipol[1] - 171.4) < 0.1) || !(pinfo.residual <= pinfo.residual0 + 1e-9) || pinfo.sweeps < 4) return 6;

    /* error path: the reference's ArgumentError */
    const int32_t bad[2] = {0, 0};
    const double R1[2] = {1.0, 1.0}, g1[1] = {1.0};
    if (cfmm_pools_add_product(ctx, 1, R1, g1, bad) != CFMM_ERR_INVALID_ARG) return 4;
    printf("error message: %s\n", cfmm_last_error(ctx));

    cfmm_ctx_destroy(ctx);
    return fabs(psiopt[1] - 171.4) < 0.1 && fabs(psiopt[0]) < 1e-3 ? 0 : 5;
}
