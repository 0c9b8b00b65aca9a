/* Plain-C client of the single-process multi-device context (cfmm_ctx_create_multi): what a Julia
 * `ccall` binding does to shard one Router over the GPUs of a node -- no Python, no torch, no IPC.
 * usage: abi_multi N id0 id1 ...   (a device id may repeat: several shards on one GPU)
 * Market: m ProductTwoCoin pools from a small LCG; checks sharded == unsharded (trades bit for bit,
 * psi to 1e-13 of max|psi|), route! on the sharded context, and the error roll-back. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cfmm_amd.h"

static unsigned long long lcg = 88172645463325252ULL;
static double u01(void)
{
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
}

#define CHECK(ctx, call)                                                                \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != CFMM_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cfmm_last_error(ctx));        \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char** argv)
{
    int32_t ids[64];
    int nd = argc > 1 ? atoi(argv[1]) : 2;
    if (nd < 1 || nd > 64) return 9;
    for (int d = 0; d < nd; ++d) ids[d] = argc > 2 + d ? atoi(argv[2 + d]) : 0;
    const int n = 64;
    const long m = 100003;
    double* R = malloc(sizeof(double) * 2 * m);
    double* g = malloc(sizeof(double) * m);
    int32_t* Ai = malloc(sizeof(int32_t) * 2 * m);
    for (long i = 0; i < m; ++i) {
        R[2 * i] = 1.0 + 1000.0 * u01();
        R[2 * i + 1] = 1.0 + 1000.0 * u01();
        g[i] = u01() < 0.5 ? 0.997 : 1.0;
        int a = (int)(u01() * n), b = (int)(u01() * (n - 1));
        if (a >= n) a = n - 1;
        if (b >= n - 1) b = n - 2;
        if (b >= a) ++b;
        Ai[2 * i] = a;
        Ai[2 * i + 1] = b;
    }
    double v[64], c[64];
    for (int j = 0; j < n; ++j) { v[j] = exp(0.2 * (2 * u01() - 1)); c[j] = 0.05 + u01(); }

    cfmm_ctx *one = NULL, *multi = NULL;
    if (cfmm_ctx_create(ids[0], n, &one) != CFMM_OK) { fprintf(stderr, "create: %s\n", cfmm_last_error(NULL)); return 2; }
    if (cfmm_ctx_create_multi(nd, ids, n, &multi) != CFMM_OK) { fprintf(stderr, "create_multi: %s\n", cfmm_last_error(NULL)); return 2; }
    if (cfmm_device_count(multi) != nd || cfmm_device_count(one) != 1) return 3;
    CHECK(one, cfmm_pools_add_product(one, m, R, g, Ai));
    CHECK(multi, cfmm_pools_add_product(multi, m, R, g, Ai));
    if (cfmm_pools_count(multi) != m || cfmm_segment_count(multi) != 1) return 3;

    /* error roll-back: a bad pool in the LAST shard's block must leave no shard changed */
    int32_t keep = Ai[2 * (m - 1) + 1];
    Ai[2 * (m - 1) + 1] = Ai[2 * (m - 1)];
    if (cfmm_pools_add_product(multi, m, R, g, Ai) != CFMM_ERR_INVALID_ARG) return 4;
    printf("rejected: %s\n", cfmm_last_error(multi));
    Ai[2 * (m - 1) + 1] = keep;
    if (cfmm_pools_count(multi) != m) return 4;

    double *D1 = malloc(16 * m), *L1 = malloc(16 * m), *D2 = malloc(16 * m), *L2 = malloc(16 * m);
    double p1[64], p2[64], a1, a2;
    CHECK(one, cfmm_find_arb(one, v));
    CHECK(multi, cfmm_find_arb(multi, v));
    CHECK(one, cfmm_get_trades(one, D1, L1));
    CHECK(multi, cfmm_get_trades(multi, D2, L2));
    for (long i = 0; i < 2 * m; ++i)
        if (D1[i] != D2[i] || L1[i] != L2[i]) { fprintf(stderr, "trade %ld differs\n", i); return 5; }
    CHECK(one, cfmm_netflows(one, p1));
    CHECK(multi, cfmm_netflows(multi, p2));
    CHECK(one, cfmm_dual_value(one, &a1));
    CHECK(multi, cfmm_dual_value(multi, &a2));
    double scale = 0, err = 0;
    for (int j = 0; j < n; ++j) { if (fabs(p1[j]) > scale) scale = fabs(p1[j]); if (fabs(p1[j] - p2[j]) > err) err = fabs(p1[j] - p2[j]); }
    printf("sharded vs unsharded: max|dpsi|/max|psi| = %.3g, dacc = %.3g\n", err / scale, fabs(a1 - a2) / fabs(a1));
    if (err > 1e-13 * scale || fabs(a1 - a2) > 1e-12 * fabs(a1)) return 6;

    /* a window of trades that straddles shard boundaries */
    double Dw[2 * 5000], Lw[2 * 5000];
    const long first = nd > 1 ? m / nd - 2500 : 1000;   /* nd > 1: straddles the first shard boundary */
    CHECK(multi, cfmm_get_trades_range(multi, 0, first, 5000, Dw, Lw));
    for (long i = 0; i < 2 * 5000; ++i)
        if (Dw[i] != D1[2 * first + i] || Lw[i] != L1[2 * first + i]) return 7;

    /* threads on/off: same bits */
    double p3[64], a3;
    CHECK(multi, cfmm_set_option(multi, "multi_threads", 0));
    CHECK(multi, cfmm_eval(multi, v, p3, &a3));
    for (int j = 0; j < n; ++j) if (p3[j] != p2[j]) return 8;
    if (a3 != a2) return 8;
    CHECK(multi, cfmm_set_option(multi, "multi_threads", 1));

    /* route! with ONE solver over all shards, against the single-device route! */
    double v1[64], v2[64], q1[64], q2[64];
    cfmm_route_info i1, i2;
    for (int j = 0; j < n; ++j) v1[j] = 1.0;
    CHECK(one, cfmm_route(one, CFMM_OBJ_LINEAR_NONNEGATIVE, c, 0, v1, 5, 1e1, 1e-5, 15000, 15000, v1, q1, &i1));
    for (int j = 0; j < n; ++j) v2[j] = 1.0;
    CHECK(multi, cfmm_route(multi, CFMM_OBJ_LINEAR_NONNEGATIVE, c, 0, v2, 5, 1e1, 1e-5, 15000, 15000, v2, q2, &i2));
    scale = err = 0;
    for (int j = 0; j < n; ++j) { if (fabs(q1[j]) > scale) scale = fabs(q1[j]); if (fabs(q1[j] - q2[j]) > err) err = fabs(q1[j] - q2[j]); }
    printf("route: evaluations %d / %d, max|dpsi|/max|psi| = %.3g\n", i1.evaluations, i2.evaluations, err / scale);
    if (err > 1e-6 * scale) return 10;

    /* device-pointer entry points are refused on a multi-device context */
    if (cfmm_sweep_dev(multi, v, p3, 0) != CFMM_ERR_UNSUPPORTED) return 11;
    cfmm_ctx_destroy(multi);
    cfmm_ctx_destroy(one);
    printf("ok nd=%d\n", nd);
    return 0;
}
