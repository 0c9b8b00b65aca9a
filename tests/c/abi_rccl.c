/* Plain-C client of the RCCL entry points of libcfmm_amd.so (include/cfmm_amd.h): north_star's "RCCL all-reduce of psi and
 * grad g over xGMI per outer iteration" reached from a non-Python host in three calls.  usage: abi_rccl [world]
 *   world = 1 (default): one process, one rank -- the communicator path end to end on one GPU (run by tests/test_c_abi_gpu.py);
 *   world = N > 1:       forks N processes BEFORE any HIP call, rank r on GPU r; rank 0 draws the 128-byte id and writes it into
 *                        one pipe per peer -- what a launcher (MPI_Bcast, a Julia Distributed remotecall) does with it; every
 *                        rank holds 1/N of the market and must return the psi of the WHOLE market (needs N GPUs: the driver's
 *                        node, not the 1-GPU box; tests/test_c_abi_gpu.py runs it wherever it sees several GPUs).
 * Market: 40 000 ProductTwoCoin pools over 16 tokens from a fixed LCG; checked against the unsharded sweep of the same pools. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "cfmm_amd.h"

#define M 40000
#define N 16

static double R[2 * M], g[M], v[N];
static int32_t Ai[2 * M];

static unsigned long long lcg_state = 88172645463325252ull;
static double u01(void)
{
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(lcg_state >> 11) / 9007199254740992.0;
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != CFMM_OK) {                                                           \
            fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, rc_, cfmm_last_error(ctx)); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static int run_rank(int rank, int world, const unsigned char* id, const double* psi_ref, double acc_ref)
{
    cfmm_ctx* ctx = NULL;
    if (cfmm_ctx_create(world > 1 ? rank : 0, N, &ctx) != CFMM_OK) {
        fprintf(stderr, "rank %d: cfmm_ctx_create: %s\n", rank, cfmm_last_error(NULL));
        return 2;
    }
    const long lo = (long)M * rank / world, hi = (long)M * (rank + 1) / world;     /* contiguous shard (src/router.jl:39's axis) */
    CHECK(cfmm_pools_add_product(ctx, hi - lo, R + 2 * lo, g + lo, Ai + 2 * lo));
    CHECK(cfmm_rccl_init_rank(ctx, id, world, rank));                               /* call 2 of 3 (call 1: the id, below) */
    double psi[N], acc;
    CHECK(cfmm_eval(ctx, v, psi, &acc));                                            /* global psi / acc on every rank */
    double err = fabs(acc - acc_ref) / fmax(fabs(acc_ref), 1.0), scale = 0.0;
    for (int j = 0; j < N; ++j) scale = fmax(scale, fabs(psi_ref[j]));
    for (int j = 0; j < N; ++j) err = fmax(err, fabs(psi[j] - psi_ref[j]) / scale);
    printf("rank %d/%d: pools [%ld, %ld)  psi[0]=%.17g acc=%.17g  rel err vs unsharded %.3e\n", rank, world, lo, hi, psi[0], acc, err);
    /* route! on the sharded context: every rank runs the same L-BFGS-B on identical {psi, acc} */
    double c[N], vout[N], psi_route[N];
    for (int j = 0; j < N; ++j) c[j] = 0.5 + 0.05 * j;
    cfmm_route_info info;
    CHECK(cfmm_route(ctx, CFMM_OBJ_LINEAR_NONNEGATIVE, c, 0, NULL, 5, 1e1, 1e-5, 15000, 15000, vout, psi_route, &info));
    double neg = 0.0;
    for (int j = 0; j < N; ++j) neg = fmin(neg, psi_route[j] / scale);
    printf("rank %d/%d: route! %d evaluations, status %d, min psi/scale %.2e\n", rank, world, info.evaluations, info.status, neg);
    CHECK(cfmm_set_rccl_comm(ctx, NULL));                                           /* exchange off: the shard alone again */
    cfmm_ctx_destroy(ctx);
    /* status 0 / 1: pgtol / factr; 4: L-BFGS-B's "abnormal termination in line search" at a corner optimum -- the way the Fortran
     * code itself ends such runs (tests/golden/route_fortran.npz: config 2, config 4's shard); feasibility is the check */
    return (err <= 1e-12 && neg >= -1e-6 && (info.status <= 1 || info.status == 4)) ? 0 : 3;
}

/* the whole market on one context, no exchange: what every rank of the sharded run must reproduce */
static int reference(int device, double* psi_ref, double* acc_ref)
{
    cfmm_ctx* ctx = NULL;
    int rank = device;
    if (cfmm_ctx_create(device, N, &ctx) != CFMM_OK) {
        fprintf(stderr, "cfmm_ctx_create: %s\n", cfmm_last_error(NULL));
        return 2;
    }
    CHECK(cfmm_pools_add_product(ctx, M, R, g, Ai));
    CHECK(cfmm_eval(ctx, v, psi_ref, acc_ref));
    cfmm_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    for (long i = 0; i < M; ++i) {
        R[2 * i] = 1000.0 * (0.05 + u01());
        R[2 * i + 1] = 1000.0 * (0.05 + u01());
        g[i] = u01() < 0.5 ? 0.997 : 1.0;
        const int a = (int)(u01() * N) % N, b = (a + 1 + (int)(u01() * (N - 1)) % (N - 1)) % N;
        Ai[2 * i] = a;
        Ai[2 * i + 1] = b;
    }
    for (int j = 0; j < N; ++j) v[j] = exp(0.3 * sin(1.7 * j));
    if (world <= 1) {
        double psi_ref[N], acc_ref;
        unsigned char id[CFMM_RCCL_ID_BYTES];
        cfmm_ctx* ctx = NULL;
        int rank = 0;
        if (reference(0, psi_ref, &acc_ref) != 0) return 2;
        CHECK(cfmm_rccl_unique_id(id));                                             /* call 1 of 3: rank 0 */
        return run_rank(0, 1, id, psi_ref, acc_ref);
    }
    /* N ranks: fork FIRST (a HIP runtime does not survive fork), one pipe per rank > 0 for the 128-byte id -- the
     * "broadcast" a real launcher does with MPI_Bcast / a Distributed.jl remotecall */
    int pipes[64][2];
    pid_t kids[64];
    if (world > 64) return 2;
    for (int r = 1; r < world; ++r)
        if (pipe(pipes[r]) != 0) return 2;
    for (int r = 0; r < world; ++r) {
        kids[r] = fork();
        if (kids[r] == 0) {
            double psi_ref[N], acc_ref;
            unsigned char id[CFMM_RCCL_ID_BYTES];
            cfmm_ctx* ctx = NULL;
            int rank = r;
            if (reference(r, psi_ref, &acc_ref) != 0) _exit(2);                     /* the whole market on this rank's GPU */
            if (r == 0) {
                if (cfmm_rccl_unique_id(id) != CFMM_OK) {                           /* call 1 of 3: rank 0 */
                    fprintf(stderr, "rank 0: cfmm_rccl_unique_id: %s\n", cfmm_last_error(ctx));
                    _exit(1);
                }
                for (int q = 1; q < world; ++q)
                    if (write(pipes[q][1], id, sizeof id) != (ssize_t)sizeof id) _exit(1);
            } else if (read(pipes[r][0], id, sizeof id) != (ssize_t)sizeof id) {
                _exit(1);
            }
            (void)rank;
            _exit(run_rank(r, world, id, psi_ref, acc_ref));
        }
    }
    int rc = 0;
    for (int r = 0; r < world; ++r) {
        int st = 0;
        waitpid(kids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 4;
    }
    printf("%s\n", rc == 0 ? "RCCL_ABI_OK" : "RCCL_ABI_FAILED");
    return rc;
}
