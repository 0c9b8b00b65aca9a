// solver_bench.cpp -- what does csrc/lbfgsb.cpp itself cost per evaluation on a route!-shaped problem?  (VERDICT r5 item 5:
// 11-12 us per evaluation at n = 512 on the config-4 shard, 40 % of that route!'s wall clock.)  Host only: the dual of a random
// ProductTwoCoin arbitrage market (LinearNonnegative, the reference's call shape nbd = 2 / u = Inf / m = 5 / factr = 1e1 /
// pgtol = 1e-5) is evaluated by a plain C loop inside the callback, whose time is measured and subtracted.
// build: g++ -O2 -std=c++17 -I include -o scripts/native/solver_bench.bin scripts/native/solver_bench.cpp -L cfmmrouter.jl_amd -lcfmm_amd -Wl,-rpath,$PWD/cfmmrouter.jl_amd -Wl,-rpath,/opt/rocm/lib
#include "cfmm_amd.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Market {
    int n;
    std::vector<double> R1, R2, g, c;
    std::vector<int> i1, i2;
    double t_cb = 0.0;
    long calls = 0;
};
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double fg(void* user, const double* v, double* G)
{
    Market& M = *static_cast<Market*>(user);
    const double t0 = now();
    double f = 0.0;
    for (int j = 0; j < M.n; ++j) { G[j] = 0.0; if (v[j] < M.c[j]) f = INFINITY; }      // LinearNonnegative: f = 0 on v >= c
    for (size_t k = 0; k < M.R1.size(); ++k) {
        const double R1 = M.R1[k], R2 = M.R2[k], g = M.g[k], v1 = v[M.i1[k]], v2 = v[M.i2[k]], kk = R1 * R2;
        const double d1 = std::fmax(std::sqrt(g * (v2 / v1) * kk) - R1, 0.0) / g, d2 = std::fmax(std::sqrt(g * (v1 / v2) * kk) - R2, 0.0) / g;
        const double l1 = std::fmax(R1 - std::sqrt(kk / (g * (v1 / v2))), 0.0), l2 = std::fmax(R2 - std::sqrt(kk / (g * (v2 / v1))), 0.0);
        f += (l1 * v1 + l2 * v2) - (d1 * v1 + d2 * v2);
        G[M.i1[k]] += l1 - d1;
        G[M.i2[k]] += l2 - d2;
    }
    M.t_cb += now() - t0;
    ++M.calls;
    return f;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, m = argc > 2 ? atoi(argv[2]) : 20000, reps = argc > 3 ? atoi(argv[3]) : 20;
    Market M;
    M.n = n;
    srand(1234);
    auto u = [] { return (rand() + 0.5) / (RAND_MAX + 1.0); };
    for (int k = 0; k < m; ++k) {
        M.R1.push_back(1000 * u()); M.R2.push_back(1000 * u()); M.g.push_back(u() < 0.5 ? 0.997 : 1.0);
        int a = (int)(n * u()), b = (int)(n * u());
        if (b == a) b = (a + 1) % n;
        M.i1.push_back(a); M.i2.push_back(b);
    }
    for (int j = 0; j < n; ++j) M.c.push_back(u());
    std::vector<double> lo(n), hi(n, INFINITY), x(n);
    std::vector<int32_t> nbd(n, 2);
    for (int j = 0; j < n; ++j) lo[j] = M.c[j] + 1e-8;
    double best = 1e30;
    cfmm_route_info info;
    for (int r = 0; r < reps; ++r) {
        for (int j = 0; j < n; ++j) x[j] = 1.0;
        M.t_cb = 0.0; M.calls = 0;
        const double t0 = now();
        cfmm_lbfgsb_minimize(n, x.data(), lo.data(), hi.data(), nbd.data(), fg, &M, 5, 1e1, 1e-5, 15000, 15000, 1, &info);
        const double solver = (now() - t0 - M.t_cb) / M.calls;
        if (solver < best) best = solver;
    }
    int at_bound = 0;
    for (int j = 0; j < n; ++j) at_bound += x[j] == lo[j];
    printf("n %d pools %d: %d iterations, %ld evaluations, status %d, %d prices on their bound; solver %.2f us per evaluation (best of %d)\n", n, m,
           info.iterations, M.calls, info.status, at_bound, best * 1e6, reps);
    return 0;
}
