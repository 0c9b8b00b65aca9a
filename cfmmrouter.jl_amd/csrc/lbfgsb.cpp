// lbfgsb.cpp -- see lbfgsb.h.  Host-only; no device code.
#include "lbfgsb.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace cfmm {
namespace {

constexpr double kEps = std::numeric_limits<double>::epsilon();
constexpr double kInf = std::numeric_limits<double>::infinity();

inline bool has_lower(int nbd) { return nbd == 1 || nbd == 2; }
inline bool has_upper(int nbd) { return nbd == 2 || nbd == 3; }

// Dense k×k solve A·X = B (B has nrhs columns, row-major), Gaussian elimination with partial
// pivoting.  k = 2·col <= 2·m is tiny.
bool solve_dense(std::vector<double> A, std::vector<double>& B, int k, int nrhs)
{
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int r = c + 1; r < k; ++r)
            if (std::fabs(A[r * k + c]) > std::fabs(A[piv * k + c])) piv = r;
        if (A[piv * k + c] == 0.0) return false;
        if (piv != c) {
            for (int j = 0; j < k; ++j) std::swap(A[c * k + j], A[piv * k + j]);
            for (int j = 0; j < nrhs; ++j) std::swap(B[c * nrhs + j], B[piv * nrhs + j]);
        }
        const double inv = 1.0 / A[c * k + c];
        for (int r = 0; r < k; ++r) {
            if (r == c) continue;
            const double f = A[r * k + c] * inv;
            if (f == 0.0) continue;
            for (int j = c; j < k; ++j) A[r * k + j] -= f * A[c * k + j];
            for (int j = 0; j < nrhs; ++j) B[r * nrhs + j] -= f * B[c * nrhs + j];
        }
    }
    for (int c = 0; c < k; ++c) {
        const double inv = 1.0 / A[c * k + c];
        for (int j = 0; j < nrhs; ++j) B[c * nrhs + j] *= inv;
    }
    return true;
}

// ---- Moré–Thuente line search (the dcsrch/dcstep pair of MINPACK-2, restated) ----------------
struct MoreThuente {
    double ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0.0, stpmax = 0.0;
    bool brackt = false;
    int stage = 1;
    double finit = 0, ginit = 0, gtest = 0, width = 0, width1 = 0;
    double stx = 0, fx = 0, gx = 0, sty = 0, fy = 0, gy = 0, stmin = 0, stmax = 0;
    enum Task { kEvaluate, kConverged, kWarning, kError };

    Task start(double f, double g, double stp)
    {
        if (stp < stpmin || stp > stpmax || g >= 0.0) return kError;
        brackt = false;
        stage = 1;
        finit = f;
        ginit = g;
        gtest = ftol * ginit;
        width = stpmax - stpmin;
        width1 = 2.0 * width;
        stx = 0.0; fx = finit; gx = ginit;
        sty = 0.0; fy = finit; gy = ginit;
        stmin = 0.0;
        stmax = stp + 4.0 * stp;
        return kEvaluate;
    }

    static void step(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                     double fp, double dp, bool& brackt, double stpmin, double stpmax)
    {
        const double sgnd = dp * (dx / std::fabs(dx));
        double stpf, stpc, stpq;
        if (fp > fx) { // case 1: higher function value -- the minimum is bracketed
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
            if (stp < stx) gamma = -gamma;
            const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
            stpc = stx + r * (stp - stx);
            stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
            stpf = std::fabs(stpc - stx) < std::fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
            brackt = true;
        } else if (sgnd < 0.0) { // case 2: derivatives of opposite sign
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
            stpc = stp + r * (stx - stp);
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
            brackt = true;
        } else if (std::fabs(dp) < std::fabs(dx)) { // case 3: derivative magnitude decreases
            const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
            const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
            double gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
            if (stp > stx) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
            if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
            else if (stp > stx) stpc = stpmax;
            else stpc = stpmin;
            stpq = stp + (dp / (dp - dx)) * (stx - stp);
            if (brackt) {
                stpf = std::fabs(stpc - stp) < std::fabs(stpq - stp) ? stpc : stpq;
                if (stp > stx) stpf = std::min(stp + 0.66 * (sty - stp), stpf);
                else stpf = std::max(stp + 0.66 * (sty - stp), stpf);
            } else {
                stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
                stpf = std::min(stpmax, stpf);
                stpf = std::max(stpmin, stpf);
            }
        } else { // case 4: derivative does not decrease
            if (brackt) {
                const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
                const double s = std::max({std::fabs(theta), std::fabs(dy), std::fabs(dp)});
                double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
                if (stp > sty) gamma = -gamma;
                const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
                stpc = stp + r * (sty - stp);
                stpf = stpc;
            } else if (stp > stx) stpf = stpmax;
            else stpf = stpmin;
        }
        if (fp > fx) {
            sty = stp; fy = fp; dy = dp;
        } else {
            if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
            stx = stp; fx = fp; dx = dp;
        }
        stp = stpf;
    }

    Task next(double f, double g, double& stp)
    {
        const double ftest = finit + stp * gtest;
        if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
        if (f <= ftest && std::fabs(g) <= gtol * (-ginit)) return kConverged;
        if (brackt && (stp <= stmin || stp >= stmax)) return kWarning;
        if (brackt && stmax - stmin <= xtol * stmax) return kWarning;
        if (stp == stpmax && f <= ftest && g <= gtest) return kWarning;
        if (stp == stpmin && (f > ftest || g >= gtest)) return kWarning;
        if (stage == 1 && f <= fx && f > ftest) {
            double fm = f - stp * gtest, fxm = fx - stx * gtest, fym = fy - sty * gtest;
            double gm = g - gtest, gxm = gx - gtest, gym = gy - gtest;
            step(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax);
            fx = fxm + stx * gtest;
            fy = fym + sty * gtest;
            gx = gxm + gtest;
            gy = gym + gtest;
        } else {
            step(stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax);
        }
        if (brackt) {
            if (std::fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
            width1 = width;
            width = std::fabs(sty - stx);
            stmin = std::min(stx, sty);
            stmax = std::max(stx, sty);
        } else {
            stmin = stp + 1.1 * (stp - stx);
            stmax = stp + 4.0 * (stp - stx);
        }
        stp = std::max(stp, stpmin);
        stp = std::min(stp, stpmax);
        if ((brackt && (stp <= stmin || stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) stp = stx;
        return kEvaluate;
    }
};

// ---- vector kernels -----------------------------------------------------------------------------
// n is a few hundred and every iteration makes ~40 passes over length-n vectors: these loops are the
// solver's cost.  Reassociation is allowed inside the dot products only (so that they vectorise);
// nothing here feeds the bit-exact pool arithmetic.
// Sums that feed the solver's DECISIONS at the noise floor of the dual value -- d'd of the Cauchy step, s'y / y'y / s'g of
// the update, d'd / g'd of the line search -- are accumulated in index order like the Fortran's ddot: reassociating them is
// harmless to ~1e-16 per sum, but on interior optima the run then stops at another point INSIDE the rounding noise (measured,
// config 5 against the Fortran fixture: 9.1e-8 with these sums in order, 6e-7 … 1.6e-6 with any one of them vectorised;
// round 6).  The W'x products and the Gram matrix (dotn / dot4n below) are reassociated: they only shape the model.
inline double dotn(const double* a, const double* b, int n)
{
#pragma clang fp reassociate(on)
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}
// four dot products against one vector in ONE pass (the masked Gram matrix of the subspace step: the masked column is
// loaded once for four partners)
inline void dot4n(const double* __restrict a, const double* __restrict b0, const double* __restrict b1,
                  const double* __restrict b2, const double* __restrict b3, int n, double out[4])
{
#pragma clang fp reassociate(on)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = 0; i < n; ++i) {
        const double ai = a[i];
        s0 += ai * b0[i];
        s1 += ai * b1[i];
        s2 += ai * b2[i];
        s3 += ai * b3[i];
    }
    out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
}
inline void axpyn(double alpha, const double* x, double* y, int n)
{
    for (int i = 0; i < n; ++i) y[i] += alpha * x[i];
}

// ---- limited-memory matrices --------------------------------------------------------------------
// S and Y are stored by COLUMN (one correction pair = two contiguous length-n vectors) in a circular
// buffer of m slots, so that W'x (dot products) and W·c (axpys) are contiguous, vectorisable passes
// and dropping the oldest pair moves nothing.  Logical column j (oldest first) lives in slot (start+j) % m.
struct Memory {
    int n = 0, m = 0, col = 0, start = 0;
    double theta = 1.0;
    std::vector<double> Sc, Yc;            // [m][n]
    std::vector<double> SY, SS, YY;        // m×m, row-major, logical order: [i][j] = s_i'y_j / s_i's_j / y_i'y_j
    std::vector<double> M;                 // (2col)×(2col) = K^{-1}, K = [[-D, L'],[L, θ S'S]]  (symmetric)
    std::vector<double> K;                 // scratch

    void init(int n_, int m_)
    {
        n = n_;
        m = m_;
        Sc.assign((size_t)n * m, 0.0);
        Yc.assign((size_t)n * m, 0.0);
        SY.assign((size_t)m * m, 0.0);
        SS.assign((size_t)m * m, 0.0);
        YY.assign((size_t)m * m, 0.0);
    }
    const double* Scol(int j) const { return &Sc[(size_t)((start + j) % m) * n]; }
    const double* Ycol(int j) const { return &Yc[(size_t)((start + j) % m) * n]; }

    void reset()
    {
        col = 0;
        start = 0;
        theta = 1.0;
    }

    void push(const std::vector<double>& s, const std::vector<double>& y, double sy, double yy)
    {
        if (col == m) { // drop the oldest pair: advance the ring, shift the small m×m tables
            start = (start + 1) % m;
            for (int i = 0; i + 1 < m; ++i)
                for (int j = 0; j + 1 < m; ++j) {
                    SY[i * m + j] = SY[(i + 1) * m + j + 1];
                    SS[i * m + j] = SS[(i + 1) * m + j + 1];
                    YY[i * m + j] = YY[(i + 1) * m + j + 1];
                }
            --col;
        }
        const int k = col;
        for (int j = 0; j < k; ++j) {
            const double* sj = Scol(j);
            const double* yj = Ycol(j);
            SY[j * m + k] = dotn(sj, y.data(), n);          // s_j'y_new
            SY[k * m + j] = dotn(s.data(), yj, n);          // s_new'y_j
            SS[j * m + k] = SS[k * m + j] = dotn(sj, s.data(), n);
            YY[j * m + k] = YY[k * m + j] = dotn(yj, y.data(), n);
        }
        SY[k * m + k] = sy;
        SS[k * m + k] = dotn(s.data(), s.data(), n);
        YY[k * m + k] = yy;
        std::copy(s.begin(), s.end(), Sc.begin() + (size_t)((start + k) % m) * n);
        std::copy(y.begin(), y.end(), Yc.begin() + (size_t)((start + k) % m) * n);
        ++col;
        theta = yy / sy;
        form_M();
    }

    bool form_M()
    {
        const int k = 2 * col;
        K.assign((size_t)k * k, 0.0);
        for (int i = 0; i < col; ++i) {
            K[i * k + i] = -SY[i * m + i]; // -D
            for (int j = 0; j < col; ++j) {
                if (i > j) { // L_ij = s_i'y_j, i > j
                    K[(col + i) * k + j] = SY[i * m + j];
                    K[j * k + (col + i)] = SY[i * m + j];
                }
                K[(col + i) * k + (col + j)] = theta * SS[i * m + j];
            }
        }
        M.assign((size_t)k * k, 0.0);
        for (int i = 0; i < k; ++i) M[i * k + i] = 1.0;
        return solve_dense(K, M, k, k);
    }

    // row b of W = [Y θS]
    void w_row(int b, double* w) const
    {
        for (int j = 0; j < col; ++j) {
            w[j] = Ycol(j)[b];
            w[col + j] = theta * Scol(j)[b];
        }
    }
    // out = W'x for a length-n vector x (2col entries)
    void Wt_times(const double* x, double* out) const
    {
        for (int j = 0; j < col; ++j) {
            out[j] = dotn(Ycol(j), x, n);
            out[col + j] = theta * dotn(Scol(j), x, n);
        }
    }
    // y += W·c for 2col coefficients c
    void W_times_add(const double* c, double* y) const
    {
        for (int j = 0; j < col; ++j) {
            axpyn(c[j], Ycol(j), y, n);
            axpyn(theta * c[col + j], Scol(j), y, n);
        }
    }
    // W'W = [[Y'Y, θ Y'S], [θ S'Y, θ² S'S]]  (2col × 2col, row-major)
    void WtW(std::vector<double>& out) const
    {
        const int k = 2 * col;
        out.assign((size_t)k * k, 0.0);
        for (int i = 0; i < col; ++i)
            for (int j = 0; j < col; ++j) {
                out[i * k + j] = YY[i * m + j];
                out[i * k + col + j] = theta * SY[j * m + i];          // y_i's_j
                out[(col + i) * k + j] = theta * SY[i * m + j];        // s_i'y_j
                out[(col + i) * k + col + j] = theta * theta * SS[i * m + j];
            }
    }
    void M_times(const std::vector<double>& v, std::vector<double>& out) const
    {
        const int k = 2 * col;
        for (int i = 0; i < k; ++i) {
            double s = 0.0;
            for (int j = 0; j < k; ++j) s += M[i * k + j] * v[j];
            out[i] = s;
        }
    }
};

// max_i |proj g_i|.  hl / hu: 1.0 where the variable has a (finite) lower / upper bound, else 0.0 -- selects instead of
// branches, so the loop vectorises (max is exact: no rounding, any order)
double projected_gradient_norm(int n, const double* x, const double* g, const double* l, const double* u,
                               const double* hl, const double* hu)
{
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
        const double gi = g[i];
        const double up = hu[i] != 0.0 ? std::max(x[i] - u[i], gi) : gi;   // gi < 0
        const double lo = hl[i] != 0.0 ? std::min(x[i] - l[i], gi) : gi;   // gi >= 0
        const double pgi = gi < 0.0 ? up : lo;
        nrm = std::max(nrm, std::fabs(pgi));
    }
    return nrm;
}

} // namespace

LbfgsbResult lbfgsb_minimize(int n, double* x, const double* lower, const double* upper, const int* nbd_in,
                             const LbfgsbFn& fg, const LbfgsbOptions& opt)
{
    LbfgsbResult res;
    const int m = std::max(1, opt.m);
    std::vector<int> nbd(nbd_in, nbd_in + n);
    std::vector<double> l(n), u(n), hlm(n), hum(n);   // hlm / hum: 1.0 where the variable has a finite lower / upper bound
    bool constrained = false, boxed = true, boxed_as_given = true;
    for (int i = 0; i < n; ++i) {
        if (nbd_in[i] != 2) boxed_as_given = false;
        l[i] = lower ? lower[i] : -kInf;
        u[i] = upper ? upper[i] : kInf;
        // infinite bounds are no bounds (the reference passes nbd = 2 with u = Inf, src/router.jl:68-70)
        const bool hl = has_lower(nbd[i]) && std::isfinite(l[i]);
        const bool hu = has_upper(nbd[i]) && std::isfinite(u[i]);
        nbd[i] = hl ? (hu ? 2 : 1) : (hu ? 3 : 0);
        hlm[i] = hl ? 1.0 : 0.0;
        hum[i] = hu ? 1.0 : 0.0;
        if (!hl) l[i] = -kInf;      // (a bound that does not exist never binds: the branch-free loops below rely on it)
        if (!hu) u[i] = kInf;
        if (nbd[i] != 0) constrained = true;
        if (nbd[i] != 2) boxed = false;
        // project the starting point into the box
        if (hl && x[i] < l[i]) x[i] = l[i];
        if (hu && x[i] > u[i]) x[i] = u[i];
    }

    if (opt.boxed_from_nbd) boxed = boxed_as_given;

    Memory mem;
    mem.init(n, m);

    std::vector<double> g(n), g_old(n), x_old(n), xcp(n), z(n), d(n), t(n), s(n), y(n);
    std::vector<double> p, c, wb, v1, v2, Mc;
    std::vector<double> rfull, wvfull, masked, du, wzr, WZ, vv, N;   // subspace-step scratch, reused across iterations
    std::vector<int> order, fixed(n), free_idx;
    order.reserve(n);
    free_idx.reserve(n);

    auto evaluate = [&](double* xx, double* gg) -> double {
        ++res.evaluations;
        return fg(xx, gg);
    };

    double f = evaluate(x, g.data());
    if (!std::isfinite(f)) {
        res.status = 5;
        res.message = "ERROR: non-finite objective at the (projected) starting point";
        res.f = f;
        return res;
    }
    double pg = projected_gradient_norm(n, x, g.data(), l.data(), u.data(), hlm.data(), hum.data());
    if (pg <= opt.pgtol) {
        res.f = f;
        res.proj_grad = pg;
        res.status = 0;
        res.message = "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL";
        return res;
    }

    int iter = 0;
    for (;;) {
        const int col = mem.col;
        const int k2 = 2 * col;
        const double theta = mem.theta;
        p.assign(k2, 0.0);
        c.assign(k2, 0.0);
        wb.assign(k2, 0.0);
        v1.assign(k2, 0.0);
        v2.assign(k2, 0.0);
        Mc.assign(k2, 0.0);

        // ---------------- generalized Cauchy point (BLNZ95, Algorithm CP) -------------------------
        if (!constrained && col > 0) {
            for (int i = 0; i < n; ++i) { xcp[i] = x[i]; fixed[i] = 0; }
        } else {
            order.clear();
            double dtd = 0.0;
            int moving = 0;
            // breakpoints, branch-free (vectorised): t_i = (x_i − u_i)/g_i for g_i < 0, (x_i − l_i)/g_i for g_i > 0, +inf where
            // that bound does not exist (l = −inf / u = +inf there: the quotient itself is +inf) or g_i = 0
            {
                const double* __restrict gp = g.data();
                const double* __restrict lp = l.data();
                const double* __restrict up = u.data();
                const double* __restrict xp = x;
                double* __restrict tp = t.data();
                double* __restrict cp = xcp.data();
                for (int i = 0; i < n; ++i) {
                    const double gi = gp[i], xi = xp[i];
                    const double num = gi < 0.0 ? xi - up[i] : xi - lp[i];
                    const double ti = num / gi;                       // g = 0: ±inf or NaN, replaced below
                    tp[i] = gi != 0.0 ? ti : kInf;
                    cp[i] = xi;
                }
                // ... then the (order-dependent) bookkeeping: the sum d'd in index order, the heap's candidates
                double* __restrict dp = d.data();
                int* __restrict fp_ = fixed.data();
                order.resize(n);
                int* __restrict op = order.data();
                int cand = 0;
                for (int i = 0; i < n; ++i) {
                    const double ti = tp[i];
                    const bool fx = ti <= 0.0;                        // at its bound with the gradient pushing outward
                    const double di = fx ? 0.0 : -gp[i];
                    fp_[i] = fx ? 1 : 0;
                    dp[i] = di;
                    dtd += di * di;                                   // (in index order: see the note above dotn)
                    moving += di != 0.0 ? 1 : 0;
                }
                for (int i = 0; i < n; ++i) {                         // the heap's candidates, in index order
                    op[cand] = i;
                    cand += (tp[i] > 0.0 && tp[i] < kInf) ? 1 : 0;
                }
                order.resize(cand);
            }
            // breakpoints are consumed in increasing order, usually only the first few: a min-heap
            // (the Fortran's hpsolb) instead of a full sort
            auto later = [&](int a, int b) { return t[a] > t[b] || (t[a] == t[b] && a > b); };
            mem.Wt_times(d.data(), p.data());   // p = W'd  (d is zero on the variables that do not move)
            double fp = -dtd;
            double fpp = -theta * fp;
            if (col > 0) {
                mem.M_times(p, v1);
                for (int j = 0; j < k2; ++j) fpp -= p[j] * v1[j];
            }
            const double fpp_org = fpp;
            double dt_min = fpp > 0.0 ? -fp / fpp : 0.0;
            double t_old = 0.0;
            // The loop below stops at the first breakpoint beyond the minimiser of the first segment: when even the EARLIEST
            // breakpoint lies beyond it (the usual case once the active set has settled) no heap is needed at all
            {
                double t_first = kInf;
                for (int q : order) t_first = std::min(t_first, t[q]);
                if (dt_min < t_first) order.clear();
                else std::make_heap(order.begin(), order.end(), later);
            }
            size_t heap_end = order.size();
            bool all_fixed = moving == 0;
            while (!all_fixed && heap_end > 0) {
                const int b = order.front();
                const double tb = t[b];
                const double dt = tb - t_old;
                if (dt_min < dt) break;
                // variable b reaches its bound
                std::pop_heap(order.begin(), order.begin() + heap_end, later);
                --heap_end;
                const double gb = g[b];
                xcp[b] = d[b] > 0.0 ? u[b] : l[b];
                const double zb = xcp[b] - x[b];
                d[b] = 0.0;
                fixed[b] = 1;
                for (int j = 0; j < k2; ++j) c[j] += dt * p[j];
                fp += dt * fpp + gb * gb + theta * gb * zb;
                fpp -= theta * gb * gb;
                if (col > 0) {
                    // M is symmetric (the inverse of the symmetric middle matrix): one product M·w_b serves
                    // w_b'Mc, w_b'Mp and w_b'Mw_b -- this loop runs once per variable that reaches its bound
                    // (hundreds per iteration on arbitrage problems, where most prices end on their bound)
                    mem.w_row(b, wb.data());
                    mem.M_times(wb, v1);
                    double wMc = 0, wMp = 0, wMw = 0;
                    for (int j = 0; j < k2; ++j) { wMc += v1[j] * c[j]; wMp += v1[j] * p[j]; wMw += v1[j] * wb[j]; }
                    fp -= gb * wMc;
                    fpp -= 2.0 * gb * wMp + gb * gb * wMw;
                    for (int j = 0; j < k2; ++j) p[j] += gb * wb[j];
                }
                fpp = std::max(kEps * fpp_org, fpp);
                dt_min = -fp / fpp;
                t_old = tb;
                all_fixed = --moving == 0;
            }
            if (!all_fixed) {
                dt_min = std::max(dt_min, 0.0);
                t_old += dt_min;
                for (int i = 0; i < n; ++i)
                    if (d[i] != 0.0) xcp[i] = x[i] + t_old * d[i];
                for (int j = 0; j < k2; ++j) c[j] += dt_min * p[j];
            }
        }

        // ---------------- subspace minimization (BLNZ95 §5.1 direct primal + MN11 projection) -----
        for (int i = 0; i < n; ++i) z[i] = xcp[i];
        free_idx.clear();
        for (int i = 0; i < n; ++i)
            if (!fixed[i]) free_idx.push_back(i);
        if (col > 0 && !free_idx.empty()) {
            const int nf = (int)free_idx.size();
            // r = Z'(g + θ(xcp − x) − W M c), kept as a full-length vector that is zero on the fixed variables
            rfull.assign(n, 0.0);
            mem.M_times(c, Mc);
            mem.W_times_add(Mc.data(), rfull.data());                 // W M c
            for (int i = 0; i < n; ++i)
                rfull[i] = fixed[i] ? 0.0 : g[i] + theta * (xcp[i] - x[i]) - rfull[i];
            wzr.assign(k2, 0.0);
            mem.Wt_times(rfull.data(), wzr.data());                   // W'Z r
            // W'Z Z'W: a sum of k2×k2 outer products over the FREE variables -- or, when fewer variables are
            // fixed than free (the usual case away from a corner of the box), W'W from the stored inner
            // products minus the outer products of the FIXED rows: O(min(free, fixed)·(2m)²) instead of O(n·(2m)²)
            // W'ZZ'W.  Few fixed (or few free) variables: W'W from the stored inner products minus (or just) the
            // outer products of those rows, O(min(free, fixed)·(2m)²) with strided row gathers.  Otherwise: the
            // masked Gram matrix as (2m)(2m+1)/2 contiguous dot products of length n (vectorised), O(n·(2m)²/2).
            const int nfix = n - nf;
            if (std::min(nf, nfix) <= 24) {
                if (nfix < nf) {
                    mem.WtW(WZ);
                    for (int i = 0; i < n; ++i) {
                        if (!fixed[i]) continue;
                        mem.w_row(i, wb.data());
                        for (int j = 0; j < k2; ++j) {
                            const double wj = wb[j];
                            for (int q = 0; q < k2; ++q) WZ[j * k2 + q] -= wj * wb[q];
                        }
                    }
                } else {
                    WZ.assign((size_t)k2 * k2, 0.0);
                    for (int a = 0; a < nf; ++a) {
                        mem.w_row(free_idx[a], wb.data());
                        for (int j = 0; j < k2; ++j) {
                            const double wj = wb[j];
                            for (int q = 0; q < k2; ++q) WZ[j * k2 + q] += wj * wb[q];
                        }
                    }
                }
            } else {
                WZ.assign((size_t)k2 * k2, 0.0);
                masked.resize(n);
                for (int a = 0; a < k2; ++a) {
                    const double* wa = a < col ? mem.Ycol(a) : mem.Scol(a - col);
                    const double sa = a < col ? 1.0 : theta;
                    for (int i = 0; i < n; ++i) masked[i] = fixed[i] ? 0.0 : sa * wa[i];
                    auto colptr = [&](int b) { return b < col ? mem.Ycol(b) : mem.Scol(b - col); };
                    int b = a;
                    for (; b + 3 < k2; b += 4) {          // four partners per pass over the masked column
                        double g4[4];
                        dot4n(masked.data(), colptr(b), colptr(b + 1), colptr(b + 2), colptr(b + 3), n, g4);
                        for (int q = 0; q < 4; ++q) {
                            const double g_ab = (b + q < col ? 1.0 : theta) * g4[q];
                            WZ[a * k2 + b + q] = g_ab;
                            WZ[(b + q) * k2 + a] = g_ab;
                        }
                    }
                    for (; b < k2; ++b) {
                        const double g_ab = (b < col ? 1.0 : theta) * dotn(masked.data(), colptr(b), n);
                        WZ[a * k2 + b] = g_ab;
                        WZ[b * k2 + a] = g_ab;
                    }
                }
            }
            std::vector<double>& v = vv;
            v.assign(k2, 0.0);
            N.assign((size_t)k2 * k2, 0.0);
            mem.M_times(wzr, v);
            for (int i = 0; i < k2; ++i)
                for (int j = 0; j < k2; ++j) {
                    double sum = 0.0;
                    for (int q = 0; q < k2; ++q) sum += mem.M[i * k2 + q] * WZ[q * k2 + j];
                    N[i * k2 + j] = (i == j ? 1.0 : 0.0) - sum / theta;
                }
            if (solve_dense(N, v, k2, 1)) {
                // d̂ = −(1/θ) r − (1/θ²) Z'W v
                du.resize(nf);
                wvfull.assign(n, 0.0);
                mem.W_times_add(v.data(), wvfull.data());             // W v
                // MN11: project the subspace minimizer onto the box; keep it if it is a descent
                // direction for the objective, otherwise truncate the step (v2.1 behaviour).
                // (full-length passes with selects instead of gathers over the free set: they vectorise; l = −inf / u = +inf
                //  where a bound does not exist, so the clamps are no-ops there; t is scratch for (z − x)·g)
                {
                    const double th2 = theta * theta;
                    const int* __restrict fxp = fixed.data();
                    double* __restrict zp = z.data();
                    double* __restrict tp = t.data();
                    for (int i = 0; i < n; ++i) {
                        const double dui = -rfull[i] / theta - wvfull[i] / th2;
                        const double zi = std::min(std::max(xcp[i] + dui, l[i]), u[i]);
                        zp[i] = fxp[i] ? xcp[i] : zi;
                        tp[i] = zp[i] - x[i];
                    }
                }
                for (int a = 0; a < nf; ++a) {
                    const int i = free_idx[a];
                    du[a] = -rfull[i] / theta - wvfull[i] / (theta * theta);
                }
                const double dd_p = dotn(t.data(), g.data(), n);
                if (dd_p > 0.0) {
                    double alpha = 1.0;
                    for (int a = 0; a < nf; ++a) {
                        const int i = free_idx[a];
                        const double dk = du[a];
                        if (dk < 0.0 && has_lower(nbd[i])) {
                            const double room = l[i] - xcp[i];
                            if (room >= 0.0) alpha = 0.0;
                            else if (dk * alpha < room) alpha = room / dk;
                        } else if (dk > 0.0 && has_upper(nbd[i])) {
                            const double room = u[i] - xcp[i];
                            if (room <= 0.0) alpha = 0.0;
                            else if (dk * alpha > room) alpha = room / dk;
                        }
                    }
                    for (int i = 0; i < n; ++i) z[i] = xcp[i];
                    for (int a = 0; a < nf; ++a) z[free_idx[a]] = xcp[free_idx[a]] + alpha * du[a];
                }
            }
        }

        // ---------------- line search along d = z − x -------------------------------------------
        double dnorm2 = 0.0, gd = 0.0;
        for (int i = 0; i < n; ++i) {                                 // (in index order: see the note above dotn)
            d[i] = z[i] - x[i];
            dnorm2 += d[i] * d[i];
            gd += g[i] * d[i];
        }
        bool ls_failed = false, noise_floor = false;
        if (!(gd < 0.0)) {
            ls_failed = true; // not a descent direction
        }
        double stp = 1.0, f_old = f;
        if (!ls_failed) {
            double stpmx = 1e10;
            if (constrained) {
                if (iter == 0) stpmx = 1.0;
                else {
                    // The Fortran's sequential rule (stpmx shrinks to a2/a1 whenever a1·stpmx passes a2) ends at the ratio of its
                    // LAST update, which lies within rounding of the smallest ratio.  Pass 1 (branch-free, vectorised) finds that
                    // minimum; pass 2 runs the sequential rule itself over the few variables whose ratio is within 1e-12 of it --
                    // every other variable either never updates or is overwritten by one of these later (its ratio is larger by
                    // far more than the rounding of the comparison), so the result is the sequential one bit for bit.
                    double rmin = stpmx;
                    bool at_bound = false;
                    {
                        const double* __restrict dp = d.data();
                        const double* __restrict lp = l.data();
                        const double* __restrict up = u.data();
                        const double* __restrict xp = x;
                        double* __restrict tp = t.data();                 // scratch: the ratios
                        for (int i = 0; i < n; ++i) {
                            const double a1 = dp[i];
                            const double a2 = a1 < 0.0 ? lp[i] - xp[i] : up[i] - xp[i];      // (±inf where the bound does not exist)
                            const double r = a1 != 0.0 ? a2 / a1 : kInf;                  // >= 0 inside the box; +inf: never binds
                            tp[i] = r;
                        }
                        // min over the ratios (exact in any order; four independent chains instead of one 4-cycle-latency chain)
                        double m0 = rmin, m1 = rmin, m2 = rmin, m3 = rmin;
                        int i = 0;
                        for (; i + 3 < n; i += 4) {
                            m0 = tp[i] < m0 ? tp[i] : m0;
                            m1 = tp[i + 1] < m1 ? tp[i + 1] : m1;
                            m2 = tp[i + 2] < m2 ? tp[i + 2] : m2;
                            m3 = tp[i + 3] < m3 ? tp[i + 3] : m3;
                        }
                        for (; i < n; ++i) m0 = tp[i] < m0 ? tp[i] : m0;
                        rmin = std::min(std::min(m0, m1), std::min(m2, m3));
                        at_bound = rmin <= 0.0;                           // a2 >= 0 with a1 < 0 (or a2 <= 0 with a1 > 0): already on the bound
                    }
                    if (at_bound) {
                        stpmx = 0.0;
                    } else {
                        const double cut = rmin * (1.0 + 1e-12);
                        for (int i = 0; i < n; ++i) {
                            if (!(t[i] <= cut)) continue;
                            const double a1 = d[i];
                            if (a1 < 0.0) {
                                const double a2 = l[i] - x[i];
                                if (a1 * stpmx < a2) stpmx = a2 / a1;
                            } else {
                                const double a2 = u[i] - x[i];
                                if (a1 * stpmx > a2) stpmx = a2 / a1;
                            }
                        }
                    }
                }
            }
            stp = (iter == 0 && !boxed) ? std::min(1.0 / std::sqrt(dnorm2), stpmx) : std::min(1.0, stpmx);
            x_old.assign(x, x + n);
            g_old = g;
            MoreThuente ls;
            ls.stpmax = stpmx;
            auto task = ls.start(f, gd, stp);
            int nfev = 0;
            if (task == MoreThuente::kError) ls_failed = true;
            while (!ls_failed && task == MoreThuente::kEvaluate) {
                if (stp == 1.0) for (int i = 0; i < n; ++i) x[i] = z[i];
                else for (int i = 0; i < n; ++i) x[i] = stp * d[i] + x_old[i];
                const double fnew = evaluate(x, g.data());
                ++nfev;
                if (!std::isfinite(fnew)) {
                    res.status = 5;
                    res.message = "ERROR: non-finite objective inside the feasible box";
                    for (int i = 0; i < n; ++i) x[i] = x_old[i];
                    res.f = f_old;
                    res.iterations = iter;
                    return res;
                }
                if (opt.stop_in_noise && fnew >= f_old &&
                    fnew - f_old <= opt.factr * kEps * std::max({std::fabs(f_old), std::fabs(fnew), 1.0})) {
                    // no decrease left beyond rounding noise (see LbfgsbOptions::stop_in_noise): keep the iterate
                    for (int i = 0; i < n; ++i) x[i] = x_old[i];
                    g = g_old;
                    f = f_old;
                    noise_floor = true;
                    break;
                }
                double gdn = 0.0;
                for (int i = 0; i < n; ++i) gdn += g[i] * d[i];
                f = fnew;
                task = ls.next(f, gdn, stp);
                if (task == MoreThuente::kEvaluate &&
                    (nfev >= opt.max_linesearch || res.evaluations >= opt.maxfun)) {
                    ls_failed = res.evaluations < opt.maxfun;
                    break;
                }
            }
            if (noise_floor) {
                res.status = 1;
                res.message = "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH (trial point within rounding noise)";
                break;
            }
            if (!ls_failed && task == MoreThuente::kError) ls_failed = true;
            if (ls_failed) { // restore the last iterate
                for (int i = 0; i < n; ++i) x[i] = x_old[i];
                g = g_old;
                f = f_old;
            }
        }
        if (ls_failed) {
            if (mem.col == 0) {
                res.status = 4;
                res.message = "ABNORMAL_TERMINATION_IN_LNSRCH";
                break;
            }
            mem.reset(); // refresh the limited-memory model and retry from the same point
            continue;
        }
        ++iter;

        // ---------------- termination tests ------------------------------------------------------
        pg = projected_gradient_norm(n, x, g.data(), l.data(), u.data(), hlm.data(), hum.data());
        if (pg <= opt.pgtol) {
            res.status = 0;
            res.message = "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL";
            break;
        }
        const double scale = std::max({std::fabs(f_old), std::fabs(f), 1.0});
        if (f_old - f <= opt.factr * kEps * scale) {
            res.status = 1;
            res.message = "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH";
            break;
        }
        if (iter >= opt.maxiter) {
            res.status = 2;
            res.message = "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT";
            break;
        }
        if (res.evaluations >= opt.maxfun) {
            res.status = 3;
            res.message = "STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT";
            break;
        }

        // ---------------- limited-memory update --------------------------------------------------
        double sy = 0.0, yy = 0.0, sg_old = 0.0;
        for (int i = 0; i < n; ++i) {                                 // (three sums in index order: see the note above dotn)
            s[i] = x[i] - x_old[i];
            y[i] = g[i] - g_old[i];
            sy += s[i] * y[i];
            yy += y[i] * y[i];
            sg_old += s[i] * g_old[i];
        }
        if (sy > kEps * (-sg_old)) mem.push(s, y, sy, yy); // else: curvature too small, skip
    }

    res.f = f;
    res.iterations = iter;
    res.proj_grad = projected_gradient_norm(n, x, g.data(), l.data(), u.data(), hlm.data(), hum.data());
    return res;
}

} // namespace cfmm
