// abi_upload.cpp -- pool validation and upload (include/cfmm_amd.h): what the reference's constructors validate
// (src/cfmms.jl:76-90) plus what its kernels silently assume (src/cfmms.jl:129 "Assumes that v > 0 and γ > 0"),
// the packed fee + token records, and the v-independent constants of the GeometricMean / UniV3 closed forms,
// prepared once on the host with the same IEEE operations the reference applies per sweep.
#include "ctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace cfmm;

namespace {

bool finite_pos(double x) { return std::isfinite(x) && x > 0.0; }

// |x| in [2^-kFastExp, 2^kFastExp]: the operand window of the sweep's fast division / square root (sweep.h)
bool in_fast_window(double x)
{
    uint64_t bits;
    std::memcpy(&bits, &x, sizeof bits);
    const int e = (int)((bits >> 52) & 0x7ff);
    return e >= 1023 - kFastExp && e <= 1023 + kFastExp;
}

// Packed fee + token record of a segment (sweep.h PackedFeeTok): {i1 | i2 << 16, index of the pool's fee among the
// segment's distinct fees}.  Built for every segment outside large-market mode (token ids fit 16 bits there); a segment
// with more than kMaxFeeTable distinct fees keeps the records for the tokens only (gvals empty: its launches read the
// fee from the gamma array).
int build_packed(cfmm_ctx* c, Segment& s, int64_t m, const double* gamma, const int32_t* Ai)
{
    s.pk = nullptr;
    s.gvals.clear();
    if (global_bins(c) || m == 0) return CFMM_OK;
    std::vector<PackedFeeTok> pk((size_t)m);
    std::vector<double> vals;
    uint64_t last_bits = 0;
    uint32_t last_idx = 0;
    bool have_last = false, table = true;
    for (int64_t i = 0; i < m; ++i) {
        uint32_t idx = 0;
        if (table) {
            uint64_t bits;
            std::memcpy(&bits, &gamma[i], sizeof bits);
            if (have_last && bits == last_bits) {
                idx = last_idx;
            } else {
                idx = (uint32_t)vals.size();
                for (uint32_t k = 0; k < (uint32_t)vals.size(); ++k) {   // <= 256 entries: a linear scan beats a hash map
                    uint64_t vb;
                    std::memcpy(&vb, &vals[k], sizeof vb);
                    if (vb == bits) { idx = k; break; }
                }
                if (idx == (uint32_t)vals.size()) {
                    if ((int)vals.size() == kMaxFeeTable) { table = false; idx = 0; }   // too many fee tiers: no table
                    else vals.push_back(gamma[i]);
                }
                last_bits = bits; last_idx = idx; have_last = true;
            }
        }
        pk[(size_t)i].tok = (uint32_t)Ai[2 * i] | ((uint32_t)Ai[2 * i + 1] << 16);
        pk[(size_t)i].gidx = idx;
    }
    int rc = upload(c, &s.pk, pk.data(), (size_t)m);
    if (rc != CFMM_OK) return rc;
    if (table) s.gvals.swap(vals);
    return CFMM_OK;
}

// What two_coin_check_cast (src/cfmms.jl:76-90) enforces structurally is implied by the [m][2]
// layout; here we check the values the closed forms assume.
int check_two_coin(cfmm_ctx* c, int64_t m, const double* R, const double* gamma, const int32_t* Ai)
{
    if (m < 0) return fail(c, CFMM_ERR_INVALID_ARG, "negative pool count");
    if (m > 0 && (!R || !gamma || !Ai)) return fail(c, CFMM_ERR_INVALID_ARG, "null pool array");
    for (int64_t i = 0; i < m; ++i) {
        if (!finite_pos(R[2 * i]) || !finite_pos(R[2 * i + 1]))
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: reserves must be finite and > 0", (long long)i);
        if (!finite_pos(gamma[i]))
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: fee gamma must be finite and > 0", (long long)i);
        const int32_t a = Ai[2 * i], b = Ai[2 * i + 1];
        if (a < 0 || a >= c->n || b < 0 || b >= c->n)
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: token index out of range [0, %d)", (long long)i, c->n);
        if (a == b)
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: the two token indices must differ", (long long)i);
    }
    return CFMM_OK;
}

// Largest price P for which find_arb_pos (src/cfmms.jl:321-337) DRAINS a tick with the prepared constants k, s_in = R_in + α,
// δmax: dd = sqrt(k/P) − s_in is > 0 and >= δmax.  The test is monotone in P (IEEE division, square root and subtraction are
// correctly rounded, hence monotone), so there is exactly one such double; it is found on the test ITSELF -- gallop from the
// algebraic boundary k/(s_in + δmax)², then bisect on the bit patterns -- so that `price <= T` on the device is the
// reference's floating-point decision, not an approximation of it.  0: the tick never drains for a positive price.
double drain_threshold(double k, double s_in, double dmax)
{
    auto drains = [&](double P) {
        const double dd = std::sqrt(k / P) - s_in;
        return dd > 0 && dd >= dmax;
    };
    auto bits = [](double x) { int64_t b; std::memcpy(&b, &x, sizeof b); return b; };
    auto from = [](int64_t b) { double x; std::memcpy(&x, &b, sizeof x); return x; };
    const int64_t lo_lim = bits(0x1p-1000), hi_lim = bits(0x1p1000);
    double c0 = k / ((s_in + dmax) * (s_in + dmax));
    if (!(c0 > 0x1p-1000)) c0 = 0x1p-1000;     // (also catches NaN)
    if (!(c0 < 0x1p1000)) c0 = 0x1p1000;
    int64_t lo, hi;                            // drains(lo), !drains(hi)
    const int64_t cb = bits(c0);
    if (drains(c0)) {
        lo = cb;
        for (int64_t step = 1;; step *= 2) {
            const int64_t nb = lo + step;
            if (nb >= hi_lim) {
                if (drains(from(hi_lim))) return from(hi_lim);
                hi = hi_lim;
                break;
            }
            if (!drains(from(nb))) { hi = nb; break; }
            lo = nb;
        }
    } else {
        hi = cb;
        for (int64_t step = 1;; step *= 2) {
            const int64_t nb = hi - step;
            if (nb <= lo_lim) {
                if (!drains(from(lo_lim))) return 0.0;
                lo = lo_lim;
                break;
            }
            if (drains(from(nb))) { lo = nb; break; }
            hi = nb;
        }
    }
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (drains(from(mid))) lo = mid;
        else hi = mid;
    }
    return from(lo);
}

int add_segment_common(cfmm_ctx* c, Segment&& s, const int32_t* Ai)
{
    if (s.m == 0) {   // an empty batch contributes no pools, no trades and no partial rows: not stored
        free_segment(s);
        return CFMM_OK;
    }
    if (global_bins(c) && s.m > 0) s.h_ai.assign(Ai, Ai + 2 * s.m);
    c->segs.push_back(std::move(s));
    c->geometry_dirty = true;
    c->have_out = false;
    c->have_trades = false;
    return CFMM_OK;
}

} // namespace

namespace cfmm {

void free_segment(Segment& s)
{
    (void)hipFree(s.R); (void)hipFree(s.w); (void)hipFree(s.gamma); (void)hipFree(s.Ai);
    (void)hipFree(s.eta); (void)hipFree(s.lR); (void)hipFree(s.pk);
    (void)hipFree(s.cur_a); (void)hipFree(s.cur_b); (void)hipFree(s.cur_c); (void)hipFree(s.curR);
    (void)hipFree(s.pg); (void)hipFree(s.cp); (void)hipFree(s.walk); (void)hipFree(s.ticks); (void)hipFree(s.thr);
    (void)hipFree(s.head);
    s = Segment{};
}

// Validates m UniV3 pools and prepares + uploads the find_arb_pos constants (see UniV3Ops) into `s`.
int univ3_build(cfmm_ctx* c, Segment& s, int64_t m, const double* current_price, const double* gamma, const int32_t* Ai,
                const int64_t* tick_off, const double* lower_ticks, const double* liquidity)
{
    const int64_t T = m > 0 ? tick_off[m] : 0;
    if (T < 0 || 2 * (T + 2 * m) > (int64_t)0x3fffffff) return fail(c, CFMM_ERR_UNSUPPORTED, "too many ticks in one segment");
    std::vector<double2> pg((size_t)m), cur_a((size_t)m), cur_b((size_t)m), curR((size_t)m);
    std::vector<double> cur_c((size_t)m);
    std::vector<TickRec> ticks;
    std::vector<int4> walk((size_t)m);
    int longest = 0;
    bool fast = true;   // every operand of the sweep's divisions / square roots inside the fast window (sweep.h)
    ticks.reserve((size_t)T + 2 * (size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        const int64_t o = tick_off[i], nt = tick_off[i + 1] - o;
        if (nt < 1) return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: needs at least one tick", (long long)i);
        if (!finite_pos(current_price[i]))
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: current_price must be finite and > 0", (long long)i);
        if (!finite_pos(gamma[i]))
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: fee gamma must be finite and > 0", (long long)i);
        const int32_t a = Ai[2 * i], b = Ai[2 * i + 1];
        if (a < 0 || a >= c->n || b < 0 || b >= c->n)
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: token index out of range [0, %d)", (long long)i, c->n);
        if (a == b) return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: the two token indices must differ", (long long)i);
        const double* lt = lower_ticks + o;
        const double* lq = liquidity + o;
        for (int64_t j = 0; j < nt; ++j) {
            if (!finite_pos(lt[j])) return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld tick %lld: price must be finite and > 0", (long long)i, (long long)j);
            if (j > 0 && !(lt[j] < lt[j - 1]))
                return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: lower_ticks must be strictly descending", (long long)i);
            if (!(lq[j] >= 0.0) || !std::isfinite(lq[j]))
                return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld tick %lld: liquidity must be finite and >= 0", (long long)i, (long long)j);
        }
        const double cp = current_price[i];
        fast = fast && in_fast_window(cp) && in_fast_window(gamma[i]);
        for (int64_t j = 0; j < nt; ++j) fast = fast && (lq[j] == 0.0 || in_fast_window(lq[j]));
        // src/cfmms.jl:235: searchsortedlast(lower_ticks, current_price, rev=true)
        int64_t lo = 0, hi = nt + 1;
        while (lo < hi - 1) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (lt[mid - 1] < cp) hi = mid;
            else lo = mid;
        }
        const int64_t ct = lo;
        if (ct < 1)
            return fail(c, CFMM_ERR_INVALID_ARG,
                        "pool %lld: current_price above the first tick (the reference would index tick 0)", (long long)i);
        // compute_at_tick(cfmm, idx), src/cfmms.jl:294-313 (idx 1-based)
        auto at_tick = [&](int64_t idx, double& k, double& al, double& be, double& R1, double& R2) {
            k = lq[idx - 1];
            const double pplus = lt[idx - 1];                 // :251
            const double pminus = idx < nt ? lt[idx] : 0.0;   // :254-259
            al = std::sqrt(k / pplus);
            be = std::sqrt(k * pminus);
            const double p = idx > ct ? pplus : (idx < ct ? pminus : cp);
            R1 = std::sqrt(k / p) - al;
            R2 = std::sqrt(k * p) - be;
        };
        {   // the current tick, shared by both walks
            double k, al, be, R1, R2;
            at_tick(ct, k, al, be, R1, R2);
            const double sA = R1 + al, sB = R2 + be;
            cur_a[(size_t)i] = make_double2(k, sA);
            cur_b[(size_t)i] = make_double2(sB, k / be - sA);   // :329
            cur_c[(size_t)i] = k / al - sB;                     // :329 on the flipped pool (:289)
            curR[(size_t)i] = make_double2(R1, R2);
            if (k == 0) { cur_b[(size_t)i].y = 0.0; cur_c[(size_t)i] = 0.0; } // 0/0: never read (k == 0 is skipped)
        }
        // Walk lists (UniV3Ops::solve_dir): the non-empty ticks beyond the current one, in walk order; every record also
        // carries the sums of the drained ticks BEFORE it, starting from what the current tick contributes when it drains
        // ({δmax, R_out}; nothing if it is empty) and accumulated with the walk's own additions; one closing record per
        // list carries the sums of the whole list.
        double kc, alc, bec, R1c, R2c;
        at_tick(ct, kc, alc, bec, R1c, R2c);
        int4 w;
        w.x = (int)ticks.size();
        int cnt = 0;
        double2 run = kc != 0 ? make_double2(cur_b[(size_t)i].y, R2c) : make_double2(0.0, 0.0);   // price falling: δmax↑, R₂ out
        for (int64_t idx = ct + 1; idx <= nt; ++idx) {        // get_upper_pools beyond the current tick, :316
            double k, al, be, R1, R2;
            at_tick(idx, k, al, be, R1, R2);
            if (k == 0) continue;                             // is_empty_pool, :288
            const double s_in = R1 + al, dmax = k / be - s_in;
            ticks.push_back(TickRec{make_double2(k, s_in), make_double2(dmax, R2 + be), R2, 0.0, run});   // :329, :334
            run.x += dmax;
            run.y += R2;
            ++cnt;
        }
        ticks.push_back(TickRec{make_double2(0.0, 0.0), make_double2(0.0, 0.0), 0.0, 1.0, run});           // closing record (pad = 1 marks it)
        w.y = cnt;
        w.z = (int)ticks.size();
        cnt = 0;
        run = kc != 0 ? make_double2(cur_c[(size_t)i], R1c) : make_double2(0.0, 0.0);                      // price rising (flipped pool, :289)
        for (int64_t idx = ct - 1; idx >= 1; --idx) {         // flip_sides.(get_lower_pools), :317,:289
            double k, al, be, R1, R2;
            at_tick(idx, k, al, be, R1, R2);
            if (k == 0) continue;
            const double s_in = R2 + be, dmax = k / al - s_in;
            ticks.push_back(TickRec{make_double2(k, s_in), make_double2(dmax, R1 + al), R1, 0.0, run});
            run.x += dmax;
            run.y += R1;
            ++cnt;
        }
        ticks.push_back(TickRec{make_double2(0.0, 0.0), make_double2(0.0, 0.0), 0.0, 1.0, run});
        w.w = cnt;
        longest = std::max(longest, std::max(w.y, w.w));
        walk[(size_t)i] = w;
        pg[(size_t)i] = make_double2(cp, gamma[i]);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    s.kind = CFMM_KIND_UNIV3;
    s.m = m;
    s.n_ticks_total = T;
    // drain thresholds of all records (the closing records and ticks that end the walk when reached -- δmax = 0 or R_out = 0,
    // :363-365 -- get 0 = "never"), a few bisection steps each: spread over the host's cores
    std::vector<double> thr(ticks.size(), 0.0);
    {
        const size_t nrec = ticks.size();
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const unsigned nthr = nrec > 65536 ? hw : 1;
        auto work = [&](size_t lo, size_t hi) {
            for (size_t e = lo; e < hi; ++e) {
                const TickRec& r = ticks[e];
                if (r.thr != 0.0 || r.dt.x == 0.0 || r.rout == 0.0) continue;   // (thr == 1 marks a closing record until here)
                thr[e] = drain_threshold(r.ks.x, r.ks.y, r.dt.x);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(work, nrec * t / nthr, nrec * (t + 1) / nthr);
        work(0, nrec / nthr);
        for (auto& th : pool) th.join();
    }
    for (size_t e = 0; e < ticks.size(); ++e) ticks[e].thr = thr[e];   // every record carries its own threshold (closing records: 0)
    thr.resize(ticks.size() + 4, 0.0);   // the scan reads four thresholds at a time
    s.has_walk = longest > 0 ? 1 : 0;
    // threshold heads (sweep.h UniV3Pools::head): the first four thresholds of both lists of every pool as floats rounded DOWN
    std::vector<uint4> head;
    if (s.has_walk) {
        head.resize(2 * (size_t)m);
        const auto enc = [](double T) -> unsigned {
            if (T == 0.0) return 0u;                                   // never drains (also: closing record, past the list)
            if (!(T >= 0x1p-120 && T <= 0x1p120)) return 0x7fc00000u;   // outside the comfortable binary32 range: NaN = "ask thr[]"
            float f = (float)T;
            if ((double)f > T) f = std::nextafterf(f, 0.0f);           // round toward zero = down (T > 0)
            unsigned b;
            std::memcpy(&b, &f, sizeof b);
            return b;
        };
        for (int64_t i = 0; i < m; ++i) {
            const int4 w = walk[(size_t)i];
            unsigned h[8];
            for (int k = 0; k < 4; ++k) {
                h[k] = k < w.y ? enc(thr[(size_t)w.x + k]) : 0u;       // beyond the list: the closing record's "never"
                h[4 + k] = k < w.w ? enc(thr[(size_t)w.z + k]) : 0u;
            }
            head[2 * (size_t)i] = make_uint4(h[0], h[1], h[2], h[3]);
            head[2 * (size_t)i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
    }
    s.fast_ok = fast ? 1 : 0;
    int rc;
    if ((rc = upload(c, &s.pg, pg.data(), (size_t)m)) || (rc = upload(c, &s.Ai, Ai, (size_t)m)) ||
        (rc = upload(c, &s.cur_a, cur_a.data(), (size_t)m)) || (rc = upload(c, &s.cur_b, cur_b.data(), (size_t)m)) ||
        (rc = upload(c, &s.cur_c, cur_c.data(), (size_t)m)) || (rc = upload(c, &s.curR, curR.data(), (size_t)m)) ||
        (rc = upload(c, &s.walk, walk.data(), (size_t)m)) || (rc = upload(c, &s.ticks, ticks.data(), ticks.size())) ||
        (rc = upload(c, &s.thr, thr.data(), thr.size())) || (rc = upload(c, &s.head, head.data(), head.size())) ||
        (rc = upload(c, &s.cp, current_price, (size_t)m)) || (rc = build_packed(c, s, m, gamma, Ai))) {
        free_segment(s);
        return rc;
    }
    return CFMM_OK;
}

} // namespace cfmm

extern "C" {

int cfmm_pools_add_product(cfmm_ctx* c, int64_t m, const double* R, const double* gamma, const int32_t* Ai)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int rc = check_two_coin(c, m, R, gamma, Ai);
    if (rc != CFMM_OK) return rc;
    if (!c->shards.empty())
        return multi_add(c, CFMM_KIND_PRODUCT, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) -> int {
            return cfmm_pools_add_product(child, hi - lo, R + 2 * lo, gamma + lo, Ai + 2 * lo);
        });
    HIP_TRY(c, hipSetDevice(c->device));
    Segment s;
    s.kind = CFMM_KIND_PRODUCT;
    s.m = m;
    s.fast_ok = 1;
    for (int64_t i = 0; i < m && s.fast_ok; ++i)
        s.fast_ok = in_fast_window(R[2 * i]) && in_fast_window(R[2 * i + 1]) && in_fast_window(gamma[i]);
    if ((rc = upload(c, &s.R, R, (size_t)m)) || (rc = upload(c, &s.gamma, gamma, (size_t)m)) ||
        (rc = upload(c, &s.Ai, Ai, (size_t)m)) || (rc = build_packed(c, s, m, gamma, Ai))) {
        free_segment(s);
        return rc;
    }
    return add_segment_common(c, std::move(s), Ai);
}

int cfmm_pools_add_geomean(cfmm_ctx* c, int64_t m, const double* R, const double* w, const double* gamma,
                           const int32_t* Ai)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int rc = check_two_coin(c, m, R, gamma, Ai);
    if (rc != CFMM_OK) return rc;
    if (m > 0 && !w) return fail(c, CFMM_ERR_INVALID_ARG, "null weight array");
    for (int64_t i = 0; i < m; ++i)
        if (!finite_pos(w[2 * i]) || !finite_pos(w[2 * i + 1]))
            return fail(c, CFMM_ERR_INVALID_ARG, "pool %lld: weights must be finite and > 0", (long long)i);
    if (!c->shards.empty())
        return multi_add(c, CFMM_KIND_GEOMEAN, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) -> int {
            return cfmm_pools_add_geomean(child, hi - lo, R + 2 * lo, w + 2 * lo, gamma + lo, Ai + 2 * lo);
        });
    // v-independent pieces of the log-space closed forms (sweep_kernels.hip, GeoMeanLogOps)
    std::vector<double2> lR((size_t)m);
    std::vector<double> etas((size_t)m);
    bool fast = true;
    for (int64_t i = 0; i < m; ++i) {
        const double e = w[2 * i] / w[2 * i + 1]; // src/cfmms.jl:188
        fast = fast && in_fast_window(R[2 * i]) && in_fast_window(R[2 * i + 1]) && in_fast_window(gamma[i]) && in_fast_window(e);
        const double lg = std::log(gamma[i]), le = std::log(e), l1 = std::log(R[2 * i]), l2 = std::log(R[2 * i + 1]);
        etas[(size_t)i] = e;
        lR[(size_t)i] = make_double2(((lg + le) + l2) + e * l1, e * ((lg + l1) - le) + l2);   // {Q1, Q2}
    }
    HIP_TRY(c, hipSetDevice(c->device));
    Segment s;
    s.kind = CFMM_KIND_GEOMEAN;
    s.m = m;
    s.fast_ok = fast ? 1 : 0;
    if ((rc = upload(c, &s.eta, etas.data(), (size_t)m)) || (rc = upload(c, &s.lR, lR.data(), (size_t)m)) ||
        (rc = upload(c, &s.R, R, (size_t)m)) || (rc = upload(c, &s.w, w, (size_t)m)) ||
        (rc = upload(c, &s.gamma, gamma, (size_t)m)) || (rc = upload(c, &s.Ai, Ai, (size_t)m)) ||
        (rc = build_packed(c, s, m, gamma, Ai))) {
        free_segment(s);
        return rc;
    }
    return add_segment_common(c, std::move(s), Ai);
}

int cfmm_pools_add_univ3(cfmm_ctx* c, int64_t m, const double* current_price, const double* gamma,
                         const int32_t* Ai, const int64_t* tick_off, const double* lower_ticks,
                         const double* liquidity)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (m < 0) return fail(c, CFMM_ERR_INVALID_ARG, "negative pool count");
    if (m > 0 && (!current_price || !gamma || !Ai || !tick_off || !lower_ticks || !liquidity))
        return fail(c, CFMM_ERR_INVALID_ARG, "null pool array");
    if (m > 0 && tick_off[0] != 0) return fail(c, CFMM_ERR_INVALID_ARG, "tick_off[0] must be 0");
    if (!c->shards.empty()) {
        for (int64_t i = 0; i < m; ++i)
            if (tick_off[i + 1] < tick_off[i]) return fail(c, CFMM_ERR_INVALID_ARG, "tick_off must be non-decreasing");
        return multi_add(c, CFMM_KIND_UNIV3, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) -> int {
            std::vector<int64_t> off((size_t)(hi - lo + 1));
            for (int64_t i = lo; i <= hi; ++i) off[(size_t)(i - lo)] = tick_off[i] - tick_off[lo];   // CSR rebased to the block
            return cfmm_pools_add_univ3(child, hi - lo, current_price + lo, gamma + lo, Ai + 2 * lo, off.data(),
                                        lower_ticks + tick_off[lo], liquidity + tick_off[lo]);
        });
    }
    Segment s;
    int rcb = univ3_build(c, s, m, current_price, gamma, Ai, tick_off, lower_ticks, liquidity);
    if (rcb != CFMM_OK) return rcb;
    // host copy of the pool definitions: update_reserves! re-derives the tick constants from them
    s.h_cp.assign(current_price, current_price + m);
    s.h_gamma.assign(gamma, gamma + m);
    s.h_ai.assign(Ai, Ai + 2 * m);
    s.h_tick_off.assign(tick_off, tick_off + m + 1);
    s.h_lt.assign(lower_ticks, lower_ticks + tick_off[m]);
    s.h_liq.assign(liquidity, liquidity + tick_off[m]);
    return add_segment_common(c, std::move(s), Ai);
}

int cfmm_pools_clear(cfmm_ctx* c)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {
        for (cfmm_ctx* child : c->shards) cfmm_pools_clear(child);
        c->psegs.clear();
        c->m_total = 0;
        c->have_out = c->have_trades = false;
        return CFMM_OK;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->segs) free_segment(s);
    c->segs.clear();
    c->m_total = 0;
    c->rows_total = 0;
    c->geometry_dirty = true;
    c->have_out = c->have_trades = false;
    return CFMM_OK;
}

} // extern "C"
