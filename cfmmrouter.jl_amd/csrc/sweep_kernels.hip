// sweep_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the per-CFMM arbitrage
// sweep of CFMMRouter.jl and the reductions route! consumes.
//
// What is replaced (paths relative to the reference root):
//   find_arb!(r::Router, v)            src/router.jl:38-42    -> sweep_kernel / sweep_multi (sweep_body)
//   find_arb!(.., ::ProductTwoCoin)    src/cfmms.jl:125-140   -> ProductOps::solve
//   find_arb!(.., ::GeometricMeanTwoCoin) src/cfmms.jl:180-196 -> GeoMeanLogOps::solve (default), GeoMeanOps::solve
//   find_arb!(.., ::UniV3) + helpers   src/cfmms.jl:294-395   -> UniV3Ops::solve_dir
//   acc loop of fn                     src/router.jl:79-83    -> per-lane acc + wave shuffles
//   scatter loop of g! / netflows!     src/router.jl:98-100, :111-119 -> LDS bins + reduce_partials
//                                      (n_tokens > 8192: flow array + gather_chunks / token_fold)
//
// Mapping to the machine.  A two-coin closed form is ~25 dependent flops with no parallelism
// inside a pool, so the unit of work is ONE LANE PER POOL (64 pools per wavefront); the
// data-parallel axis is the pool index, exactly the axis the reference threads over.  Pool
// state is stored as coalesced streams (reserve pairs 16 B + a packed {tokens, fee index} record 8 B per
// lane) and trades leave as one 16 B/lane stream.  Each wavefront owns a private
// copy of the n_tokens netflow bins in LDS and scatters (Lambda - Delta) into it with
// ds_add_f64; the block then folds its copies in a fixed order and writes one partial row to
// global memory.  A second tiny kernel folds the rows, again in a fixed order, so a sweep
// is reproducible bit-for-bit for a fixed launch geometry -- there is no global float atomic.
// The dual scalar is accumulated per lane in tile order and folded with wave shuffles.
//
// Numerics.  Everything is binary64.  This translation unit is compiled with
// -ffp-contract=off and the expressions keep the reference's operation order; with IEEE
// correctly-rounded / and sqrt the ProductTwoCoin and UniV3 trades are bit-identical to the
// reference arithmetic (everything v-independent in the UniV3 forms is prepared at upload with the
// same IEEE operations).  GeometricMeanTwoCoin is evaluated in log space by default (~1e-15 of
// the reserve scale from the reference's pow forms; GeoMeanOps keeps those, with the device
// library's pow).  HBM-bound by design: no MFMA (there is no contraction anywhere on this path).

#include "sweep.h"

#include <hip/hip_ext.h>

namespace cfmm {

struct Trade {
    double d1, d2, l1, l2;
};

// Julia's max(x, 0.0): NaN propagates, max(-0.0, 0.0) == +0.0.
__device__ __forceinline__ double max0(double x)
{
    double r = x > 0.0 ? x : 0.0;
    return (x != x) ? x : r;
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded binary64 division and square root without the range scaffolding
// ---------------------------------------------------------------------------------------------
// For a / b the compiler emits   d = v_div_scale(b, b, a);  y = v_rcp(d);  two Newton steps on y (4 fma);
// n = v_div_scale(a, b, a);  q = n·y;  r = fma(−d, q, n);  v_div_fmas(r, y, q);  v_div_fixup     (11 instructions),
// and for sqrt(x) a compare / select / ldexp pair around   y = v_rsq(x);  s = x·y;  h = y/2;  two coupled Newton steps
// (7 fma)   plus a class test                                                                        (16 instructions).
// The scaffolding only acts outside a huge exponent range: for finite, normal operands with |exponent| <= 300 or so
// v_div_scale returns its input, v_div_fmas is a plain fma, v_div_fixup returns its first operand and the ldexp pair
// scales by 2^0.  The SAME core sequences without it therefore return the SAME correctly rounded bits whenever every
// operand is inside [2^-kFastExp, 2^kFastExp] -- pool constants are checked at upload, the prices by every block while
// it stages them (`FAST` below); anything else takes the compiler's sequences.  What this buys beyond the 3 + 7
// instructions: the refined reciprocal y depends on the DIVISOR only, so it is computed once per token (prices) and
// once per fee tier while they are staged in LDS, and a division by a price or by a fee costs three instructions.
// (tests/test_gpu_parity.py: every ProductTwoCoin / UniV3 trade bit-equal to the CPU restatement with fast_math on and
// off; tests/native/fastmath_check.hip: 2^30 random operands against the compiler's / and sqrt.)
__device__ __forceinline__ double rcp_refined(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
// a / b given yb = rcp_refined(b);  a finite (any sign, zero included: a = ±0 returns a·yb = ±0 like IEEE for b > 0 --
// the residual fma then adds +0 to −0, so the sign of a zero quotient is restored explicitly)
__device__ __forceinline__ double div_by(double a, double b, double yb)
{
    const double q = a * yb;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, yb, q);
}
__device__ __forceinline__ double div_by_signed_zero(double a, double b, double yb)
{
    const double q = div_by(a, b, yb);
    return a == 0.0 ? a : q;      // b > 0 everywhere this is used: ±0 / b = ±0
}
__device__ __forceinline__ double fast_div(double a, double b) { return div_by(a, b, rcp_refined(b)); }
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    double d = __builtin_fma(-s, s, x);
    h = __builtin_fma(h, r, h);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    return __builtin_fma(d, h, s);
}
// exp(x) for |x| < 700 (no overflow / underflow handling: inside the window of the fast arithmetic the argument is the
// logarithm of a reserve, |x| <= ~312), < 1 ulp like the device library's: x = k ln2 + r, |r| <= ln2/2,
// exp(r) = 1 + r + r^2 g(r) with g the degree-9 Chebyshev interpolant of (e^r - 1 - r)/r^2 (approximation error 1.6e-17,
// scripts/fit_exp.py), result ldexp(., k).  19 instructions against the library's 38: that one handles the whole
// double range (two compares, four selects) and the compiler expands its Horner steps into v_mov + v_fmac pairs; here
// each step is ONE v_fma with the coefficient in scalar registers.  NaN in, NaN out.
__device__ __forceinline__ double fma_sc(double x, double acc, double c)   // x * acc + c, c from SGPRs
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(acc), "s"(c));
    return r;
}
__device__ __forceinline__ double fast_exp(double x)
{
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-k, 0x1.62e42fefa39efp-1, x);
    r = __builtin_fma(-k, 0x1.abc9e3b39803fp-56, r);
    double p = 0x1.af39091a8441ap-26;
    p = fma_sc(r, p, 0x1.2891d2ecb3ed9p-22);
    p = fma_sc(r, p, 0x1.71de0d863c737p-19);
    p = fma_sc(r, p, 0x1.a019b8cbe6585p-16);
    p = fma_sc(r, p, 0x1.a01a01a7ce75dp-13);
    p = fma_sc(r, p, 0x1.6c16c1789caa1p-10);
    p = fma_sc(r, p, 0x1.11111111109a6p-7);
    p = fma_sc(r, p, 0x1.5555555553d38p-5);
    p = fma_sc(r, p, 0x1.5555555555556p-3);
    p = fma_sc(r, p, 0x1.0000000000001p-1);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// |x| in [2^-kFastExp, 2^kFastExp] (false for NaN, infinities, zero, denormals)
__device__ __forceinline__ bool in_fast_window(double x)
{
    const int e = (__double2hiint(x) >> 20) & 0x7ff;
    return e >= 1023 - kFastExp && e <= 1023 + kFastExp;
}

// Keeps a freshly loaded value in registers at this point of the program.  (Without it the compiler defers the second
// half of a {γ, rcp(γ)} table read into the branch that uses it by SELECTING BETWEEN POINTERS -- the LDS entry or a
// stack slot holding the 0.0 of the unpacked path -- and reads it back with a flat load: scratch traffic per tile.)
__device__ __forceinline__ double pinned(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Prices of a pool's two tokens as the sweep hands them to solve(): the values, their refined reciprocals (FAST only;
// staged per token in LDS), the fee's refined reciprocal (FAST only: from the LDS fee table, or computed per pool) and
// log v2 − log v1 (log-space GeometricMean only).
struct Px {
    double v1, v2, y1, y2, yg, dlv;
};

// What Ops::solve_dir returns: a two-coin pool trades in at most one direction (the reference's four outputs are
// (Δ₁, 0), (0, Λ₂) or (0, Δ₂), (Λ₁, 0)), so the common case travels as {d, l} + a direction and the epilogue of a pool
// (trade record, dual scalar, netflow bins) works on two values instead of four.  kDirBoth: all four values in a Trade
// (γ > 1 pools trading both ways, the overlap of ProductTwoCoin's predicates, NaN prices).
constexpr int kDirNone = 0, kDir1 = 1, kDir2 = 2, kDirBoth = 3;
__device__ __forceinline__ void expand_dir(int dir, double d, double l, Trade& t)
{
    if (dir == kDirBoth) return;
    t.d1 = dir == kDir1 ? d : 0.0;
    t.d2 = dir == kDir2 ? d : 0.0;
    t.l1 = dir == kDir2 ? l : 0.0;
    t.l2 = dir == kDir1 ? l : 0.0;
}

// ---------------------------------------------------------------------------------------------
// ProductTwoCoin -- src/cfmms.jl:125-140
// ---------------------------------------------------------------------------------------------
// Every Ops::load() only ISSUES loads (no arithmetic on what it loaded): the compiler then keeps all of a tile's loads
// in flight together.  (Round 2 unpacked the {tokens, fee index} record inside load(); the compiler answered with
// s_waitcnt vmcnt(0) BEFORE it issued the reserve load -- two dependent memory round trips per tile.)  The record is
// taken apart in resolve(), after the tile's data has arrived.

struct ProductOps {
    static constexpr bool kNeedsLogPrices = false;
    static constexpr bool kPrefetch = true;      // tile_loop: request the next tile's pool state before solving this one
    struct Raw {
        double2 R;
        double g;
        int2 ai;      // packed: {tok, gidx} until resolve()
        double yg;    // refined reciprocal of the fee (FAST with a fee table)
    };
    ProductPools p;
    // GBINS (n_tokens > 8192): the plain gamma / Ai arrays.  Otherwise ALWAYS the packed {tokens, fee index} record --
    // one load, no choice between two pointers for the compiler to merge and sink to the use (round 3: with a
    // `pk ? pk[i] : Ai[i]` in here the record's load ended up at the TOP of the next tile, its latency exposed again) --
    // plus the fee itself from the gamma array when the launch has no fee table (p.gbase < 0: too many fee tiers).
    template <bool GBINS>
    __device__ __forceinline__ Raw load(int64_t i) const
    {
        Raw r;
        r.R = p.R[i];
        r.yg = 0.0;
        if constexpr (GBINS) {
            r.g = p.gamma[i];
            r.ai = p.Ai[i];
        } else {
            const PackedFeeTok k = p.pk[i];
            r.ai = make_int2((int)k.tok, (int)k.gidx);
            r.g = p.gbase < 0 ? p.gamma[i] : 0.0;
        }
        return r;
    }
    // after stage_prices(): take the packed record apart; the fee from the LDS table {γ, rcp_refined(γ)}, or -- no table --
    // its reciprocal refined here (FAST only)
    template <bool GBINS, bool FAST>
    __device__ __forceinline__ void resolve(Raw& r, const double2* gtab_lds) const
    {
        if constexpr (!GBINS) {
            const unsigned tok = (unsigned)r.ai.x;
            if (p.gbase >= 0) {
                const double2 gy = gtab_lds[p.gbase + r.ai.y];
                r.g = gy.x;
                r.yg = pinned(gy.y);
            } else if constexpr (FAST) {
                r.yg = rcp_refined(r.g);
            }
            r.ai = make_int2((int)(tok & 0xffffu), (int)(tok >> 16));
        }
    }
    __device__ __forceinline__ int2 tokens(const Raw& r) const { return r.ai; }
    // All four closed forms exactly as written in the reference (:134-138).
    __device__ __forceinline__ void solve_full(double R1, double R2, double g, double v1, double v2, Trade& t) const
    {
        const double k = R1 * R2;          // :132
        const double m12 = v2 / v1;        // m of :134/:138
        const double m21 = v1 / v2;        // m of :135/:137
        const double gm12 = g * m12;       // γ*m (== m*γ bitwise)
        const double gm21 = g * m21;
        t.d1 = max0(sqrt(gm12 * k) - R1) / g;   // :125,:134
        t.d2 = max0(sqrt(gm21 * k) - R2) / g;   // :125,:135
        t.l1 = max0(R1 - sqrt(k / gm21));       // :126,:137
        t.l2 = max0(R2 - sqrt(k / gm12));       // :126,:138
    }

    // At most one direction trades (Δ₁,Λ₂ > 0 ⇔ γ·v₂R₂ > v₁R₁;  Δ₂,Λ₁ > 0 ⇔ γ·v₁R₁ > v₂R₂), so only
    // that direction's two closed forms are evaluated -- with the reference's own expressions on
    // the selected operands, hence bit-identical values.  The predicates carry a 1e-12 relative
    // margin (>> the 1e-16 rounding of the forms), so a direction is only skipped where the
    // reference's max(·, 0) provably clamps to 0; the (measure-zero) overlap runs the full forms.
    // Returns the direction of the trade: kDirNone, kDir1 (Δ₁ = d, Λ₂ = l), kDir2 (Δ₂ = d, Λ₁ = l) or kDirBoth
    // (the four values in t: the overlap of the two predicates, or NaN inputs).
    template <bool FAST>
    __device__ __forceinline__ int solve_dir(const Raw& r, const Px& px, double& d, double& l, Trade& t) const
    {
        const double R1 = r.R.x, R2 = r.R.y, g = r.g, v1 = px.v1, v2 = px.v2;
        constexpr double kMargin = 1.0 + 1e-12;
        const double a = v1 * R1, b = v2 * R2;
        const bool p1 = (g * b) * kMargin >= a;    // direction 1 possibly active
        const bool p2 = (g * a) * kMargin >= b;    // direction 2 possibly active
        d = l = 0.0;
        if (p1 != p2) {
            const double k = R1 * R2;                          // :132
            const double r_in = p1 ? R1 : R2, r_out = p1 ? R2 : R1;
            if constexpr (FAST) {
                // m = v_out / v_in through the divisor's staged reciprocal; operands inside the window: same bits
                const double gm = g * div_by(p1 ? v2 : v1, p1 ? v1 : v2, p1 ? px.y1 : px.y2);
                d = div_by(__builtin_fmax(fast_sqrt(gm * k) - r_in, 0.0), g, px.yg);   // :125 (finite: max0 == fmax)
                l = __builtin_fmax(r_out - fast_sqrt(fast_div(k, gm)), 0.0);           // :126
            } else {
                const double gm = g * ((p1 ? v2 : v1) / (p1 ? v1 : v2));   // γ*m, m = v_out / v_in
                d = max0(sqrt(gm * k) - r_in) / g;             // :125
                l = max0(r_out - sqrt(k / gm));                // :126
            }
            return p1 ? kDir1 : kDir2;
        }
        if (p1 || a != a || b != b) {
            // both directions within the margin (γ ≈ 1 at the no-arbitrage price), or a NaN among the inputs
            // (both predicates are false on NaN): the reference's four forms, which propagate it
            solve_full(R1, R2, g, v1, v2, t);
            return kDirBoth;
        }
        return kDirNone;
    }
};

// ---------------------------------------------------------------------------------------------
// GeometricMeanTwoCoin -- src/cfmms.jl:180-196
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double geom_arb_delta(double m, double r1, double r2, double eta, double g)
{
    const double inner = (((g * m) * eta) * r1) * pow(r2, eta);    // :180
    return max0(pow(inner, 1.0 / (eta + 1.0)) - r2) / g;
}
__device__ __forceinline__ double geom_arb_lambda(double m, double r1, double r2, double eta, double g)
{
    const double base = (r2 * pow(r1, 1.0 / eta)) / ((eta * g) * m); // :181
    return max0(r1 - pow(base, eta / (1.0 + eta)));
}

struct GeoMeanOps {
    static constexpr bool kNeedsLogPrices = false;
    static constexpr bool kPrefetch = true;
    struct Raw {
        double2 R, w;
        double g;
        int2 ai;
        double yg;    // unused (interface of process_pool)
    };
    GeoMeanPools p;
    template <bool GBINS>
    __device__ __forceinline__ Raw load(int64_t i) const { return Raw{p.R[i], p.w[i], p.gamma[i], p.Ai[i], 0.0}; }
    template <bool GBINS, bool FAST>
    __device__ __forceinline__ void resolve(Raw&, const double2*) const {}
    __device__ __forceinline__ int2 tokens(const Raw& r) const { return r.ai; }
    // Same idea as ProductOps::solve: Δ₁,Λ₂ > 0 ⇔ γ·m₁₂·η·R₂ > R₁ and Δ₂,Λ₁ > 0 ⇔ γ·m₂₁·R₁/η > R₂
    // (the bases of :180 exceed r2^(η+1)); only the live direction's two forms (4 pow instead of
    // 8) are evaluated, with the reference's expressions on the selected operands.
    template <bool FAST>   // (no fast variant: pow dominates and the forms keep the reference's operation order)
    __device__ __forceinline__ int solve_dir(const Raw& r, const Px& px, double& d, double& l, Trade& t) const
    {
        d = l = 0.0;
        solve(r, px, t);
        return kDirBoth;
    }
    __device__ __forceinline__ void solve(const Raw& r, const Px& px, Trade& t) const
    {
        const double v1 = px.v1, v2 = px.v2;
        const double R1 = r.R.x, R2 = r.R.y, g = r.g;
        const double eta = r.w.x / r.w.y;        // :188
        const double ieta = 1.0 / eta;
        const double m12 = v2 / v1, m21 = v1 / v2;
        constexpr double kMargin = 1.0 + 1e-11;  // pow is good to ~1 ulp; keep a wide margin
        const bool p1 = (((g * m12) * eta) * R2) * kMargin >= R1;
        const bool p2 = (((g * m21) * ieta) * R1) * kMargin >= R2;
        t.d1 = t.d2 = t.l1 = t.l2 = 0.0;
        if (p1 != p2) {
            const double m = p1 ? m12 : m21, e = p1 ? eta : ieta;
            const double ra = p1 ? R2 : R1, rb = p1 ? R1 : R2;
            const double d = geom_arb_delta(m, ra, rb, e, g);    // :190 / :191
            const double l = geom_arb_lambda(m, ra, rb, e, g);   // :194 / :193
            t.d1 = p1 ? d : 0.0;
            t.d2 = p1 ? 0.0 : d;
            t.l1 = p1 ? 0.0 : l;
            t.l2 = p1 ? l : 0.0;
        } else if (p1) {
            t.d1 = geom_arb_delta(m12, R2, R1, eta, g);   // :190
            t.d2 = geom_arb_delta(m21, R1, R2, ieta, g);  // :191
            t.l1 = geom_arb_lambda(m21, R1, R2, ieta, g); // :193
            t.l2 = geom_arb_lambda(m12, R2, R1, eta, g);  // :194
        }
    }
};

// Log-space evaluation of the same two closed forms (default for GeometricMeanTwoCoin).
// With c = γ·m·e·r_a (the pow-free factor of :180), l_x = log x:
//     X = (c·r_b^e)^(1/(e+1))                    = exp((l_c + e·l_b) / (e+1))      the tendered side's new reserve
//     Y = ((r_b·r_a^(1/e)) / (e·γ·m))^(e/(1+e))  = X·r_a/c                          the received side's new reserve
// (the second identity: at the optimum the pool's marginal price equals the fee-adjusted market
// price, which fixes the RATIO of the two new reserves).  Nothing per-pool is left inside a
// logarithm: l_c = log γ + log e + log r_a + (log v_out − log v_in), so with log v staged per TOKEN
// in LDS once per block (stage_prices) and the v-independent sums prepared per POOL at upload
//     direction 1 (e = η):    exponent = (Q1 + Δ) / (η+1),       Q1 = log γ + log η + log R2 + η·log R1
//     direction 2 (e = 1/η):  exponent = (Q2 − η·Δ) / (η+1),     Q2 = η·(log γ + log R1 − log η) + log R2
// with Δ = log v2 − log v1.  Per trading pool that leaves 1 exp + 3 divisions (the exponent, Y and
// the final /γ) instead of 4 pow + 6 divisions; pools inside the
// no-arbitrage band cost four multiplies and two compares.  The exponent carries an absolute rounding
// error of a few 1e-16·max(1, η·|l|)/(η+1), so trades agree with the reference-order forms to ~1e-15 of
// the reserve scale (asserted at 1e-12 in tests/test_gpu_parity.py); unlike r2^η in the reference,
// nothing here can overflow.
struct GeoMeanLogOps {
    static constexpr bool kNeedsLogPrices = true;
    static constexpr bool kPrefetch = true;
    struct Raw {
        double2 R, Q;
        double eta, g;
        int2 ai;      // packed: {tok, gidx} until resolve()
        double yg;
    };
    GeoMeanPools p;
    template <bool GBINS>
    __device__ __forceinline__ Raw load(int64_t i) const
    {
        Raw r;
        r.R = p.R[i];
        r.Q = p.Q[i];
        r.eta = p.eta[i];
        r.yg = 0.0;
        if constexpr (GBINS) {
            r.g = p.gamma[i];
            r.ai = p.Ai[i];
        } else {
            const PackedFeeTok k = p.pk[i];
            r.ai = make_int2((int)k.tok, (int)k.gidx);
            r.g = p.gbase < 0 ? p.gamma[i] : 0.0;
        }
        return r;
    }
    template <bool GBINS, bool FAST>
    __device__ __forceinline__ void resolve(Raw& r, const double2* gtab_lds) const
    {
        if constexpr (!GBINS) {
            const unsigned tok = (unsigned)r.ai.x;
            if (p.gbase >= 0) {
                const double2 gy = gtab_lds[p.gbase + r.ai.y];
                r.g = gy.x;
                r.yg = pinned(gy.y);
            } else if constexpr (FAST) {
                r.yg = rcp_refined(r.g);
            }
            r.ai = make_int2((int)(tok & 0xffffu), (int)(tok >> 16));
        }
    }
    __device__ __forceinline__ int2 tokens(const Raw& r) const { return r.ai; }
    // one direction of the log-space forms: {d, l} for direction 1 (dir1) or 2
    template <bool FAST>
    __device__ __forceinline__ void one_direction(const Raw& r, const Px& px, bool dir1, double n, double dd, double& d, double& l) const
    {
        const double R1 = r.R.x, R2 = r.R.y, g = r.g, eta = r.eta;
        const double ra = dir1 ? R2 : R1, rb = dir1 ? R1 : R2;
        const double A = dir1 ? (r.Q.x + px.dlv) : (r.Q.y - eta * px.dlv);
        double X, Y;
        if constexpr (FAST) {   // same correctly rounded quotients for operands inside the window (checked at upload / staging)
            X = fast_exp(fast_div(A, eta + 1.0));
            Y = fast_div((X * ra) * dd, n);
            d = div_by(max0(X - rb), g, px.yg);
        } else {
            X = exp(A / (eta + 1.0));     // the tendered side's reserve after the trade
            Y = ((X * ra) * dd) / n;      // X·r_a/c, c = n/d
            d = max0(X - rb) / g;
        }
        l = max0(ra - Y);
    }
    template <bool FAST>
    __device__ __forceinline__ int solve_dir(const Raw& r, const Px& px, double& d, double& l, Trade& t) const
    {
        const double v1 = px.v1, v2 = px.v2;                 // px.dlv = log v2 − log v1
        const double R1 = r.R.x, R2 = r.R.y, g = r.g;
        const double eta = r.eta;                     // η = w₁/w₂, prepared at upload
        const double n1 = ((g * v2) * eta) * R2, d1 = v1;   // c₁ = n1/d1: direction 1 trades iff c₁ > R₁
        const double n2 = (g * v1) * R1, d2 = v2 * eta;     // c₂ = n2/d2: direction 2 trades iff c₂ > R₂
        const bool p1 = n1 > R1 * d1, p2 = n2 > R2 * d2;
        d = l = 0.0;
        if (v1 != v1 || v2 != v2) { t.d1 = t.d2 = t.l1 = t.l2 = v1 + v2; return kDirBoth; }   // NaN prices propagate (reference: pow of NaN)
        if (!(p1 || p2)) return kDirNone;
        // the (normally only) live direction
        one_direction<FAST>(r, px, p1, p1 ? n1 : n2, p1 ? d1 : d2, d, l);
        if (!(p1 && p2)) return p1 ? kDir1 : kDir2;
        // BOTH directions live (needs γ > 1): direction 2 as a second trip through the same code
        t.d1 = d;
        t.l2 = l;
        one_direction<FAST>(r, px, false, n2, d2, t.d2, t.l1);
        return kDirBoth;
    }
};

// ---------------------------------------------------------------------------------------------
// UniV3 / BoundedProduct -- src/cfmms.jl:294-395 (lane per pool, serial tick walk)
// ---------------------------------------------------------------------------------------------
// Everything compute_at_tick (:294-313) derives is independent of v, so it is evaluated ONCE at
// upload (abi_upload.cpp, same IEEE operations, hence the same bits) into the constants
// find_arb_pos (:321-337) actually uses:
//   * the CURRENT tick, visited first by both walks, as one record per pool
//       cur_a = {k, sA = R₁+α}   cur_b = {sB = R₂+β, δmax↑ = k/β − sA}   cur_c = δmax↓ = k/α − sB
//     (the flipped pool of :289 swaps sA/sB), plus curR = {R₁, R₂}, read only when the tick drains;
//   * the ticks beyond it as per-direction walk lists of the NON-EMPTY ticks only, one 64-byte record per tick
//       ks = {k, s_in}   dt = {δmax, s_out}   rout = R_out   psum = sums of the drained ticks before it
//     ("in"/"out" already flipped), and the ticks' drain thresholds as a contiguous array (solve_dir).
// A sweep then costs one division and one or two square roots for the current tick and for the one tick the walk ends
// in, whatever the number of ticks in between, empty ticks cost nothing, and a pool that trades inside its
// current tick (the common case; every BoundedProduct pool) touches only coalesced per-pool
// streams.  `initial` (:352,:374) can only be true on the current tick, and only if it is non-empty.
// HEADS: the walk consults the per-pool threshold heads (UniV3Pools::head) -- segments with walk lists.  The lean
// instantiation (BoundedProduct segments: no list anywhere, config 5) carries neither the loads nor the registers
// (same-box A/B: 21.2 vs 21.5 us HBM-resident on config5 with the head code compiled in).
template <bool HEADS>
struct UniV3OpsT {
    static constexpr bool kNeedsLogPrices = false;
    static constexpr bool kPrefetch = false;     // (measured +4 % on the multi-tick walk, +-0 on BoundedProduct segments)
    struct Raw {
        double2 pg, ca, cb;   // pg = {current_price, γ}
        double cc;
        int2 ai;              // packed: {tok, gidx} until resolve()
        int4 walk;
        uint4 hu, hd;         // threshold heads of the two walk lists (UniV3Pools::head; zeros without one)
        int64_t i;
        double yg;
    };
    UniV3Pools p;
    template <bool GBINS>
    __device__ __forceinline__ Raw load(int64_t i) const
    {
        Raw r;
        r.ca = p.cur_a[i];
        r.cb = p.cur_b[i];
        r.cc = p.cur_c[i];
        r.walk = p.has_walk ? p.walk[i] : make_int4(0, 0, 0, 0);
        r.hu = r.hd = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (HEADS) {
            if (p.head) {                                 // (kernel argument: uniform) 32 contiguous bytes per lane
                r.hu = p.head[2 * i];
                r.hd = p.head[2 * i + 1];
            }
        }
        r.i = i;
        r.yg = 0.0;
        if constexpr (GBINS) {
            r.pg = p.pg[i];
            r.ai = p.Ai[i];
        } else {   // packed record: price alone + {tokens, fee-table index} (+ the fee itself without a table)
            const PackedFeeTok k = p.pk[i];
            r.pg = make_double2(p.cp[i], p.gbase < 0 ? p.pg[i].y : 0.0);
            r.ai = make_int2((int)k.tok, (int)k.gidx);
        }
        return r;
    }
    template <bool GBINS, bool FAST>
    __device__ __forceinline__ void resolve(Raw& r, const double2* gtab_lds) const
    {
        if constexpr (!GBINS) {
            const unsigned tok = (unsigned)r.ai.x;
            if (p.gbase >= 0) {
                const double2 gy = gtab_lds[p.gbase + r.ai.y];
                r.pg.y = gy.x;
                r.yg = pinned(gy.y);
            } else if constexpr (FAST) {
                r.yg = rcp_refined(r.pg.y);
            }
            r.ai = make_int2((int)(tok & 0xffffu), (int)(tok >> 16));
        }
    }
    __device__ __forceinline__ int2 tokens(const Raw& r) const { return r.ai; }

    // find_arb_pos (:321-337) on one prepared walk-list entry; yp = rcp_refined(price) (FAST)
    template <bool FAST>
    __device__ __forceinline__ void list_tick(const TickRec& rec, double price, double yp, double& d, double& l) const
    {
        const double2 ks = rec.ks, dt = rec.dt;                    // one 64-byte line per visited tick, requested at once
        const double rout = rec.rout;
        const double dd = (FAST ? fast_sqrt(div_by(ks.x, price, yp)) : sqrt(ks.x / price)) - ks.y;   // :323
        d = 0.0;
        l = 0.0;                                                   // :325-327
        if (dd > 0) {
            if (dd >= dt.x) {                                      // :330-332
                d = dt.x;
                l = rout;
            } else {
                l = dt.y - (FAST ? fast_sqrt(price * ks.x) : sqrt(price * ks.x));   // :334
                d = dd;
            }
        }
    }

    // The part of find_arb! before the walk lists (:340-361 / :381 and the current tick): block-uniformly FAST or not.
    // Returns false when the pool does not trade (or the price is NaN: t is then all-NaN).  cur: what the current tick
    // did -- kCurEmpty (no liquidity: contributes nothing, the walk goes on, :355-358), kCurPartial ({sd, sl} set: the
    // walk ends in this tick), kCurDrained (contributes {δmax, R_out}: not loaded here, see solve_dir).
    static constexpr int kCurEmpty = 0, kCurPartial = 1, kCurDrained = 2;
    template <bool FAST>
    __device__ __forceinline__ bool head(const Raw& r, const Px& px, Trade& t, bool& up, double& g, double& yg, double& price,
                                         double& yp, double& sd, double& sl, int& cur, bool& inside) const
    {
        inside = false;
        const double cp = r.pg.x;
        g = r.pg.y;
        yg = r.yg;
        const double pr = FAST ? div_by(px.v1, px.v2, px.y2) : px.v1 / px.v2;   // :340
        t.d1 = t.d2 = t.l1 = t.l2 = 0.0;
        sd = sl = 0.0;
        up = false;
        price = 1.0;
        yp = 1.0;
        cur = kCurEmpty;
        if (pr != pr) { t.d1 = t.d2 = t.l1 = t.l2 = pr; return false; }   // NaN prices propagate instead of "no trade"
        if (g * cp <= pr && pr <= (FAST ? div_by(cp, g, yg) : cp / g)) return false;   // :347-349
        up = pr < g * cp;                                                  // :351
        if constexpr (FAST) price = up ? div_by(pr, g, yg) : fast_div(1.0, g * pr);
        else price = up ? pr / g : 1.0 / (g * pr);                         // :361 / :381
        if constexpr (FAST) yp = rcp_refined(price);
        // current tick: `initial` is true here unless the tick is empty (:355-358), so no break test
        const double k0 = r.ca.x;
        if (k0 != 0) {
            const double s_in = up ? r.ca.y : r.cb.x, s_out = up ? r.cb.x : r.ca.y;
            const double dmax = up ? r.cb.y : r.cc;
            const double dd = (FAST ? fast_sqrt(div_by(k0, price, yp)) : sqrt(k0 / price)) - s_in;   // :323
            cur = kCurPartial;
            if (dd > 0) {                                                  // :325-327
                if (dd >= dmax) {                                          // :330-332
                    cur = kCurDrained;
                } else {
                    sl = s_out - (FAST ? fast_sqrt(price * k0) : sqrt(price * k0));   // :334
                    sd = dd;
                    // the target price lies INSIDE this tick by more than a relative 2^-29: sqrt(k/price) is below its
                    // value at the tick's far boundary, s_in + δmax, by more than 2^-30 of it -- no later tick can be
                    // entered (its s_in is the square root at a boundary price beyond this one's; rounding is monotone)
                    if (p.has_walk) inside = dd < dmax - 0x1p-30 * (s_in + dmax);   // (BoundedProduct segments have no lists: nothing to skip)
                }
            }
        }
        return true;
    }

    // The walk (:353-365 / :375-385).  The reference visits tick after tick; every tick it DRAINS contributes the
    // v-independent pair {δmax, R_out}, and whether it drains one is a monotone test on the price: sqrt(k/price) − s_in
    // > 0 and >= δmax holds for every price up to a threshold T and for none above it (division, square root and
    // subtraction are correctly rounded, hence monotone).  So the upload stores, per walk list in walk order,
    //   * the thresholds T_j CONTIGUOUSLY (p.thr: 8 ticks per 64-byte line; found by bisection on the reference's own
    //     floating-point test, so `price <= T_j` IS that test), a 0 closing every list, and
    //   * in tick j's record the sums {Σδ, Σλ} of everything BEFORE it -- current tick and ticks 0..j−1 all drained --
    //     accumulated in the walk's order with the walk's operations (same bits as walking), plus a closing record
    //     per list that carries the sums of the whole list.
    // A pool whose current tick drains (or is empty) then costs one scan of its thresholds and ONE record, however deep
    // it walks, instead of one scattered 64-byte line per visited tick; from the first tick that is not drained the
    // reference's tick-by-tick evaluation takes over (one partially filled tick, normally), and it goes on to the
    // following tick only inside a 2^-40 band around that tick's threshold -- outside it the next tick cannot be
    // entered: its s_in is sqrt(k/p⁺) with p⁺ <= this tick's far boundary < price, and rounding is monotone.  A pool
    // that ends inside its current tick (kCurPartial) reads no list at all (round 4; `inside` in head()) unless its
    // target price is within 2^-29 of the tick's far boundary, where the plain walk decides (it stops at the first list tick).
    template <bool FAST>
    __device__ __forceinline__ int solve_dir(const Raw& r, const Px& px, double& d, double& l, Trade& t) const
    {
        bool up, inside;
        int cur;
        double g, yg, price, yp, sd, sl;
        d = l = 0.0;
        if (!head<FAST>(r, px, t, up, g, yg, price, yp, sd, sl, cur, inside)) return t.d1 != t.d1 ? kDirBoth : kDirNone;   // (NaN price: t is all-NaN)
        const int begin = up ? r.walk.x : r.walk.z;
        // a pool that ends well inside its current tick (the common case) touches no list at all; the reference's next
        // find_arb_pos would return zeros and break (:363-365).  Within 2^-29 of the far boundary the plain walk decides.
        const int count = inside ? 0 : (up ? r.walk.y : r.walk.w);
        int j = 0;
        const bool jump = cur != kCurPartial && count > 0;
        TickRec rec;
        double thr_j = 0.0;                                                // T[j] of the tick the scan stopped at (TickRec::thr)
        bool have = false;
        if (jump) {
            // ticks 0..j−1 drain.  Round 5: the first four list ticks are decided from the pool's threshold head -- binary32
            // values rounded DOWN at upload, so with lo = the float and hi = the next float up (lo <= T <= hi):
            // price <= lo PROVES the tick drains, price > hi PROVES it does not; a price in between (a relative 2^-23 band
            // around a threshold), a threshold outside the float range (NaN pattern) or a walk deeper than four ticks falls
            // back to the exact array below.  Same decisions as the exact scan, without touching thr[] for ~4 of 5 walkers.
            bool exact = true;                                             // the exact scan (still) has to run, from tick j
            // T[j] of the tick the scan stops at, for the band test below -- from the scan's own registers, NOT from the record
            // (measured: taking it from rec.thr makes the walk wait for the record earlier, 44.4 -> 49.2 us with the heads off).
            // The head path knows only hi >= T[j]: good enough, the test "price > T[j](1 + 2^-40)" is then merely conservative
            // (a price inside (T[j], hi](1 + 2^-40) visits one more tick, which returns zeros and ends the walk the same way).
            double thr_t = 0.0;
            if (HEADS && p.head) {
                const uint4 h = up ? r.hu : r.hd;
                const unsigned hb[4] = {h.x, h.y, h.z, h.w};
                int adv = 0;
                bool stop = false, amb = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double lo = (double)__uint_as_float(hb[k]), hi = (double)__uint_as_float(hb[k] + 1u);
                    const bool go = !stop && !amb;
                    const bool drains = price <= lo, not_drains = hb[k] == 0u || price > hi;   // (0 = never; NaN: neither)
                    adv += (go && drains) ? 1 : 0;
                    thr_t = (go && !drains && not_drains) ? (hb[k] == 0u ? 0.0 : hi) : thr_t;
                    stop = stop || (go && !drains && not_drains);
                    amb = amb || (go && !drains && !not_drains);
                }
                if (!amb) {
                    j = adv;
                    exact = !stop && j < count;        // all four drain and the list goes on: the exact scan continues at tick 4
                }
            }
            if (exact) {
                // Four thresholds per round trip (the dependent chain of a walking pool is
                // walk span -> thresholds -> one record); every list ends in a 0 and the array is padded, so reading past
                // a short list is harmless and the first failing test ends the scan exactly like one-by-one.
                const double* T = p.thr + begin;
                for (;;) {
                    const double t0 = T[j], t1 = T[j + 1], t2 = T[j + 2], t3 = T[j + 3];
                    const int adv = !(price <= t0) ? 0 : !(price <= t1) ? 1 : !(price <= t2) ? 2 : !(price <= t3) ? 3 : 4;
                    thr_t = adv == 0 ? t0 : adv == 1 ? t1 : adv == 2 ? t2 : t3;
                    j += adv;
                    if (adv < 4 || j >= count) break;
                }
            }
            // (prices <= 0 from a caller's device vector pass every test, the closing 0 included: the closing record, whose
            //  threshold is 0 whichever path found it -- ADVICE r5)
            thr_t = j < count ? thr_t : 0.0;
            j = j < count ? j : count;
            rec = p.ticks[begin + j];                                      // (j == count: the list's closing record)
            thr_j = thr_t;                                                 // T[j] (or the float just above it) of the tick the scan stopped at
            have = true;
            sd = rec.psum.x;
            sl = rec.psum.y;
        } else if (cur == kCurDrained) {
            const double2 R = p.curR[r.i];                                 // no list in this direction: the current tick alone
            sd = up ? r.cb.y : r.cc;
            sl = up ? R.y : R.x;
        }
        for (; j < count; ++j) {                                           // :353 / :375, empty ticks elided
            double dj, lj;
            if (!have) {
                rec = p.ticks[begin + j];
                thr_j = rec.thr;
            }
            have = false;
            list_tick<FAST>(rec, price, yp, dj, lj);
            if (dj == 0 || lj == 0) break;                                 // :363-365 (initial is false here)
            sd += dj;
            sl += lj;
            if (jump && price > thr_j * (1.0 + 0x1p-40)) break;            // the next tick cannot be entered (see above)
        }
        d = FAST ? div_by_signed_zero(sd, g, yg) : sd / g;                 // :366-372 / :386-391
        l = sl;
        return up ? kDir1 : kDir2;
    }
};

using UniV3Ops = UniV3OpsT<true>;
using UniV3OpsLean = UniV3OpsT<false>;

// ---------------------------------------------------------------------------------------------
// The sweep: src/router.jl:38-42 fused with :79-83 and :98-100
// ---------------------------------------------------------------------------------------------
// Trade stores (Δ, Λ: 16 B per lane, written once, never re-read by the sweep) are write-through (`sc1`): the bytes
// leave during the sweep instead of in a serial flush of dirty L2 lines at its end (MI355X_MICROARCH.md, boundary row:
// + B / 6 TB/s for B dirty bytes; measured against plain and non-temporal stores in rounds 1 and 2: -1..-2 us per
// 1M-pool sweep).
// The s_nop is part of the store: a VMEM store of more than 8 bytes reads its upper data registers one cycle after
// issue, and the VALU must not overwrite them in that cycle (gfx9 hazard "VMEM store > 8 bytes followed by a write of
// the VGPRs holding the write data: 1 wait state").  The compiler inserts that wait state for its own stores; it
// cannot see through inline asm -- and the values stored here are selects computed right before the next store, so the
// next v_cndmask may land in the registers of this one (found as 9 % wrong rows in the two-row trade layout).
typedef double d2v __attribute__((ext_vector_type(2)));
// nt (SweepArgs::nt_stores, launch-uniform): non-temporal instead of write-through -- for markets whose pool state comes from
// HBM on every sweep (too large for the Infinity Cache, or a caller that rotates over many markets): the trade lines then
// bypass the cache hierarchy instead of displacing pool state that is about to be read.
__device__ __forceinline__ void store_pair(double2* dst, double x, double y, int nt)
{
    d2v val = {x, y};
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(dst), "v"(val) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(val) : "memory");
}

// Partial rows are written through as well: no dirty line is left for the end-of-kernel release in front of the fold
// launch (A/B: -0.3..-0.45 us per step on single-family launches, +-0.1 on config3; profiles/r03_ab_row_write_through.txt).
__device__ __forceinline__ void store_row(double* dst, double x)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(dst), "v"(x) : "memory");
}

// One block's share of a segment: tiles bid, bid+nblocks, ... of the segment's pools; its
// partial row goes to partials[row].
// GBINS = true is the large-market mode (n_tokens > kMaxLdsTokens, v and the bins no longer fit
// LDS): v is gathered straight from global memory (it stays L2-resident) and every pool's two
// flows Λ−Δ are written to a flow array; Ψ is then PULLED per token over a token -> (pool, side)
// incidence list built at upload (gather_chunks / token_fold below) -- no float atomics, fixed
// summation order.  The partial rows carry only the dual scalar.
// LDS of a sweeping block: the prices {v, rcp_refined(v)} per token, the launch's fee table {γ, rcp_refined(γ)},
// log v (launches with a log-space GeometricMean segment), the netflow bins (one copy per wavefront, or one
// shared copy) and one slot per wavefront for the dual-scalar fold.
struct SweepLds {
    double2* vy;       // [n_pad] {v, rcp_refined(v)}  (a.v_shift == 3: [n_pad] doubles, the prices alone)
    double2* gtab;     // [gtab_n] {γ, rcp_refined(γ)}
    double* lv;        // [n_pad] log v (only when a.need_logv)
    double* bins;      // [copies][n_pad]
    double* wsum;      // [kWaves] dual-scalar fold
    double* flags;     // [kWaves + 1] stage_prices: per-wavefront "prices in the fast window", then "launch is live"
    double* my_bins;   // this wavefront's copy
};

template <int BLOCK, bool GBINS>
__device__ __forceinline__ SweepLds carve_lds(const SweepArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    SweepLds L;
    if constexpr (GBINS) {
        L.vy = nullptr;
        L.gtab = nullptr;
        L.lv = L.bins = L.my_bins = L.flags = nullptr;
        L.wsum = lds;
        return L;
    }
    L.vy = reinterpret_cast<double2*>(lds);
    L.gtab = reinterpret_cast<double2*>(lds + ((size_t)a.n_pad << (a.v_shift - 3)));
    L.lv = reinterpret_cast<double*>(L.gtab + a.gtab_n);
    L.bins = L.lv + (a.need_logv ? a.n_pad : 0);
    L.wsum = L.bins + (size_t)a.copies * a.n_pad;
    L.flags = L.wsum + BLOCK / 64;
    L.my_bins = L.bins + (size_t)(a.copies == 1 ? 0 : (threadIdx.x >> 6)) * a.n_pad;
    return L;
}

// Pre-armed launch: one lane waits for the host's word (see SweepArgs::arm_word).  1 = v is in place, 0 = the launch is
// cancelled -- this sequence number with the cancel bit, or ANY later sequence number: the host has moved on (it
// overwrites the one word while blocks of an abandoned launch may still be starting) --, -1 = gave up waiting.
__device__ __forceinline__ int wait_armed(const SweepArgs& a)
{
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        const unsigned long long w = __hip_atomic_load(a.arm_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (w == a.arm_seq) return 1;
        if ((w & ~kArmCancel) > a.arm_seq || w == (a.arm_seq | kArmCancel)) return 0;
        if ((long long)wall_clock64() - t0 > a.arm_timeout) return -1;
        __builtin_amdgcn_s_sleep(1);
    }
}

// A block reports WHY it poisons its row (NaN in every column): one sticky word in mapped host memory, read by the
// host when it meets a non-finite result (abi_sweep.cpp).  Rare paths only.
__device__ __forceinline__ void report(const SweepArgs& a, unsigned long long bit)
{
    if (a.flags && threadIdx.x == 0) __hip_atomic_fetch_or(a.flags, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bins <- 0, fee table and prices -> LDS, barrier.  Whole block.  Result (block-uniform): kStageLive unless this is a
// pre-armed launch that was cancelled or gave up waiting for its prices (then also kStageGaveUp); kStageFast when every
// staged price lies in the window of the fast arithmetic.
constexpr int kStageLive = 1, kStageFast = 2, kStageGaveUp = 4;
template <int BLOCK, bool GBINS>
__device__ __forceinline__ int stage_prices(const SweepArgs& a, const SweepLds& L)
{
    if constexpr (GBINS) {
        __syncthreads();
        return kStageLive;
    }
    const int tid = threadIdx.x;
    const int n_zero = a.copies * a.n_pad;                // LDS bins to clear
    const bool logs = a.need_logv != 0;
    const bool armed = a.arm_word != nullptr;             // kernel argument: block-uniform
    bool live = true, gave_up = false;
    if (armed) {
        // pre-armed launch: everything that does not need the prices first, then the wait
        for (int j = tid; j < n_zero; j += BLOCK) L.bins[j] = 0.0;
        for (int j = tid; j < a.gtab_n; j += BLOCK) {
            const double g = a.gtab[j];
            L.gtab[j] = make_double2(g, rcp_refined(g));
        }
        if (tid == 0) L.flags[BLOCK / 64] = (double)wait_armed(a);
        __syncthreads();
        live = L.flags[BLOCK / 64] > 0.0;
        gave_up = L.flags[BLOCK / 64] < 0.0;
    }
    bool in_window = true;
    for (int j = tid; j < a.n; j += BLOCK) {
        // armed: the host wrote v through the PCIe BAR after this kernel may have started -- system-scope loads
        const double vj = armed ? __hip_atomic_load(a.v + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : a.v[j];
        if (a.v_shift == 4) L.vy[j] = make_double2(vj, rcp_refined(vj));
        else reinterpret_cast<double*>(L.vy)[j] = vj;
        in_window = in_window && in_fast_window(vj);
        if (logs) L.lv[j] = log(vj);     // once per token per block: the only logarithm of a GeometricMean sweep
    }
    if (!armed) {
        // the price loads are in flight while the bins are cleared
        for (int j = tid; j < n_zero; j += BLOCK) L.bins[j] = 0.0;
        for (int j = tid; j < a.gtab_n; j += BLOCK) {
            const double g = a.gtab[j];
            L.gtab[j] = make_double2(g, rcp_refined(g));
        }
    }
    // block-wide AND of in_window through LDS (the library's __syncthreads_and would add 256 bytes of STATIC LDS to
    // every kernel, which the 160 KiB dynamic ceiling of hipFuncSetAttribute then no longer leaves room for)
    const bool wave_in = __all(in_window ? 1 : 0) != 0;
    if ((tid & 63) == 0) L.flags[tid >> 6] = wave_in ? 1.0 : 0.0;
    __syncthreads();
    bool all_in = true;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) all_in = all_in && L.flags[w] != 0.0;
    return (live ? kStageLive : 0) | (all_in ? kStageFast : 0) | (gave_up ? kStageGaveUp : 0);
}

// One pool: prices from LDS, closed form, trade record, dual scalar, netflow bins.
template <class Ops, bool MAT, bool GBINS, bool FAST>
__device__ __forceinline__ void process_pool(const Ops& ops, const SweepArgs& a, const SweepLds& L,
                                             const typename Ops::Raw& raw_in, int64_t i, double& acc)
{
    typename Ops::Raw raw = raw_in;
    ops.template resolve<GBINS, FAST>(raw, L.gtab);   // packed records: {tokens, fee-table index} -> tokens, fee
    const int2 tok = ops.tokens(raw);
    Px px;                                            // v[r.cfmms[i].Ai]
    px.yg = raw.yg;
    if constexpr (GBINS) {
        px.v1 = a.v[tok.x];
        px.v2 = a.v[tok.y];
        px.y1 = px.y2 = 0.0;
        px.dlv = Ops::kNeedsLogPrices ? log(px.v2 / px.v1) : 0.0;   // large markets: v is not staged, one logarithm per pool
    } else {
        if constexpr (FAST) {   // {v, rcp_refined(v)} pairs (FAST implies a.stage_y)
            const double2 a1 = L.vy[tok.x], a2 = L.vy[tok.y];
            px.v1 = a1.x; px.y1 = a1.y;
            px.v2 = a2.x; px.y2 = a2.y;
        } else {                // pairs, or -- markets too wide for them -- the prices alone (a.v_shift = 4 / 3)
            const char* base = reinterpret_cast<const char*>(L.vy);
            px.v1 = *reinterpret_cast<const double*>(base + ((size_t)tok.x << a.v_shift));
            px.v2 = *reinterpret_cast<const double*>(base + ((size_t)tok.y << a.v_shift));
            px.y1 = px.y2 = 0.0;
        }
        px.dlv = 0.0;
        if constexpr (Ops::kNeedsLogPrices)   // log v staged per token, or -- markets too wide for that row -- one logarithm per pool
            px.dlv = a.need_logv ? L.lv[tok.y] - L.lv[tok.x] : log(px.v2 / px.v1);
    }
    Trade t;
    double d, l;
    int dir;
    dir = ops.template solve_dir<FAST>(raw, px, d, l, t);
    const double v1 = px.v1, v2 = px.v2;
    // One direction (the common case): the tendered token's bin gets −d, the received token's +l, the dual scalar
    // l·v_out − d·v_in -- the reference's expressions (src/router.jl:82, :99) with their zero terms dropped: same bits.
    // A compact trade record needs two values with a clear sign bit; the tiny negative / −0.0 amounts the reference's
    // tick arithmetic leaves on degenerate UniV3 boundaries (and anything kDirBoth) take the general path below.
    const bool one = dir == kDir1 || dir == kDir2;
    const bool plain = one && (!MAT || !a.compact || (__double2hiint(d) | __double2hiint(l)) >= 0);
    if (plain) {
        const bool d1 = dir == kDir1;
        if (MAT) {
            if (a.compact) {
                store_pair(a.Delta + i, d1 ? d : -d, l, a.nt_stores);   // {+Δ₁, Λ₂} or {−Δ₂, Λ₁}: the sign bit carries the direction
            } else {
                store_pair(a.Delta + i, d1 ? d : 0.0, d1 ? 0.0 : d, a.nt_stores);
                store_pair(a.Lambda + i, d1 ? 0.0 : l, d1 ? l : 0.0, a.nt_stores);
            }
        }
        acc += l * (d1 ? v2 : v1) - d * (d1 ? v1 : v2);
        if constexpr (GBINS) {
            a.gflow[i] = d1 ? make_double2(0.0 - d, l) : make_double2(l, 0.0 - d);
        } else {
            const double f_in = 0.0 - d;                       // Λ − Δ of the tendered token
            if (f_in != 0.0) atomicAdd(&L.my_bins[d1 ? tok.x : tok.y], f_in);   // ds_add_f64
            if (l != 0.0) atomicAdd(&L.my_bins[d1 ? tok.y : tok.x], l);
        }
        return;
    }
    if (dir == kDirNone) {
        if (MAT) {
            if (a.compact) {
                store_pair(a.Delta + i, 0.0, 0.0, a.nt_stores);
            } else {
                store_pair(a.Delta + i, 0.0, 0.0, a.nt_stores);
                store_pair(a.Lambda + i, 0.0, 0.0, a.nt_stores);
            }
        }
        if constexpr (GBINS) a.gflow[i] = make_double2(0.0, 0.0);
        return;
    }
    expand_dir(dir, d, l, t);
    if (MAT) {
        if (a.compact) {   // one 16-byte record per pool (see SweepArgs)
            // a record can carry one direction whose two values have a clear sign bit (NaN payloads survive the
            // sign flip) while the other direction is exactly +0; everything else -- both directions trading and
            // the tiny negative / -0.0 values the reference's tick arithmetic produces on degenerate UniV3
            // boundaries -- takes the overflow rows, so the encoding is lossless bit for bit
            const int d1h = __double2hiint(t.d1), d2h = __double2hiint(t.d2), l1h = __double2hiint(t.l1), l2h = __double2hiint(t.l2);
            const int z1 = d2h | __double2loint(t.d2) | l1h | __double2loint(t.l1);   // 0 <=> Δ₂ and Λ₁ are +0.0
            const int z2 = d1h | __double2loint(t.d1) | l2h | __double2loint(t.l2);   // 0 <=> Δ₁ and Λ₂ are +0.0
            const bool dir1 = (z1 == 0) & ((d1h | l2h) >= 0);
            const bool dir2 = (z2 == 0) & ((d2h | l1h) >= 0);
            double ra = t.d1, rb = t.l2;
            if (!dir1) {
                if (dir2) {
                    ra = -t.d2;          // sign bit set (−0.0 included): direction 2
                    rb = t.l1;
                } else {
                    a.Lambda[i] = make_double2(t.d1, t.d2);
                    a.Over[i] = make_double2(t.l1, t.l2);
                    ra = 0.0;
                    rb = -1.0;
                }
            }
            store_pair(a.Delta + i, ra, rb, a.nt_stores);
        } else {
            store_pair(a.Delta + i, t.d1, t.d2, a.nt_stores);
            store_pair(a.Lambda + i, t.l1, t.l2, a.nt_stores);
        }
    }
    // src/router.jl:82  dot(Λ, v[Ai]) - dot(Δ, v[Ai])
    acc += (t.l1 * v1 + t.l2 * v2) - (t.d1 * v1 + t.d2 * v2);
    // src/router.jl:99 / :115  G[Ai] .+= Λ .- Δ
    const double f1 = t.l1 - t.d1, f2 = t.l2 - t.d2;
    if constexpr (GBINS) {
        a.gflow[i] = make_double2(f1, f2);
    } else {
        if (f1 != 0.0) atomicAdd(&L.my_bins[tok.x], f1);   // ds_add_f64
        if (f2 != 0.0) atomicAdd(&L.my_bins[tok.y], f2);
    }
}

// The tile loop of one pool family over a block's share of a segment: lane tid takes pools
// (bid + k·nblocks)·BLOCK + tid, k = 0, 1, ...  Returns this lane's dual-scalar part.
// The first tile's pool state is requested BEFORE v and the bins are staged in LDS, so that HBM round trip is not
// exposed behind the staging barrier.  Tile order alternates between consecutive sweeps (a.reverse): block-strided
// "phases" are walked first-to-last by one sweep and last-to-first by the next, so each sweep begins on the pool data
// the previous one touched last -- the part that is still in the XCD's 4 MB L2 (a forward-only walk over a
// 5.5 MB-per-XCD working set is the LRU worst case: 0 % hits; measured on a plain read stream of the
// same 44 MB: 9.6 -> 6.9 us, profiles/r02_launch_floor.txt).
// Requesting tile k+1 before tile k is solved (Ops::kPrefetch, single-family launches): +-0.1 us cache-warm, -4..-6 %
// HBM-resident on the two-coin families at +6..8 VGPRs (round 3 measured the same and dropped it for the registers; since round 4's per-kernel
// arithmetic they are free: profiles/r04_ab_prefetch_stores.txt); +4 % on the multi-tick UniV3 walk, which keeps the plain loop.
template <class Ops, bool MAT, int BLOCK, bool GBINS, bool FAST, bool PREFETCH>
__device__ __forceinline__ void tile_loop(const Ops& ops, const SweepArgs& a, const SweepLds& L, typename Ops::Raw cur,
                                          int64_t i, int64_t step, int64_t left, double& acc)
{
    if constexpr (PREFETCH) {
        // tile k+1's pool state is requested before tile k is solved: two tiles' loads in flight per lane
        bool ok = left > 0;
        while (ok) {
            const bool more = left > 1;
            typename Ops::Raw nxt = cur;
            if (more) nxt = ops.template load<GBINS>(i + step);
            process_pool<Ops, MAT, GBINS, FAST>(ops, a, L, cur, i, acc);
            i += step;
            --left;
            ok = more;
            cur = nxt;
        }
    } else {
        bool ok = left > 0;
        while (ok) {
            process_pool<Ops, MAT, GBINS, FAST>(ops, a, L, cur, i, acc);
            i += step;
            ok = --left > 0;
            if (ok) cur = ops.template load<GBINS>(i);
        }
    }
}

// FASTK selects the arithmetic of the WHOLE kernel (round 4).  Rounds 1-3 compiled both tile loops into every kernel and
// chose per block; the two copies cost 12-36 VGPRs over the larger of the two alone (ProductTwoCoin 70 vs 56 / 58,
// UniV3 84 vs 63 / 63, the fused materialising launch 116 vs 80 / 97: profiles/r04_kernel_resources.txt), i.e. one to
// three wavefronts per SIMD.  Now the HOST picks the kernel wherever it can:
//   kArithFull  the compiler's full-range sequences (fast_math = 0, pool constants or prices outside the window);
//   kArithFast  the fast arithmetic: every pool constant of the launch is inside the window (Segment::fast_ok, checked at
//               upload) and so are the prices -- which the host KNOWS (host-pointer sweeps, cfmm_route).  Every block still
//               checks the prices it stages; outside the window it does NOT compute: it poisons its row (NaN in every column
//               -- an error, never a wrong number) and reports kFlagWindow (reachable only through a pre-armed launch whose
//               host-side check raced; abi_sweep.cpp cancels such launches before they run);
//   kArithAuto  round 5, device-pointer sweeps (cfmm_sweep_dev): the library cannot see these prices, so the kernel carries
//               BOTH tile loops and every block chooses from the prices it staged (block-uniform, and the same choice in every
//               block: all stage the same vector) -- prices outside the window, NaN included, take the full-range loop and
//               behave like the reference's arithmetic, instead of round 4's "refuse, report on a later call" protocol
//               (ADVICE r4, medium).  Costs the registers of rounds 1-3 again, which the default geometry (4 wavefronts per
//               SIMD) never needed.
constexpr int kArithFull = 0, kArithFast = 1, kArithAuto = 2;
template <class Ops, bool MAT, int BLOCK, bool GBINS, int FASTK, bool MULTI>
__device__ __forceinline__ double sweep_tiles(const Ops& ops, const SweepArgs& a, const SweepLds& L, int bid, int nblocks,
                                              bool& poison, bool& live)
{
    static_assert(!(GBINS && FASTK != kArithFull), "large-market mode runs on the compiler's sequences");
    double acc = 0.0;
    const int64_t stride = (int64_t)nblocks * BLOCK;
    const int64_t i0 = (int64_t)bid * BLOCK + threadIdx.x;
    int64_t left = i0 < a.m ? (a.m - i0 + stride - 1) / stride : 0;      // tiles of this lane
    const int64_t step = a.reverse ? -stride : stride;
    const int64_t i = a.reverse ? i0 + (left - 1) * stride : i0;
    typename Ops::Raw cur = {};
    if (left > 0) cur = ops.template load<GBINS>(i);
    const int staged = stage_prices<BLOCK, GBINS>(a, L);
    poison = (staged & kStageLive) == 0;              // a pre-armed launch that is not needed (or gave up)
    live = !poison || (staged & kStageGaveUp) != 0;   // false: CANCELLED by the host (a launch that gave up waiting still reports NaN)
    if (staged & kStageGaveUp) report(a, kFlagGaveUp);
    if (FASTK == kArithFast && !poison && (staged & kStageFast) == 0) {
        poison = true;                                // prices outside the window of this kernel's arithmetic
        report(a, kFlagWindow);
    }
    if (poison) left = 0;
    // next-tile prefetch: single-family launches of the two-coin families (HBM-resident -4..-6 %, cache-warm +-0.1 us); the
    // fused multi-family launch keeps the plain loop (same-box A/B on config3: -2 % HBM-resident but +1.5 % on the warm step)
    constexpr bool kPre = Ops::kPrefetch && !MULTI;
    if constexpr (FASTK == kArithAuto) {
        if (staged & kStageFast) tile_loop<Ops, MAT, BLOCK, GBINS, true, kPre>(ops, a, L, cur, i, step, left, acc);
        else tile_loop<Ops, MAT, BLOCK, GBINS, false, kPre>(ops, a, L, cur, i, step, left, acc);
    } else {
        tile_loop<Ops, MAT, BLOCK, GBINS, FASTK == kArithFast, kPre>(ops, a, L, cur, i, step, left, acc);
    }
    return acc;
}

// Block epilogue: fold the dual scalar (lanes by wave shuffles, waves through LDS, fixed order), fold
// the bin copies in a fixed order and write the block's partial row.
// Column j of a single-block launch's result, straight to its consumer (SweepArgs::direct): a plain store for device
// consumers, or the column's two self-validating granules {tag, 32 bits} as ONE 16-byte system-scope store -- adjacent lanes
// write adjacent columns, so a wavefront's store covers full 64-byte lines on the PCIe side (see fold_finish).
typedef unsigned long long u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void publish_column(const SweepArgs& a, int j, double val)
{
    if (a.direct_host.gran) {
        const unsigned long long tag = (a.direct_host.tag & 0xffffffffull) << 32, u = (unsigned long long)__double_as_longlong(val);
        u2v g = {tag | (u & 0xffffffffull), tag | (u >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(a.direct_host.gran + 2 * (size_t)j), "v"(g) : "memory");
    } else {
        a.direct_out[j] = val;
    }
}

template <int BLOCK, bool GBINS>
__device__ __forceinline__ void finish_row(const SweepArgs& a, const SweepLds& L, double acc, int row_id, bool poison, bool publish)
{
    const double nan = __builtin_nan("");
    if (poison) acc = nan;                               // every column of a poisoned row is NaN: whoever folds it sees an error
    constexpr int kWaves = BLOCK / 64;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) L.wsum[wave] = acc;
    __syncthreads();

    const int n_cols = GBINS ? 0 : a.n;                  // Ψ columns of the partial row
    if (!GBINS && a.direct) {
        // the only block of its launch: the row is the result.  A cancelled pre-armed launch publishes nothing (the fold of
        // such an evaluation returns without output as well: the host has moved on and may reuse the sequence tag)
        if (!publish) return;
        for (int j = tid; j < n_cols; j += BLOCK) {
            double s = L.bins[j];
            for (int c = 1; c < a.copies; ++c) s += L.bins[(size_t)c * a.n_pad + j];
            publish_column(a, j, poison ? nan : s);
        }
        if (tid == 0) {
            double s = L.wsum[0];
            for (int w = 1; w < kWaves; ++w) s += L.wsum[w];
            publish_column(a, n_cols, poison ? nan : s);
        }
        return;
    }
    double* row = a.partials + (size_t)row_id * a.row_pitch;   // 128-byte aligned rows (SweepArgs::row_pitch)
    for (int j = tid; j < n_cols; j += BLOCK) {
        double s = L.bins[j];
        for (int c = 1; c < a.copies; ++c) s += L.bins[(size_t)c * a.n_pad + j];
        store_row(row + j, poison ? nan : s);
    }
    if (tid == 0) {
        double s = L.wsum[0];
        for (int w = 1; w < kWaves; ++w) s += L.wsum[w];
        store_row(row + n_cols, poison ? nan : s);
    }
}

// One block's share of ONE segment: tiles bid, bid+nblocks, ... of the segment's pools; its partial
// row goes to partials[row_id].
template <class Ops, bool MAT, int BLOCK, bool GBINS, int FASTK, bool MULTI>
__device__ __forceinline__ void sweep_body(const Ops& ops, const SweepArgs& a, int bid, int nblocks, int row_id)
{
    const SweepLds L = carve_lds<BLOCK, GBINS>(a);
    bool poison, live;
    const double acc = sweep_tiles<Ops, MAT, BLOCK, GBINS, FASTK, MULTI>(ops, a, L, bid, nblocks, poison, live);
    finish_row<BLOCK, GBINS>(a, L, acc, row_id, poison, live);
}

template <class Ops, bool MAT, int BLOCK, bool GBINS, int FASTK>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(Ops ops, SweepArgs a)
{
    sweep_body<Ops, MAT, BLOCK, GBINS, FASTK, false>(ops, a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.x);
}

// Several segments (pool families) in ONE launch, so HBM-bound ProductTwoCoin blocks and ALU-bound GeometricMean /
// UniV3 blocks are co-resident on every CU and overlap, and the sweep pays one launch + one kernel boundary instead of nseg.
template <bool MAT, int BLOCK, bool GBINS, int FASTK>
__global__ __launch_bounds__(BLOCK) void sweep_multi(MultiArgs ma)
{
    const int bidx = (int)blockIdx.x;
    int sidx, local, nblocks;
    if (ma.xcd_map) {
        // XCD-aware, cost-weighted block -> segment map.  Blocks are dealt round-robin to the 8 XCDs (block
        // b runs on XCD b % 8), so "segment = b % nseg" would put ALL blocks of one pool family on the same
        // XCDs (nseg = 2: ProductTwoCoin on XCDs 0,2,4,6, GeometricMean on 1,3,5,7 -- half the chip does all
        // the arithmetic).  Here the 8 blocks of one deal (one per XCD) share a segment, consecutive deals
        // walk through a 32-entry pattern in which segment s appears seg_w[s] times (its share of the
        // launch's work: pools x cost per pool, so that all blocks finish together), and the pattern is
        // rotated by one every 32 deals (one pass over an XCD's 32 CUs), so every XCD -- and, as the
        // dispatcher fills CUs in order, every CU -- hosts all families.  Placement is only a performance
        // assumption: any placement computes the same result.
        const int x = bidx & 7, j = bidx >> 3, q = j >> 5, p = (j + q) & 31;
        sidx = ma.pattern[p];
        const int w = ma.seg_w[sidx];
        local = (q * w + ma.rank[p]) * 8 + x;
        nblocks = ((int)gridDim.x >> 8) * w * 8;
    } else {   // small grids (not a multiple of 256 blocks): block b -> segment b % nseg
        nblocks = (int)gridDim.x / ma.nseg;
        sidx = bidx % ma.nseg;
        local = bidx / ma.nseg;
    }
    const MultiSeg& sg = ma.seg[sidx];
    SweepArgs a = ma.common;
    a.m = sg.m;
    a.Delta = sg.Delta;
    a.Lambda = sg.Lambda;
    a.Over = sg.Over;
    a.gflow = sg.gflow;
    switch (sg.kind) {
    case 0:
        sweep_body<ProductOps, MAT, BLOCK, GBINS, FASTK, true>(ProductOps{sg.pools.p}, a, local, nblocks, bidx);
        break;
    case 1: // log-space forms only; geomean_exact routers are swept by per-segment launches
        sweep_body<GeoMeanLogOps, MAT, BLOCK, GBINS, FASTK, true>(GeoMeanLogOps{sg.pools.g}, a, local, nblocks, bidx);
        break;
    default:
        sweep_body<UniV3Ops, MAT, BLOCK, GBINS, FASTK, true>(UniV3Ops{sg.pools.u}, a, local, nblocks, bidx);
        break;
    }
}

// ---------------------------------------------------------------------------------------------
// Row fold: out[j] = sum over rows of partials[row][j]  (src/router.jl:81-83, :98-100 summed over blocks)
// ---------------------------------------------------------------------------------------------
// One block owns kReduceCols adjacent columns (64 B = half a 128-byte line of every row; rows are 128-byte aligned,
// SweepArgs::row_pitch).  lane = (row-lane r,
// column c): a wavefront holds 8 row-lanes x 8 columns.  Each lane sums its rows in increasing
// order (kBatch independent loads in flight), the row-lanes of a wavefront are folded by a fixed
// shuffle tree, the wavefronts by a fixed-order LDS pass: bit-reproducible for a fixed geometry.
//
// Block -> column group.  The two column groups of one 128-byte line are folded by two blocks; blocks are dealt round-robin
// to the 8 XCDs (block b runs on XCD b % 8, each with its own L2), so with "column group = block index" every line of the
// partial rows was fetched from the fabric TWICE, by two different L2s (round 4: 1491 KiB per fold for 526 KB of rows,
// together with rows that were not line-aligned).  Here the pair of groups {2p, 2p+1} belongs to blocks b = x + 8·(2j) and
// x + 8·(2j+1) with p = x + 8j: same XCD, consecutive deals -- the second request of a line is served by that XCD's L2
// (or merged with the first in flight).  Grid = 16·ceil(pairs / 8) blocks; a block beyond the last group returns.
// Placement is only a performance assumption: any placement computes the same result.
__device__ __forceinline__ int fold_colblock(int b)
{
    const int x = b & 7, q = b >> 3;
    return 2 * (x + 8 * (q >> 1)) + (q & 1);
}
static int fold_grid(int n1)
{
    const int groups = (n1 + kReduceCols - 1) / kReduceCols, pairs = (groups + 1) / 2;
    return 16 * ((pairs + 7) / 8);
}

__device__ __forceinline__ double fold_columns(const double* __restrict__ partials, int rows, int n1, int pitch, int colblock, double* red)
{
    constexpr int kRowLanes = kFoldBlock / kReduceCols;
    constexpr int kWaves = kFoldBlock / 64;
    constexpr int kBatch = 4;
    const int c = threadIdx.x % kReduceCols;
    const int r = threadIdx.x / kReduceCols;
    const int col = colblock * kReduceCols + c;
    double s = 0.0;
    if (col < n1) {
        const double* p = partials + col;
        int row = r;
        for (; row + (kBatch - 1) * kRowLanes < rows; row += kBatch * kRowLanes) {
            double x[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) x[b] = p[(size_t)(row + b * kRowLanes) * pitch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) s += x[b];
        }
        for (; row < rows; row += kRowLanes) s += p[(size_t)row * pitch];
    }
#pragma unroll
    for (int off = 32; off >= kReduceCols; off >>= 1) s += __shfl_down(s, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < kReduceCols) red[wave * kReduceCols + lane] = s;
    __syncthreads();
    double tsum = 0.0;
    if (threadIdx.x < kReduceCols) {
        tsum = red[c];
        for (int k = 1; k < kWaves; ++k) tsum += red[k * kReduceCols + c];
    }
    return tsum;   // valid in threads [0, kReduceCols) with col < n1
}

// Tail of a fold block: its (up to) kReduceCols outputs go to `out` (plain stores, device consumers) or -- host.gran
// set -- to mapped host memory as SELF-VALIDATING granules {tag, 32 bits of the double} (two per column): the block's 8
// columns leave as 16 granules = 128 contiguous, 128-byte aligned bytes written by ONE store instruction (lane 2c + h
// carries half h of column c): two full 64-byte lines on the PCIe side -- a line written in pieces costs a
// read-modify-write per piece at the host's memory controller (measured: 2x slower evaluations).  Columns past n1
// travel as zeros so that the last block writes full lines too.  The host re-reads the granules until all carry the
// tag: no drain of the output stores, no ticket, no flag word.  Wavefront 0 only; tsum valid in lanes [0, kReduceCols).
__device__ __forceinline__ void fold_finish(double tsum, bool ok, int n1, int colblock, double* out, HostOut host)
{
    const int tid = threadIdx.x;
    if (tid >= 64) return;
    const int col = colblock * kReduceCols + tid;
    if (host.gran) {
        const double val = (tid < kReduceCols && col < n1) ? (ok ? tsum : __builtin_nan("")) : 0.0;
        const long long bits = __shfl(__double_as_longlong(val), (tid >> 1) & (kReduceCols - 1), 64);
        if (tid < 2 * kReduceCols) {
            const unsigned long long tag = (host.tag & 0xffffffffull) << 32, u = (unsigned long long)bits;
            __hip_atomic_store(host.gran + 2 * (size_t)colblock * kReduceCols + tid,
                               tag | ((tid & 1) ? (u >> 32) : (u & 0xffffffffull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (tid < kReduceCols && col < n1) {
        out[col] = ok ? tsum : __builtin_nan("");
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void reduce_partials(const double* __restrict__ partials, int rows, int n1, int pitch,
                                                         double* __restrict__ out, HostOut host, ArmWord arm)
{
    __shared__ double red[(BLOCK / 64) * kReduceCols];
    const int colblock = fold_colblock((int)blockIdx.x);
    if (colblock * kReduceCols >= n1) return;             // (block-uniform) padding of the grid to whole XCD deals
    // the fold of a pre-armed evaluation that was cancelled (or never got its prices) has nothing to publish; the
    // word cannot change between the threads' loads: the host moves on only after this launch's outputs
    if (arm.word && __hip_atomic_load(arm.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != arm.seq) return;
    const double tsum = fold_columns(partials, rows, n1, pitch, colblock, red);
    fold_finish(tsum, true, n1, colblock, out, host);
}

// Fold + all-reduce in one launch (sharded runs, see sweep.h).  The exchange uses self-validating
// 8-byte granules {tag = sequence number, 32 bits of payload} (MI355X_MICROARCH.md, hand-off form R2:
// "the data IS the flag"): each column travels as two granules (low / high half of the double), each
// written by ONE aligned 8-byte system-scope store, so there is no separate flag, no store drain and
// no second hop -- a reader simply re-reads a peer's granules (system-scope loads, which bypass the
// caches) until both carry this evaluation's tag.  This rank's own columns never leave registers.
// Double buffering by sequence parity: a rank rewrites gran[parity] for seq+2 only after its seq+1
// launch, which waited for every peer's seq+1 granules, i.e. for every peer's seq launch -- the one
// that read gran[parity] -- to have completed.  Waits are bounded by wall-clock time (NaN output).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void reduce_gather(const double* __restrict__ partials, int rows, int n1, int pitch,
                                                       double* __restrict__ out, PeerSet ps)
{
    __shared__ double red[(BLOCK / 64) * kReduceCols];
    const int colblock = fold_colblock((int)blockIdx.x);
    if (colblock * kReduceCols >= n1) return;             // (block-uniform) padding of the grid to whole XCD deals
    // a cancelled pre-armed evaluation is cancelled on EVERY rank (the ranks run the same solver in lockstep):
    // nobody publishes, nobody waits, and the sequence number is reused by the next launch
    if (ps.arm.word && __hip_atomic_load(ps.arm.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ps.arm.seq) return;
    const double tsum = fold_columns(partials, rows, n1, pitch, colblock, red);
    const int tid = threadIdx.x;
    if (tid >= 64) return;                                // the exchange is wavefront 0's business
    const int parity = (int)(ps.seq & 1ull);
    const unsigned long long tag = (ps.seq % 0xffffffffull + 1ull) << 32;   // never 0 (= an empty buffer)
    const int col = colblock * kReduceCols + tid;
    if (tid < kReduceCols && col < n1 && ps.world > 1) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(tsum);
        unsigned long long* g = ps.gran[ps.rank] + 2 * ((long long)parity * ps.count + col);
        __hip_atomic_store(g, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // lane = (peer slot q, column c): 8 peers x 8 columns per pass, two passes cover kMaxPeers = 16
    const int q = tid / kReduceCols, c = tid % kReduceCols;
    const int colc = colblock * kReduceCols + c;
    const double own = __shfl(tsum, c, 64);
    double x[2] = {0.0, 0.0};
    bool ok = true;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = q + 8 * pass;
        if (p >= ps.world || colc >= n1) continue;
        if (p == ps.rank) { x[pass] = own; continue; }
        const unsigned long long* g = ps.gran[p] + 2 * ((long long)parity * ps.count + colc);
        const long long t0 = (long long)wall_clock64();
        for (;;) {
            const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((a & 0xffffffff00000000ull) == tag && (b & 0xffffffff00000000ull) == tag) {
                x[pass] = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
                break;
            }
            if ((long long)wall_clock64() - t0 > ps.timeout_ticks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    ok = __all(ok);
    double s = 0.0;                                       // rank order on every rank: bit-identical results
    for (int p = 0; p < ps.world; ++p) s += __shfl(p < 8 ? x[0] : x[1], (p & 7) * kReduceCols + c, 64);
    fold_finish(s, ok, n1, colblock, out, ps.host);
}

// Large-market Ψ (see sweep_body<..., GBINS = true>).  entries[] lists, token by token, the flat
// indices 2·pool + side of the flows that belong to the token; it is cut into chunks of at most
// kGatherChunk entries so that hub tokens (a numeraire with 10⁵ pools) are spread over many
// wavefronts.  One wavefront per chunk: lane-strided partial sums, then a fixed shuffle tree.
__global__ __launch_bounds__(256) void gather_chunks(const int2* __restrict__ chunks, const int* __restrict__ entries,
                                                     const double* __restrict__ flow, double* __restrict__ chunk_sums,
                                                     int n_chunks)
{
    // 16 lanes per chunk (4 chunks per wavefront): a typical token has a few dozen incident
    // pools, so a full wavefront per chunk would idle most lanes and be latency-bound
    constexpr int kGroup = 16;
    const int chunk = (blockIdx.x * 256 + threadIdx.x) / kGroup, lane = threadIdx.x % kGroup;
    double s = 0.0;
    if (chunk < n_chunks) {
        const int2 ch = chunks[chunk];
        for (int e = ch.x + lane; e < ch.y; e += kGroup) s += flow[entries[e]];
    }
#pragma unroll
    for (int off = kGroup / 2; off > 0; off >>= 1) s += __shfl_down(s, off, kGroup);
    if (lane == 0 && chunk < n_chunks) chunk_sums[chunk] = s;
}

// out[t] = sum of token t's chunk sums, in chunk order (t < n); the block after the last token
// block folds the dual-scalar column of the partial rows into out[n] (lane-strided, fixed tree).
__global__ __launch_bounds__(256) void token_fold(const int* __restrict__ tok_chunk_off,
                                                  const double* __restrict__ chunk_sums, double* __restrict__ out, int n,
                                                  const double* __restrict__ acc_rows, int rows)
{
    const int token_blocks = (n + 255) / 256;
    if ((int)blockIdx.x < token_blocks) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        if (t >= n) return;
        double s = 0.0;
        for (int c = tok_chunk_off[t]; c < tok_chunk_off[t + 1]; ++c) s += chunk_sums[c];
        out[t] = s;
        return;
    }
    __shared__ double wsum[4];
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += acc_rows[r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

hipError_t launch_gather(const int2* chunks, const int* entries, const double* flow, double* chunk_sums, int n_chunks,
                         const int* tok_chunk_off, double* out, int n, const double* acc_rows, int rows, hipStream_t s)
{
    if (n_chunks > 0)
        hipLaunchKernelGGL(gather_chunks, dim3((n_chunks + 15) / 16), dim3(256), 0, s, chunks, entries, flow, chunk_sums,
                           n_chunks);
    hipLaunchKernelGGL(token_fold, dim3((n + 255) / 256 + 1), dim3(256), 0, s, tok_chunk_off, chunk_sums, out, n,
                       acc_rows, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// Plain launch, or -- when a start/stop event pair is given -- a launch whose events are written by
// the command processor at the kernel's first and last wavefront (hipExtLaunchKernel): that is the
// kernel's own execution span, the quantity rocprofv3 reports, without the ~2.5 us that a
// hipEventRecord / launch / hipEventRecord bracket adds.
template <class K, class... A>
static void launch_k(K kernel, dim3 g, dim3 b, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1, A... args)
{
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, g, b, (std::uint32_t)lds, s, e0, e1, 0u, args...);
    else hipLaunchKernelGGL(kernel, g, b, lds, s, args...);
}

size_t sweep_lds_bytes(int n_pad, int copies, int block, int need_logv, int gtab_n, int stage_y)
{
    const size_t words = (size_t)n_pad * ((stage_y ? 2 : 1) + (need_logv ? 1 : 0) + copies) + 2 * (size_t)gtab_n + 2 * (size_t)(block / 64) + 2;
    return words * sizeof(double);
}

// Kernel instantiations (round 5: 82).  Per family {full-range, fast, auto} x {materialising, fused} x {512, 1024 threads};
// the reference-order GeometricMean forms and the large-market mode (GBINS, 512 threads) run full-range only.
template <class Ops, int FASTK>
static hipError_t set_lds_attr(size_t bytes)
{
    hipError_t e;
#define CFMM_SET(MAT, B)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_kernel<Ops, MAT, B, false, FASTK>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                  \
    if (e != hipSuccess) return e;
    CFMM_SET(true, kMidBlock) CFMM_SET(false, kMidBlock) CFMM_SET(true, kBigBlock) CFMM_SET(false, kBigBlock)
#undef CFMM_SET
    return hipSuccess;
}

template <int B, int FASTK>
static void launch_multi_b(const MultiArgs& ma, const LaunchCfg& c, bool mat, hipStream_t s)
{
    dim3 g(c.grid), b(B);
    if (mat) launch_k(&sweep_multi<true, B, false, FASTK>, g, b, c.lds_bytes, s, c.ev_start, c.ev_stop, ma);
    else launch_k(&sweep_multi<false, B, false, FASTK>, g, b, c.lds_bytes, s, c.ev_start, c.ev_stop, ma);
}

template <int B>
static void launch_multi_a(const MultiArgs& ma, const LaunchCfg& c, bool mat, hipStream_t s)
{
    if (c.arith == kArithFast) launch_multi_b<B, kArithFast>(ma, c, mat, s);
    else if (c.arith == kArithAuto) launch_multi_b<B, kArithAuto>(ma, c, mat, s);
    else launch_multi_b<B, kArithFull>(ma, c, mat, s);
}

hipError_t launch_multi(const MultiArgs& ma, const LaunchCfg& c, bool mat, hipStream_t s)
{
    if (ma.common.gflow) {   // large-market mode: kMidBlock, full-range arithmetic
        dim3 g(c.grid), b(kMidBlock);
        if (mat) launch_k(&sweep_multi<true, kMidBlock, true, kArithFull>, g, b, c.lds_bytes, s, c.ev_start, c.ev_stop, ma);
        else launch_k(&sweep_multi<false, kMidBlock, true, kArithFull>, g, b, c.lds_bytes, s, c.ev_start, c.ev_stop, ma);
    } else if (c.block == kBigBlock) {
        launch_multi_a<kBigBlock>(ma, c, mat, s);
    } else {
        launch_multi_a<kMidBlock>(ma, c, mat, s);
    }
    return hipGetLastError();
}

hipError_t prepare_kernels(size_t max_lds_bytes)
{
    hipError_t em;
#define CFMM_SETM(MAT, B, F)                                                                          \
    em = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_multi<MAT, B, false, F>),           \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);         \
    if (em != hipSuccess) return em;
#define CFMM_SETM4(F) CFMM_SETM(true, kBigBlock, F) CFMM_SETM(false, kBigBlock, F) CFMM_SETM(true, kMidBlock, F) CFMM_SETM(false, kMidBlock, F)
    CFMM_SETM4(kArithFull) CFMM_SETM4(kArithFast) CFMM_SETM4(kArithAuto)
#undef CFMM_SETM4
#undef CFMM_SETM
    hipError_t e;
    if ((e = set_lds_attr<ProductOps, kArithFull>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<ProductOps, kArithFast>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<ProductOps, kArithAuto>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<GeoMeanOps, kArithFull>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<GeoMeanLogOps, kArithFull>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<GeoMeanLogOps, kArithFast>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<GeoMeanLogOps, kArithAuto>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<UniV3Ops, kArithFull>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<UniV3Ops, kArithFast>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<UniV3Ops, kArithAuto>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<UniV3OpsLean, kArithFull>(max_lds_bytes)) != hipSuccess) return e;
    if ((e = set_lds_attr<UniV3OpsLean, kArithFast>(max_lds_bytes)) != hipSuccess) return e;
    return set_lds_attr<UniV3OpsLean, kArithAuto>(max_lds_bytes);
}

template <class Ops, int FASTK>
static void launch_any_f(const Ops& ops, const SweepArgs& a, const LaunchCfg& c, bool mat, hipStream_t s)
{
    dim3 g(c.grid);
    hipEvent_t e0 = c.ev_start, e1 = c.ev_stop;
    if (c.block == kBigBlock) {
        if (mat) launch_k(&sweep_kernel<Ops, true, kBigBlock, false, FASTK>, g, dim3(kBigBlock), c.lds_bytes, s, e0, e1, ops, a);
        else launch_k(&sweep_kernel<Ops, false, kBigBlock, false, FASTK>, g, dim3(kBigBlock), c.lds_bytes, s, e0, e1, ops, a);
    } else {
        if (mat) launch_k(&sweep_kernel<Ops, true, kMidBlock, false, FASTK>, g, dim3(kMidBlock), c.lds_bytes, s, e0, e1, ops, a);
        else launch_k(&sweep_kernel<Ops, false, kMidBlock, false, FASTK>, g, dim3(kMidBlock), c.lds_bytes, s, e0, e1, ops, a);
    }
}

template <class Ops, bool HAS_FAST = true>
static hipError_t launch_any(const Ops& ops, const SweepArgs& a, const LaunchCfg& c, bool mat, hipStream_t s)
{
    if (a.m <= 0) return hipSuccess;
    if (a.gflow) { // large-market mode: kMidBlock, full-range arithmetic
        dim3 g(c.grid);
        hipEvent_t e0 = c.ev_start, e1 = c.ev_stop;
        if (mat) launch_k(&sweep_kernel<Ops, true, kMidBlock, true, kArithFull>, g, dim3(kMidBlock), c.lds_bytes, s, e0, e1, ops, a);
        else launch_k(&sweep_kernel<Ops, false, kMidBlock, true, kArithFull>, g, dim3(kMidBlock), c.lds_bytes, s, e0, e1, ops, a);
    } else if constexpr (HAS_FAST) {
        if (c.arith == kArithFast) launch_any_f<Ops, kArithFast>(ops, a, c, mat, s);
        else if (c.arith == kArithAuto) launch_any_f<Ops, kArithAuto>(ops, a, c, mat, s);
        else launch_any_f<Ops, kArithFull>(ops, a, c, mat, s);
    } else {
        launch_any_f<Ops, kArithFull>(ops, a, c, mat, s);
    }
    return hipGetLastError();
}

hipError_t launch_sweep(const ProductPools& p, const SweepArgs& a, const LaunchCfg& c, bool mat, hipStream_t s)
{
    return launch_any(ProductOps{p}, a, c, mat, s);
}
hipError_t launch_sweep(const GeoMeanPools& p, const SweepArgs& a, const LaunchCfg& c, bool mat, hipStream_t s)
{
    if (p.reference_order) return launch_any<GeoMeanOps, false>(GeoMeanOps{p}, a, c, mat, s);
    return launch_any(GeoMeanLogOps{p}, a, c, mat, s);
}
hipError_t launch_sweep(const UniV3Pools& p, const SweepArgs& a, const LaunchCfg& c, bool mat, hipStream_t s)
{
    if (p.head && !a.gflow) return launch_any(UniV3Ops{p}, a, c, mat, s);
    return launch_any(UniV3OpsLean{p}, a, c, mat, s);
}

hipError_t launch_reduce(const double* partials, int rows, int n1, int pitch, double* out, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                         HostOut host, ArmWord arm)
{
    dim3 g(fold_grid(n1));
    launch_k(&reduce_partials<kFoldBlock>, g, dim3(kFoldBlock), 0, s, e0, e1, partials, rows, n1, pitch, out, host, arm);
    return hipGetLastError();
}

// update_reserves!(r) for the two-coin families -- src/router.jl:127-132 with the update the routing
// problem prescribes (find_arb! docstring, src/cfmms.jl:26-31): R <- (R + γΔ) − Λ, in place, from
// the trades of the latest materialising sweep; GeometricMean segments refresh the exponents'
// v-independent constants {Q1, Q2} (see GeoMeanLogOps) with the same expressions as the upload.
// one pool's trades from the buffers (plain or compact layout, see SweepArgs)
__device__ __forceinline__ void read_trade(const double2* __restrict__ Delta, const double2* __restrict__ Lambda,
                                           const double2* __restrict__ Over, int compact, long long i, double2& d, double2& l)
{
    if (!compact) {
        d = Delta[i];
        l = Lambda[i];
        return;
    }
    const double2 r = Delta[i];
    if (r.y == -1.0) {
        d = Lambda[i];
        l = Over[i];
    } else if (__builtin_signbit(r.x)) {
        d = make_double2(0.0, -r.x);
        l = make_double2(r.y, 0.0);
    } else {
        d = make_double2(r.x, 0.0);
        l = make_double2(0.0, r.y);
    }
}

__global__ __launch_bounds__(256) void update_two_coin(double2* __restrict__ R, const double* __restrict__ gamma,
                                                       const double2* __restrict__ Delta,
                                                       const double2* __restrict__ Lambda,
                                                       const double2* __restrict__ Over, int compact,
                                                       double2* __restrict__ Q, const double* __restrict__ eta, long long m,
                                                       int* __restrict__ left_window)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const double g = gamma[i];
    const double2 r = R[i];
    double2 d, l;
    read_trade(Delta, Lambda, Over, compact, i, d, l);
    const double2 rn = make_double2((r.x + g * d.x) - l.x, (r.y + g * d.y) - l.y);
    R[i] = rn;
    // a reserve that leaves the operand window of the fast arithmetic (sweep.h kFastExp) sends the segment back to the
    // compiler's division / square-root sequences
    if (left_window && !(in_fast_window(rn.x) && in_fast_window(rn.y))) *left_window = 1;
    if (Q) {
        const double e = eta[i], lg = log(g), le = log(e), l1 = log(rn.x), l2 = log(rn.y);
        Q[i] = make_double2(((lg + le) + l2) + e * l1, e * ((lg + l1) - le) + l2);
    }
}

hipError_t launch_update_two_coin(double2* R, const double* gamma, const double2* Delta, const double2* Lambda,
                                  const double2* Over, int compact, double2* Q, const double* eta, int64_t m, int* left_window,
                                  hipStream_t s)
{
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(update_two_coin, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, R, gamma, Delta, Lambda, Over,
                       compact, Q, eta, (long long)m, left_window);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void expand_trades(const double2* __restrict__ rec, const double2* __restrict__ ovA,
                                                     const double2* __restrict__ ovB, double2* __restrict__ Delta,
                                                     double2* __restrict__ Lambda, long long m)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    double2 d, l;
    read_trade(rec, ovA, ovB, 1, i, d, l);
    Delta[i] = d;
    Lambda[i] = l;
}

hipError_t launch_expand_trades(const double2* rec, const double2* ovA, const double2* ovB, double2* Delta, double2* Lambda,
                                int64_t m, hipStream_t s)
{
    if (m <= 0) return hipSuccess;
    hipLaunchKernelGGL(expand_trades, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, rec, ovA, ovB, Delta, Lambda,
                       (long long)m);
    return hipGetLastError();
}

hipError_t launch_reduce_gather(const double* partials, int rows, int n1, int pitch, double* out, hipStream_t s, const PeerSet& ps,
                                hipEvent_t e0, hipEvent_t e1)
{
    dim3 g(fold_grid(n1));
    launch_k(&reduce_gather<kFoldBlock>, g, dim3(kFoldBlock), 0, s, e0, e1, partials, rows, n1, pitch, out, ps);
    return hipGetLastError();
}

} // namespace cfmm
