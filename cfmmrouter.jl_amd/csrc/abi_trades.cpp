// abi_trades.cpp -- what a materialising sweep leaves behind (include/cfmm_amd.h): the trades r.Δs / r.Λs of
// src/router.jl:7-8 as host arrays (cfmm_get_trades*) or device arrays (cfmm_trades_dev), update_reserves!
// (src/router.jl:127-132) on the device, and the pool state read-back.
#include "ctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace cfmm;

namespace {

// Device arrays in the reference's layout ({Δ₁, Δ₂} / {Λ₁, Λ₂} per pool) for the trades currently on the device:
// the plain buffers themselves, or the expansion of the compact records (one kernel, asynchronous on the stream).
int expanded_trades(cfmm_ctx* c, const double2** dD, const double2** dL)
{
    if (!c->trades_compact) {
        *dD = c->d_delta;
        *dL = c->d_lambda;
        return CFMM_OK;
    }
    if (c->m_total > c->x_cap) {
        (void)hipFree(c->d_xdelta); (void)hipFree(c->d_xlambda);
        c->d_xdelta = c->d_xlambda = nullptr;
        c->x_cap = 0;
        c->x_valid = false;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_xdelta), (size_t)c->m_total * sizeof(double2)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_xlambda), (size_t)c->m_total * sizeof(double2)));
        c->x_cap = c->m_total;
    }
    if (!c->x_valid && c->have_trades) {
        hipError_t e = launch_expand_trades(c->d_delta, c->d_lambda, c->d_over, c->d_xdelta, c->d_xlambda, c->m_total, c->stream);
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "expand launch failed: %s", hipGetErrorString(e));
        c->x_valid = true;
    }
    *dD = c->d_xdelta;
    *dL = c->d_xlambda;
    return CFMM_OK;
}

int ensure_staging(cfmm_ctx* c)
{
    TradeStaging& t = c->tstage;
    if (t.ready) return CFMM_OK;
    for (int k = 0; k < TradeStaging::kThreads; ++k) {
        HIP_TRY(c, hipStreamCreateWithFlags(&t.stream[k], hipStreamNonBlocking));
        for (int s = 0; s < TradeStaging::kSlots; ++s) {
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&t.slot[k][s]), (size_t)TradeStaging::kChunkRows * sizeof(double2), hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&t.done[k][s], hipEventDisableTiming));
        }
    }
    t.ready = true;
    return CFMM_OK;
}

// D2H of trade rows [row0, row0 + count) into Delta / Lambda ([count][2] each, either may be null).
// Round 2 copied the 16-byte records into a pageable vector and decoded them on one thread (8 ms per 1M pools, and two
// more full copies whenever a single pool had used the overflow rows).  Now the records are expanded on the DEVICE
// (expand_trades: 0.03 ms per 1M pools, overflow rows included) and the two result arrays stream to the host in 1 MiB
// chunks through pinned double buffers: kThreads workers, each with its own stream, overlap the PCIe copies with the
// copies from the pinned slots into the caller's (pageable) arrays -- the part that bounds the call.
int download_trades(cfmm_ctx* c, int64_t row0, int64_t count, double* Delta, double* Lambda)
{
    if (count == 0 || (!Delta && !Lambda)) return CFMM_OK;
    const double2 *dD = nullptr, *dL = nullptr;
    int rc = expanded_trades(c, &dD, &dL);
    if (rc != CFMM_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the sweep and the expansion have completed
    if (count < 2 * TradeStaging::kChunkRows) {     // small ranges: plain copies
        if (Delta) HIP_TRY(c, hipMemcpy(Delta, dD + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        if (Lambda) HIP_TRY(c, hipMemcpy(Lambda, dL + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        return CFMM_OK;
    }
    rc = ensure_staging(c);
    if (rc != CFMM_OK) return rc;
    struct Chunk { const double2* src; double* dst; int64_t rows; };
    std::vector<Chunk> chunks;
    for (int arr = 0; arr < 2; ++arr) {
        const double2* src = (arr == 0 ? dD : dL) + row0;
        double* dst = arr == 0 ? Delta : Lambda;
        if (!dst) continue;
        for (int64_t r = 0; r < count; r += TradeStaging::kChunkRows)
            chunks.push_back({src + r, dst + 2 * r, std::min<int64_t>(TradeStaging::kChunkRows, count - r)});
    }
    TradeStaging& t = c->tstage;
    const int device = c->device;
    std::vector<hipError_t> errs((size_t)TradeStaging::kThreads, hipSuccess);
    auto work = [&](int k) {
        hipError_t e = hipSetDevice(device);
        // chunks k, k + T, k + 2T, ...: the copy of the next chunk is in flight while this one is moved out of its slot
        std::vector<size_t> mine;
        for (size_t j = (size_t)k; j < chunks.size(); j += TradeStaging::kThreads) mine.push_back(j);
        auto issue = [&](size_t idx) {
            const Chunk& ch = chunks[mine[idx]];
            const int s = (int)(idx % TradeStaging::kSlots);
            hipError_t ee = hipMemcpyAsync(t.slot[k][s], ch.src, (size_t)ch.rows * sizeof(double2), hipMemcpyDeviceToHost, t.stream[k]);
            if (ee == hipSuccess) ee = hipEventRecord(t.done[k][s], t.stream[k]);
            return ee;
        };
        if (e == hipSuccess && !mine.empty()) e = issue(0);
        for (size_t idx = 0; idx < mine.size() && e == hipSuccess; ++idx) {
            if (idx + 1 < mine.size()) e = issue(idx + 1);
            if (e != hipSuccess) break;
            const int s = (int)(idx % TradeStaging::kSlots);
            e = hipEventSynchronize(t.done[k][s]);
            if (e != hipSuccess) break;
            const Chunk& ch = chunks[mine[idx]];
            std::memcpy(ch.dst, t.slot[k][s], (size_t)ch.rows * sizeof(double2));
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(t.stream[k]);
        errs[(size_t)k] = e;
    };
    std::vector<std::thread> helpers;
    for (int k = 1; k < TradeStaging::kThreads; ++k) helpers.emplace_back(work, k);
    work(0);
    for (auto& th : helpers) th.join();
    for (hipError_t e : errs)
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "trade download failed: %s", hipGetErrorString(e));
    return CFMM_OK;
}

} // namespace

namespace cfmm {

void free_trade_staging(cfmm_ctx* c)
{
    TradeStaging& t = c->tstage;
    for (int k = 0; k < TradeStaging::kThreads; ++k) {
        for (int s = 0; s < TradeStaging::kSlots; ++s) {
            if (t.slot[k][s]) (void)hipHostFree(t.slot[k][s]);
            if (t.done[k][s]) (void)hipEventDestroy(t.done[k][s]);
            t.slot[k][s] = nullptr;
            t.done[k][s] = nullptr;
        }
        if (t.stream[k]) (void)hipStreamDestroy(t.stream[k]);
        t.stream[k] = nullptr;
    }
    t.ready = false;
}

} // namespace cfmm

extern "C" {

int cfmm_get_trades_range(cfmm_ctx* c, int32_t seg, int64_t first, int64_t count, double* Delta, double* Lambda)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) return multi_get_trades_range(c, seg, first, count, Delta, Lambda);
    if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb first");
    if (seg < 0 || seg >= (int32_t)c->segs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    const Segment& s = c->segs[(size_t)seg];
    if (first < 0 || count < 0 || first + count > s.m) return fail(c, CFMM_ERR_INVALID_ARG, "row range out of bounds");
    if (count == 0) return CFMM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    return download_trades(c, s.trade_off + first, count, Delta, Lambda);
}

int cfmm_get_trades(cfmm_ctx* c, double* Delta, double* Lambda)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb first");
    if (c->m_total == 0) return CFMM_OK;
    if (!c->shards.empty()) {
        for (size_t k = 0; k < c->psegs.size(); ++k) {
            const auto& ps = c->psegs[k];
            const int rc = multi_get_trades_range(c, (int32_t)k, 0, ps.m, Delta ? Delta + 2 * ps.trade_off : nullptr,
                                                  Lambda ? Lambda + 2 * ps.trade_off : nullptr);
            if (rc != CFMM_OK) return rc;
        }
        return CFMM_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    return download_trades(c, 0, c->m_total, Delta, Lambda);
}

int cfmm_trades_dev(cfmm_ctx* c, const double** d_delta, const double** d_lambda)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_trades_dev");
    int rc = ensure_geometry(c);
    if (rc != CFMM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    // device consumers get the reference's layout: the compact records of the latest materialising sweep are
    // expanded (asynchronously, on the context's stream) into {Δ₁, Δ₂} / {Λ₁, Λ₂} arrays -- call again after
    // every sweep whose trades are wanted
    if (!c->have_trades) c->trades_compact = (c->opt_compact_trades != 0 && !global_bins(c)) ? 1 : 0;
    const double2 *dD = nullptr, *dL = nullptr;
    rc = expanded_trades(c, &dD, &dL);
    if (rc != CFMM_OK) return rc;
    if (d_delta) *d_delta = reinterpret_cast<const double*>(dD);
    if (d_lambda) *d_lambda = reinterpret_cast<const double*>(dL);
    return CFMM_OK;
}

int cfmm_update_reserves(cfmm_ctx* c)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {
        if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb / cfmm_route first");
        for (size_t d = 0; d < c->shards.size(); ++d) {
            cfmm_ctx* child = c->shards[d];
            if (child->segs.empty()) continue;
            const int rc = cfmm_update_reserves(child);
            if (rc != CFMM_OK) {
                c->have_trades = c->have_out = false;   // some shards have moved: the trades describe no consistent market any more
                return fail(c, rc, "shard %d: %s", (int)d, child->err.c_str());
            }
        }
        c->have_trades = c->have_out = false;
        return CFMM_OK;
    }
    if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb / cfmm_route first");
    bool any_univ3 = false;
    for (const Segment& s : c->segs) any_univ3 = any_univ3 || s.kind == CFMM_KIND_UNIV3;
    if (any_univ3 && (int)c->trade_v.size() != c->n)
        return fail(c, CFMM_ERR_STATE, "UniV3 pools need the prices of the trades: run the materialising sweep through "
                                       "cfmm_find_arb / cfmm_route (host pointer), not cfmm_sweep_dev");
    HIP_TRY(c, hipSetDevice(c->device));
    // Phase 1 (no side effects): the replacement of every UniV3 segment.  The pool's state is its price: find_arb!
    // (src/cfmms.jl:339-395) moves a trading pool to the internal price P = p/γ (price falling, :361) or γ·p (price
    // rising, :381 in the flipped frame), p = v₁/v₂ -- through every fully drained tick and part of the last one -- and
    // leaves a pool inside its no-arbitrage band (:347-349) alone.  P above the first tick means the pool ran out of
    // liquidity on that side and rests at the first tick's upper price.  Tick constants are then re-derived exactly
    // as at upload (compute_at_tick, :294-313).  A failure here leaves the context untouched (the call can be retried).
    std::vector<Segment> fresh(c->segs.size());
    std::vector<std::vector<double>> new_cp(c->segs.size());
    auto drop_fresh = [&]() { for (Segment& ns : fresh) free_segment(ns); };
    const double* v = c->trade_v.data();
    for (size_t k = 0; k < c->segs.size(); ++k) {
        const Segment& s = c->segs[k];
        if (s.kind != CFMM_KIND_UNIV3) continue;
        std::vector<double>& cp = new_cp[k];
        cp = s.h_cp;
        for (int64_t i = 0; i < s.m; ++i) {
            const double g = s.h_gamma[(size_t)i], q = s.h_cp[(size_t)i];
            const double pr = v[s.h_ai[(size_t)(2 * i)]] / v[s.h_ai[(size_t)(2 * i + 1)]];   // :340
            if (g * q <= pr && pr <= q / g) continue;                                        // :347-349
            const double P = pr < g * q ? pr / g : g * pr;
            const double top = s.h_lt[(size_t)s.h_tick_off[(size_t)i]];
            cp[(size_t)i] = P > top ? top : P;
        }
        const int rc = univ3_build(c, fresh[k], s.m, cp.data(), s.h_gamma.data(), s.h_ai.data(), s.h_tick_off.data(),
                                   s.h_lt.data(), s.h_liq.data());
        if (rc != CFMM_OK) { drop_fresh(); return rc; }
    }
    // Phase 2: R <- R + γΔ − Λ for the two-coin families on the device (no host traffic); from the first launch on the
    // trades count as consumed, so that a failure cannot lead to a second application of the same trades.
    c->have_trades = false;
    c->have_out = false;
    int* d_left = nullptr;   // per segment: 1 = a new reserve left the operand window of the fast arithmetic
    std::vector<int> left(c->segs.size(), 0);
    if (hipMalloc(reinterpret_cast<void**>(&d_left), c->segs.size() * sizeof(int)) != hipSuccess ||
        hipMemsetAsync(d_left, 0, c->segs.size() * sizeof(int), c->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d_left);
        drop_fresh();
        c->trade_v.clear();
        return fail(c, CFMM_ERR_HIP, "update_reserves: scratch allocation failed");
    }
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < c->segs.size() && e == hipSuccess; ++k) {
        Segment& s = c->segs[k];
        if (s.kind == CFMM_KIND_UNIV3) continue;
        e = launch_update_two_coin(s.R, s.gamma, c->d_delta + s.trade_off, c->d_lambda + s.trade_off, c->d_over + s.trade_off,
                                   c->trades_compact, s.kind == CFMM_KIND_GEOMEAN ? s.lR : nullptr, s.eta, s.m, d_left + k, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(left.data(), d_left, left.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // also: nothing in flight still reads the old UniV3 constants
    (void)hipFree(d_left);
    c->trade_v.clear();
    c->x_valid = false;
    if (e != hipSuccess) {
        drop_fresh();
        return fail(c, CFMM_ERR_HIP, "update launch failed: %s (two-coin reserves may have moved; the trades are consumed)", hipGetErrorString(e));
    }
    // Phase 3: swap in the UniV3 replacements (pointer moves only: cannot fail).
    for (size_t k = 0; k < c->segs.size(); ++k) {
        Segment& s = c->segs[k];
        if (s.kind != CFMM_KIND_UNIV3) {
            if (left[k]) s.fast_ok = 0;
            continue;
        }
        Segment& ns = fresh[k];
        (void)hipFree(s.pg); (void)hipFree(s.Ai); (void)hipFree(s.cur_a); (void)hipFree(s.cur_b); (void)hipFree(s.cur_c);
        (void)hipFree(s.curR); (void)hipFree(s.walk); (void)hipFree(s.ticks); (void)hipFree(s.thr);
        (void)hipFree(s.cp); (void)hipFree(s.pk); (void)hipFree(s.head);
        s.head = ns.head;
        s.pg = ns.pg; s.Ai = ns.Ai; s.cur_a = ns.cur_a; s.cur_b = ns.cur_b; s.cur_c = ns.cur_c; s.curR = ns.curR;
        s.walk = ns.walk; s.ticks = ns.ticks; s.thr = ns.thr; s.has_walk = ns.has_walk;
        s.cp = ns.cp; s.pk = ns.pk; s.gvals.swap(ns.gvals); s.fast_ok = ns.fast_ok;
        s.h_cp.swap(new_cp[k]);
        ns = Segment{};   // ownership moved
    }
    return CFMM_OK;
}

int cfmm_get_reserves(cfmm_ctx* c, int32_t seg, double* R)
{
    if (!c || !R) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {
        if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
        const int nd = (int)c->shards.size();
        for (int d = 0; d < nd; ++d) {
            int64_t lo, hi;
            shard_range(c->psegs[(size_t)seg].m, d, nd, lo, hi);
            if (hi == lo) continue;
            const int rc = cfmm_get_reserves(c->shards[(size_t)d], child_segment(c, seg, d), R + 2 * lo);
            if (rc != CFMM_OK) return fail(c, rc, "shard %d: %s", d, c->shards[(size_t)d]->err.c_str());
        }
        return CFMM_OK;
    }
    if (seg < 0 || seg >= (int32_t)c->segs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    const Segment& s = c->segs[(size_t)seg];
    if (s.kind == CFMM_KIND_UNIV3) return fail(c, CFMM_ERR_INVALID_ARG, "UniV3 segments have prices, not reserves: cfmm_get_prices");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(R, s.R, (size_t)s.m * sizeof(double2), hipMemcpyDeviceToHost));
    return CFMM_OK;
}

int cfmm_get_prices(cfmm_ctx* c, int32_t seg, double* current_price)
{
    if (!c || !current_price) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {
        if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
        const int nd = (int)c->shards.size();
        for (int d = 0; d < nd; ++d) {
            int64_t lo, hi;
            shard_range(c->psegs[(size_t)seg].m, d, nd, lo, hi);
            if (hi == lo) continue;
            const int rc = cfmm_get_prices(c->shards[(size_t)d], child_segment(c, seg, d), current_price + lo);
            if (rc != CFMM_OK) return fail(c, rc, "shard %d: %s", d, c->shards[(size_t)d]->err.c_str());
        }
        return CFMM_OK;
    }
    if (seg < 0 || seg >= (int32_t)c->segs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    const Segment& s = c->segs[(size_t)seg];
    if (s.kind != CFMM_KIND_UNIV3) return fail(c, CFMM_ERR_INVALID_ARG, "not a UniV3 segment: cfmm_get_reserves");
    std::copy(s.h_cp.begin(), s.h_cp.end(), current_price);
    return CFMM_OK;
}

} // extern "C"
