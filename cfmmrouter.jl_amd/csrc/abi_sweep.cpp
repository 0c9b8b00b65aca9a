// abi_sweep.cpp -- one evaluation on the device: launch geometry, the sweep launches of every pool family, the row
// fold; host-pointer sweeps (cfmm_find_arb / cfmm_eval) and the pre-armed evaluations cfmm_route uses.
//   find_arb!(r, v)          src/router.jl:38-42     -> enqueue_sweep(materialize = true)
//   fn / g! per evaluation   src/router.jl:73-102    -> enqueue_sweep(materialize = false): {Ψ, acc}
//   netflows(r)              src/router.jl:111-125   -> cfmm_netflows (the Ψ of the latest sweep)
#include "ctx.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

using namespace cfmm;

namespace {

// Grid cap for the fat (512/1024-thread) blocks: HALF a machine of resident threads -- one 1024-thread block
// (16 wavefronts) per CU.  Round 1 ran a full machine (two blocks per CU); with consecutive sweeps walking the
// tiles in alternating directions (option "alternate") fewer, longer lanes win: each lane owns 2x the tiles, so
// more of a sweep starts on L2-resident data, and there are half as many partial rows and LDS prologues
// (measured, sweep span us at 256 / 384 / 512 blocks: product1m 9.7 / 10.6 / 10.6, config5 19.1 / 22.6 / 21.7,
// config-4 shard 7.1 / - / 8.1; 128 blocks: 14.9 / 24.5 / 9.2).
int fat_grid_cap(const cfmm_ctx* c, int block)
{
    (void)c;
    return kResidentThreads / 2 / block;
}

// Fused multi-family launches: 512 blocks of 512 threads in total measured best on config3
// (19.9 us per step vs 21.9 at 1024 blocks and 21.4 at 256; bench.py --opt block=.. --opt max_grid=..).
int fused_grid_cap(const cfmm_ctx* c, int block)
{
    return std::min(fat_grid_cap(c, block), kResidentThreads / 2 / block);
}

// Launch geometry for a segment of m pools.  Small markets: 512-thread blocks, one tile each.  Large markets:
// 1024-thread blocks, at most one per CU, each striding over many tiles -- this keeps the number of partial rows
// (and the fold kernel) small.  Large-market mode (global bins) uses 512-thread blocks throughout.
void plan_segment(const cfmm_ctx* c, Segment& s)
{
    // tiny single-family markets: ONE block, whose row is the result (SweepArgs::direct: no fold launch)
    if (c->opt_direct_small != 0 && c->segs.size() == 1 && s.m <= kDirectPools && !global_bins(c) && c->opt_block == 0 &&
        c->opt_max_grid == 0) {
        s.block = kBigBlock;
        s.grid = 1;
        return;
    }
    const int64_t tiles_mid = std::max<int64_t>(1, (s.m + kMidBlock - 1) / kMidBlock);
    const bool small = global_bins(c) || c->opt_block == kMidBlock || (c->opt_block == 0 && tiles_mid <= 256);
    if (small) {
        s.block = kMidBlock;
        const int64_t cap = c->opt_max_grid > 0 ? c->opt_max_grid : (tiles_mid <= 256 ? 256 : fat_grid_cap(c, kMidBlock));
        s.grid = (int)std::min<int64_t>(tiles_mid, cap);
    } else {
        s.block = kBigBlock;
        const int64_t tiles = std::max<int64_t>(1, (s.m + kBigBlock - 1) / kBigBlock);
        s.grid = (int)std::min<int64_t>(tiles, c->opt_max_grid > 0 ? c->opt_max_grid : fat_grid_cap(c, s.block));
    }
}

// Relative cost of one pool evaluation per family, in tenths of a ProductTwoCoin evaluation (options
// "cost_geomean" / "cost_univ3"; measured on config3 / mixed markets, see DESIGN).  Used only to divide
// the blocks of a fused launch among its segments so that they finish together.
int64_t family_cost(const cfmm_ctx* c, const Segment& s)
{
    switch (s.kind) {
    case CFMM_KIND_PRODUCT: return 10;
    case CFMM_KIND_GEOMEAN: return c->opt_cost_geomean;
    default: return c->opt_cost_univ3 + (s.m > 0 && s.n_ticks_total / s.m > 2 ? 6 : 0);   // multi-tick ladders: a threshold scan + one more record
    }
}

// XCD-aware, cost-weighted map of a fused launch (grid a multiple of 256 blocks): 32-deal pattern in
// which segment s appears seg_w[s] times, spread evenly (largest-remainder weights, Bresenham order).
void plan_xcd_map(cfmm_ctx* c, Group& g)
{
    g.xcd_map = false;
    if (!g.multi || g.grid % 256 != 0 || global_bins(c)) return;
    double cost[kMaxMulti], total = 0.0;
    for (int k = 0; k < g.nseg; ++k) {
        const Segment& s = c->segs[(size_t)g.first + k];
        cost[k] = (double)s.m * (double)family_cost(c, s);
        total += cost[k];
    }
    if (!(total > 0.0)) return;
    int w[kMaxMulti], sum = 0;
    double frac[kMaxMulti];
    for (int k = 0; k < g.nseg; ++k) {
        const double share = 32.0 * cost[k] / total;
        w[k] = std::max(1, (int)share);
        frac[k] = share - (int)share;
        sum += w[k];
    }
    while (sum < 32) {   // hand the remaining deals to the largest remainders
        int best = 0;
        for (int k = 1; k < g.nseg; ++k) if (frac[k] > frac[best]) best = k;
        ++w[best]; frac[best] = -1.0; ++sum;
    }
    while (sum > 32) {   // (only when several tiny segments were rounded up to one deal each)
        int big = 0;
        for (int k = 1; k < g.nseg; ++k) if (w[k] > w[big]) big = k;
        --w[big]; --sum;
    }
    // Bresenham spread: at every position pick the segment that is furthest behind its share
    int given[kMaxMulti] = {0};
    for (int p = 0; p < 32; ++p) {
        int best = -1;
        double lag_best = -1e30;
        for (int k = 0; k < g.nseg; ++k) {
            if (given[k] >= w[k]) continue;
            const double lag = (double)(p + 1) * w[k] / 32.0 - given[k];
            if (lag > lag_best) { lag_best = lag; best = k; }
        }
        g.pattern[p] = (unsigned char)best;
        g.rank[p] = (unsigned char)given[best];
        ++given[best];
    }
    for (int k = 0; k < g.nseg; ++k) {
        g.seg_w[k] = w[k];
        c->segs[(size_t)g.first + k].grid = (g.grid / 256) * w[k] * 8;
    }
    g.xcd_map = true;
}

// Prices are staged in LDS as {v, rcp_refined(v)} pairs unless the market is too wide for them (sweep.h SweepArgs::v_shift)
bool stage_pairs(const cfmm_ctx* c, int block)
{
    return !global_bins(c) && sweep_lds_bytes(c->n_pad, 1, block, 1, kMaxFeeTable, 1) <= 160 * 1024;
}

int bin_copies(const cfmm_ctx* c, int block)
{
    if (global_bins(c)) return 1;
    const int waves = block / 64;
    if (c->opt_bin_copies == 1) return 1;
    // incl. the log-price row and the fee table a launch may stage
    const size_t per_wave = sweep_lds_bytes(c->n_pad, waves, block, 1, kMaxFeeTable, stage_pairs(c, block) ? 1 : 0);
    if (c->opt_bin_copies == 2) return per_wave <= 160 * 1024 ? waves : 1;
    // auto: one private copy per wavefront while the launch geometry's blocks still fit a CU's 160 KiB of LDS together
    // (1024-thread blocks: one per CU; 512-thread blocks: two)
    return per_wave <= (block == kBigBlock ? 128 : 64) * 1024 ? waves : 1;
}

// Large-market mode: token -> (pool, side) incidence in CSR form, cut into chunks of at most
// kGatherChunk entries (hub tokens are spread over many wavefronts), plus the flow scratch.
int build_incidence(cfmm_ctx* c)
{
    const int64_t m = c->m_total;
    if (2 * m > (int64_t)INT32_MAX) return fail(c, CFMM_ERR_UNSUPPORTED, "large-market mode supports up to 2^30 pools");
    std::vector<int> off((size_t)c->n + 1, 0);
    for (const auto& s : c->segs)
        for (int64_t k = 0; k < 2 * s.m; ++k) ++off[(size_t)s.h_ai[(size_t)k] + 1];
    for (int t = 0; t < c->n; ++t) off[(size_t)t + 1] += off[(size_t)t];
    std::vector<int> entries((size_t)(2 * m)), cursor(off.begin(), off.end() - 1);
    for (const auto& s : c->segs)
        for (int64_t i = 0; i < s.m; ++i)
            for (int side = 0; side < 2; ++side)
                entries[(size_t)cursor[(size_t)s.h_ai[(size_t)(2 * i + side)]]++] = (int)(2 * (s.trade_off + i) + side);
    std::vector<int2> chunks;
    std::vector<int> tok_chunk_off((size_t)c->n + 1, 0);
    for (int t = 0; t < c->n; ++t) {
        for (int b = off[(size_t)t]; b < off[(size_t)t + 1]; b += kGatherChunk)
            chunks.push_back(make_int2(b, std::min(b + kGatherChunk, off[(size_t)t + 1])));
        tok_chunk_off[(size_t)t + 1] = (int)chunks.size();
    }
    (void)hipFree(c->d_flow); (void)hipFree(c->d_entries); (void)hipFree(c->d_chunks);
    (void)hipFree(c->d_tok_chunk_off); (void)hipFree(c->d_chunk_sums);
    c->d_flow = nullptr; c->d_entries = nullptr; c->d_chunks = nullptr; c->d_tok_chunk_off = nullptr; c->d_chunk_sums = nullptr;
    c->n_chunks = (int)chunks.size();
    int rc;
    if ((rc = upload(c, &c->d_entries, entries.data(), entries.size())) ||
        (rc = upload(c, &c->d_chunks, chunks.data(), chunks.size())) ||
        (rc = upload(c, &c->d_tok_chunk_off, tok_chunk_off.data(), tok_chunk_off.size())))
        return rc;
    if (m > 0) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_flow), (size_t)m * sizeof(double2)));
    if (c->n_chunks > 0) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_chunk_sums), (size_t)c->n_chunks * sizeof(double)));
    return CFMM_OK;
}

hipEvent_t take_event(cfmm_ctx* c)
{
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

} // namespace

namespace cfmm {

int ensure_geometry(cfmm_ctx* c)
{
    if (!c->geometry_dirty) return CFMM_OK;
    int64_t rows = 0, trades = 0;
    bool fusable = c->opt_fuse_segments != 0 && c->segs.size() >= 2 && c->opt_geomean_exact == 0;
    bool any_big = false;
    for (auto& s : c->segs) {
        plan_segment(c, s);
        s.trade_off = trades;
        trades += s.m;
        any_big = any_big || s.block == kBigBlock;
    }
    c->groups.clear();
    if (fusable) {
        // fused launches use 512-thread blocks (Product / GeoMean blocks interleave on every CU) unless asked otherwise
        const int block = (any_big && c->opt_block == kBigBlock) ? kBigBlock : kMidBlock;
        (void)any_big;
        for (size_t first = 0; first < c->segs.size(); first += kMaxMulti) {
            Group g;
            g.first = (int)first;
            g.nseg = (int)std::min<size_t>(kMaxMulti, c->segs.size() - first);
            g.multi = g.nseg >= 2;
            g.block = block;
            int64_t tiles = 1;
            for (int k = 0; k < g.nseg; ++k) {
                Segment& sg = c->segs[first + k];
                sg.block = block;
                tiles = std::max<int64_t>(tiles, (sg.m + block - 1) / block);
            }
            const int64_t cap = std::max<int64_t>(
                1, (c->opt_max_grid > 0 ? c->opt_max_grid : fused_grid_cap(c, block)) / g.nseg);
            const int per_seg = (int)std::min<int64_t>(tiles, cap);
            for (int k = 0; k < g.nseg; ++k) c->segs[first + k].grid = per_seg;
            g.grid = per_seg * g.nseg;
            plan_xcd_map(c, g);   // may re-divide the same number of blocks among the segments by cost
            g.row_off = rows;
            c->segs[first].row_off = rows;
            rows += g.grid;
            c->groups.push_back(g);
        }
    } else {
        for (size_t i = 0; i < c->segs.size(); ++i) {
            Segment& sg = c->segs[i];
            Group g;
            g.first = (int)i;
            g.block = sg.block;
            g.grid = sg.grid;
            g.row_off = rows;
            sg.row_off = rows;
            rows += sg.grid;
            c->groups.push_back(g);
        }
    }
    // fee tables of the launches: the packed records of a launch's segments index ONE table staged in LDS
    {
        std::vector<double> tabs(c->groups.size() * (size_t)kMaxFeeTable, 1.0);
        for (size_t gi = 0; gi < c->groups.size(); ++gi) {
            Group& g = c->groups[gi];
            int total = 0;
            bool ok = c->opt_pack != 0 && !global_bins(c);
            for (int k = 0; k < g.nseg && ok; ++k) {
                const Segment& sg = c->segs[(size_t)g.first + k];
                if (!sg.pk || sg.gvals.empty()) ok = false;   // (empty: the segment has more fee tiers than a table holds)
                total += (int)sg.gvals.size();
            }
            g.gtab_n = ok && total <= kMaxFeeTable ? total : 0;
            if (g.gtab_n == 0) continue;
            int base = 0;
            for (int k = 0; k < g.nseg; ++k) {
                Segment& sg = c->segs[(size_t)g.first + k];
                sg.gbase = base;
                std::copy(sg.gvals.begin(), sg.gvals.end(), tabs.begin() + (std::ptrdiff_t)(gi * kMaxFeeTable + (size_t)base));
                base += (int)sg.gvals.size();
            }
        }
        if (tabs.size() > c->gtab_cap) {
            (void)hipFree(c->d_gtab);
            c->d_gtab = nullptr;
            c->gtab_cap = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_gtab), tabs.size() * sizeof(double)));
            c->gtab_cap = tabs.size();
        }
        if (!tabs.empty()) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(c->d_gtab, tabs.data(), tabs.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    c->rows_total = rows;
    c->m_total = trades;
    c->touched_bytes = 0;
    for (const auto& s : c->segs)   // bytes read per pool in the packed layout + one 16-byte trade record (a lower bound for multi-tick UniV3)
        c->touched_bytes += s.m * (int64_t)(s.kind == CFMM_KIND_PRODUCT ? 24 + 16 : s.kind == CFMM_KIND_GEOMEAN ? 48 + 16 : (s.has_walk ? 104 : 56) + 16);
    if (rows > c->rows_cap) {
        (void)hipFree(c->d_partials);
        c->d_partials = nullptr;
        c->rows_cap = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_partials), (size_t)rows * row_width(c) * sizeof(double)));
        HIP_TRY(c, hipMemset(c->d_partials, 0, (size_t)rows * row_width(c) * sizeof(double)));
        c->rows_cap = rows;
    }
    if (trades > c->trade_cap) {
        (void)hipFree(c->d_delta); (void)hipFree(c->d_lambda); (void)hipFree(c->d_over);
        c->d_delta = c->d_lambda = c->d_over = nullptr;
        c->trade_cap = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_delta), (size_t)trades * sizeof(double2)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_lambda), (size_t)trades * sizeof(double2)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_over), (size_t)trades * sizeof(double2)));
        c->trade_cap = trades;
    }
    if (global_bins(c)) {
        int rc = build_incidence(c);
        if (rc != CFMM_OK) return rc;
    }
    c->geometry_dirty = false;
    c->have_trades = false;
    c->x_valid = false;
    c->have_out = false;
    return CFMM_OK;
}
// Enqueue one full evaluation on c->stream: every segment's sweep, then the row fold.
//   want_host_out: the fold delivers {Ψ, acc} to the pinned staging buffer as self-validating granules (the caller
//                  polls them: host_sweep_end / armed_wait) instead of writing d_out;
//   arm_seq != 0:  pre-armed launch (SweepArgs::arm_word): v arrives later through c->d_arm.
//   price_window:  what the host knows about v -- kPricesInWindow (host-pointer sweeps whose prices lie inside the
//                  window of the fast arithmetic), kPricesOutside, or kPricesUnknown (device-pointer sweeps, pre-armed
//                  launches: the fast kernel is launched and its blocks verify, see sweep_tiles).
int enqueue_sweep(cfmm_ctx* c, const double* d_v, double* d_out, bool materialize, bool want_host_out, uint64_t arm_seq,
                  int price_window)
{
    int rc = ensure_geometry(c);
    if (rc != CFMM_OK) return rc;
    const bool timed = c->opt_time_kernels != 0 && c->pending.size() < (1u << 20); // harvest with cfmm_kernel_times
    const bool gb = global_bins(c);
    const bool sharded = !c->peers.empty();   // fold + all-reduce over the peer mappings in one launch
    const unsigned long long* arm_word = arm_seq ? reinterpret_cast<const unsigned long long*>(c->d_arm + c->n_pad) : nullptr;
    const bool rccl = c->rccl_comm != nullptr;   // the fold's {Ψ, acc} are this rank's part: all-reduced in-stream behind it
    const bool host_out = want_host_out && !gb && !rccl && (sharded || c->rows_total > 0) && c->d_stage != nullptr;
    HostOut ho{nullptr, 0};
    if (host_out) {
        ++c->out_seq;
        ho.gran = reinterpret_cast<unsigned long long*>(c->d_stage + c->gran_off);
        ho.tag = c->out_seq % 0xffffffffull + 1ull;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // a launch of one block needs no fold: its row goes straight to the consumer (single-GPU contexts: a sharded fold also
    // exchanges, and RCCL all-reduces d_out behind the fold)
    const bool direct = c->groups.size() == 1 && c->groups[0].grid == 1 && !c->groups[0].multi && !gb && !sharded;
    size_t group_index = 0;
    for (const Group& g : c->groups) {
        const size_t gi = group_index++;
        SweepArgs a;
        std::memset(&a, 0, sizeof a);
        a.v = d_v;
        a.n = c->n;
        a.n_pad = c->n_pad;
        a.v_shift = stage_pairs(c, g.block) ? 4 : 3;
        a.gtab = c->d_gtab ? c->d_gtab + gi * kMaxFeeTable : nullptr;
        a.gtab_n = a.gtab ? g.gtab_n : 0;
        a.need_logv = 0;
        if (!gb && c->opt_geomean_exact == 0)
            for (int k = 0; k < g.nseg; ++k)
                if (c->segs[(size_t)g.first + k].kind == CFMM_KIND_GEOMEAN) a.need_logv = 1;
        if (a.need_logv && sweep_lds_bytes(c->n_pad, 1, g.block, 1, a.gtab_n, a.v_shift == 4 ? 1 : 0) > 160 * 1024)
            a.need_logv = 0;   // the log-price row does not fit next to v and one bin copy: one logarithm per pool instead
        a.copies = bin_copies(c, g.block);
        a.compact = (c->opt_compact_trades != 0 && !gb) ? 1 : 0;
        a.partials = c->d_partials + (size_t)g.row_off * row_width(c);
        a.row_pitch = row_width(c);
        a.reverse = c->opt_alternate != 0 ? (int)(c->sweep_count & 1) : 0;
        a.arm_word = arm_word;
        a.arm_seq = arm_seq;
        a.arm_timeout = std::min<long long>(std::max<long long>(c->opt_arm_timeout_ms, 1), 10000) * 100000ll;   // ms -> ticks of the 100 MHz wall clock, at most 10 s
        // sharded (cfmm_set_peers): a rank whose host is late by less than the peer timeout must not lose the evaluation --
        // the other ranks' fold + gather launches wait that long for its granules anyway, so waiting for the price vector
        // equally long turns a stalled host into a slow evaluation on every rank instead of a failed route on all of them
        if (sharded) a.arm_timeout = std::max<long long>(a.arm_timeout, c->peer_timeout_ticks);
        a.flags = c->d_stage ? reinterpret_cast<unsigned long long*>(c->d_stage + c->flag_off) : nullptr;
        a.nt_stores = c->opt_stream_stores == 2 || (c->opt_stream_stores == 0 && c->touched_bytes > (int64_t)256 << 20) ? 1 : 0;
        if (direct) {
            a.direct = 1;
            a.direct_out = d_out;
            a.direct_host = ho;
            a.reverse = 0;   // two tiles at most, the whole market in one CU's L1: nothing for the alternation to reuse -- and every
                             // evaluation of such a market, fused or materialising, then returns the same bits at the same prices
        }
        const size_t lds = gb ? (size_t)(g.block / 64) * sizeof(double)
                              : sweep_lds_bytes(c->n_pad, a.copies, g.block, a.need_logv, a.gtab_n, a.v_shift == 4 ? 1 : 0);
        hipEvent_t ea = nullptr, eb = nullptr;
        if (timed) { // start/stop written by the command processor around this launch (hipExtLaunchKernel)
            ea = take_event(c);
            eb = take_event(c);
            if (!ea || !eb) ea = eb = nullptr;
        }
        // the kernel on the fast arithmetic: every pool constant of the launch inside the window (checked at upload), prices
        // staged as {v, rcp(v)} pairs, and the prices themselves inside it as far as the host knows
        bool fast = c->opt_fast_math != 0 && !gb && a.v_shift == 4 && price_window != kPricesOutside;
        for (int k = 0; k < g.nseg && fast; ++k) {
            const Segment& s = c->segs[(size_t)g.first + k];
            fast = s.fast_ok != 0 && !(s.kind == CFMM_KIND_GEOMEAN && c->opt_geomean_exact != 0);
        }
        // ... which the host knows for host-pointer sweeps and cfmm_route (pre-armed launches: armed_eval checks the prices before
        // it signals and cancels a launch whose prices turn out to be outside); a device-pointer sweep (prices unknown) gets the
        // kernel that carries both loops and decides per block from the prices it stages
        const int arith = !fast ? 0 : (price_window == kPricesUnknown && arm_seq == 0 && c->opt_dev_prices_in_window == 0) ? 2 : 1;
        const auto gbase_of = [&](const Segment& s) { return a.gtab_n ? s.gbase : -1; };   // -1: fees from the gamma array
        auto product_of = [&](const Segment& s) { return ProductPools{s.R, s.gamma, s.Ai, s.pk, gbase_of(s)}; };
        auto geomean_of = [&](const Segment& s) {
            return GeoMeanPools{s.R, s.w, s.gamma, s.Ai, s.eta, s.lR, (int)c->opt_geomean_exact, s.pk, gbase_of(s)};
        };
        auto univ3_of = [&](const Segment& s) {
            return UniV3Pools{s.pg, s.Ai, s.cur_a, s.cur_b, s.cur_c, s.curR, s.walk, s.ticks, s.thr,
                              c->opt_univ3_heads != 0 ? s.head : nullptr, s.has_walk, s.cp, s.pk, gbase_of(s)};
        };
        hipError_t e = hipSuccess;
        if (g.multi) {
            MultiArgs ma;
            std::memset(&ma, 0, sizeof ma);
            ma.nseg = g.nseg;
            ma.xcd_map = g.xcd_map ? 1 : 0;
            std::memcpy(ma.pattern, g.pattern, sizeof ma.pattern);
            std::memcpy(ma.rank, g.rank, sizeof ma.rank);
            for (int k = 0; k < kMaxMulti; ++k) ma.seg_w[k] = g.seg_w[k];
            ma.common = a;
            ma.common.gflow = gb ? c->d_flow : nullptr; // mode flag for the launcher; per-segment bases below
            for (int k = 0; k < g.nseg; ++k) {
                const Segment& s = c->segs[(size_t)g.first + k];
                MultiSeg& ms = ma.seg[k];
                ms.kind = s.kind;
                ms.m = s.m;
                ms.Delta = materialize ? c->d_delta + s.trade_off : nullptr;
                ms.Lambda = materialize ? c->d_lambda + s.trade_off : nullptr;
                ms.Over = materialize ? c->d_over + s.trade_off : nullptr;
                ms.gflow = gb ? c->d_flow + s.trade_off : nullptr;
                switch (s.kind) {
                case CFMM_KIND_PRODUCT: ms.pools.p = product_of(s); break;
                case CFMM_KIND_GEOMEAN: ms.pools.g = geomean_of(s); break;
                default: ms.pools.u = univ3_of(s); break;
                }
            }
            LaunchCfg cfg{g.block, g.grid, lds, arith, ea, eb};
            e = launch_multi(ma, cfg, materialize, c->stream);
        } else {
            const Segment& s = c->segs[(size_t)g.first];
            a.m = s.m;
            a.Delta = materialize ? c->d_delta + s.trade_off : nullptr;
            a.Lambda = materialize ? c->d_lambda + s.trade_off : nullptr;
            a.Over = materialize ? c->d_over + s.trade_off : nullptr;
            a.gflow = gb ? c->d_flow + s.trade_off : nullptr;
            LaunchCfg cfg{g.block, g.grid, lds, arith, ea, eb};
            switch (s.kind) {
            case CFMM_KIND_PRODUCT: e = launch_sweep(product_of(s), a, cfg, materialize, c->stream); break;
            case CFMM_KIND_GEOMEAN: e = launch_sweep(geomean_of(s), a, cfg, materialize, c->stream); break;
            default: e = launch_sweep(univ3_of(s), a, cfg, materialize, c->stream); break;
            }
        }
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "sweep launch failed: %s", hipGetErrorString(e));
        if (ea && eb) c->pending.push_back({ea, eb, 0});
    }
    c->last_host_out = host_out;
    hipEvent_t ra = nullptr, rb = nullptr;
    const bool bracket = gb || (c->rows_total == 0 && !sharded);   // several launches / a memset: bracket them with plain events
    if (timed && !direct) {
        ra = take_event(c);
        rb = take_event(c);
        if (!ra || !rb) ra = rb = nullptr;
        if (ra && bracket) HIP_TRY(c, hipEventRecord(ra, c->stream));
    }
    if (direct) {
        // (the sweep's only block has published {Ψ, acc} itself)
    } else if (sharded) {
        PeerSet ps;
        std::memset(&ps, 0, sizeof ps);
        const int64_t count = c->n + 1;
        for (size_t p = 0; p < c->peers.size(); ++p)
            ps.gran[p] = reinterpret_cast<unsigned long long*>(c->peers[p]);
        ps.world = (int)c->peers.size();
        ps.rank = c->peer_rank;
        ps.count = count;
        ps.seq = ++c->peer_seq;
        ps.timeout_ticks = c->peer_timeout_ticks;
        ps.host = ho;
        ps.arm = ArmWord{arm_word, arm_seq};
        // a world of ONE rank has nobody to exchange with: the plain fold (same columns, same order; measured 0.6 us per step
        // less than the gather launch with its 200-byte peer table -- N = 1 under a launcher then costs what plain N = 1 costs)
        hipError_t e = (ps.world == 1 && c->rows_total > 0 && !gb)
            ? launch_reduce(c->d_partials, (int)c->rows_total, c->n + 1, row_width(c), d_out, c->stream, bracket ? nullptr : ra,
                            bracket ? nullptr : rb, ho, ArmWord{arm_word, arm_seq})
            : launch_reduce_gather(c->d_partials, (int)c->rows_total, c->n + 1, row_width(c), d_out, c->stream, ps,
                                   bracket ? nullptr : ra, bracket ? nullptr : rb);
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "fold + gather launch failed: %s", hipGetErrorString(e));
    } else if (c->rows_total > 0) {
        hipError_t e;
        if (gb) { // pull Ψ per token over the incidence list, then fold the dual-scalar column
            e = launch_gather(c->d_chunks, c->d_entries, reinterpret_cast<const double*>(c->d_flow), c->d_chunk_sums,
                              c->n_chunks, c->d_tok_chunk_off, d_out, c->n, c->d_partials, (int)c->rows_total, c->stream);
        } else {
            e = launch_reduce(c->d_partials, (int)c->rows_total, c->n + 1, row_width(c), d_out, c->stream, bracket ? nullptr : ra,
                              bracket ? nullptr : rb, ho, ArmWord{arm_word, arm_seq});
        }
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "reduce launch failed: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(c, hipMemsetAsync(d_out, 0, (size_t)(c->n + 1) * sizeof(double), c->stream));
    }
    if (ra && rb) {
        if (bracket) HIP_TRY(c, hipEventRecord(rb, c->stream));
        c->pending.push_back({ra, rb, 1});
    }
    // the sweep and its fold ARE on the stream from here on: the context's bookkeeping says so whatever the collective does
    if (materialize) {
        c->have_trades = true;
        c->x_valid = false;
        c->trades_compact = (c->opt_compact_trades != 0 && !gb) ? 1 : 0;
    }
    ++c->sweep_count;
    if (rccl) {   // north_star: "RCCL all-reduce of Ψ and ∇g over xGMI per outer iteration" -- n + 1 doubles, on the same stream
        rc = rccl_all_reduce_out(c, d_out);
        if (rc != CFMM_OK) {
            c->have_out = false;   // d_out holds this rank's LOCAL {Ψ, acc}, not the market's: nothing may be read from it
            return rc;
        }
    }
    return CFMM_OK;
}

int check_prices(cfmm_ctx* c, const double* v)
{
    if (!v) return fail(c, CFMM_ERR_INVALID_ARG, "v is null");
    for (int j = 0; j < c->n; ++j)
        if (!(v[j] > 0.0) || !std::isfinite(v[j]))
            return fail(c, CFMM_ERR_INVALID_ARG, "v[%d] must be finite and > 0 (src/cfmms.jl:129)", j);
    return CFMM_OK;
}

// every price in [2^-kFastExp, 2^kFastExp] (sweep.h): the host's half of the fast kernels' precondition
bool prices_in_fast_window(const double* v, int n)
{
    for (int j = 0; j < n; ++j) {
        uint64_t bits;
        std::memcpy(&bits, v + j, sizeof bits);
        const int e = (int)((bits >> 52) & 0x7ff);
        if (e < 1023 - kFastExp || e > 1023 + kFastExp) return false;
    }
    return true;
}

// Why did a sweep deliver NaN?  The blocks report through the sticky word (sweep.h kFlagWindow / kFlagGaveUp).
// Returns the reported bits among `mask` and clears exactly those (a report nobody asked about stays for whoever does).
unsigned long long take_flags(cfmm_ctx* c, unsigned long long mask)
{
    if (!c->h_stage) return 0;
    // atomic on the mapped word: launches queued behind the current one may report (system-scope fetch_or over PCIe) while the
    // host clears -- a plain read-modify-write could lose their bit
    unsigned long long* w = reinterpret_cast<unsigned long long*>(c->h_stage + c->flag_off);
    if ((__atomic_load_n(w, __ATOMIC_ACQUIRE) & mask) == 0) return 0;
    return __atomic_fetch_and(w, ~mask, __ATOMIC_ACQ_REL) & mask;
}

// First half of a host-pointer sweep: stage v, enqueue the evaluation (asynchronous).
int host_sweep_begin(cfmm_ctx* c, const double* v, bool materialize)
{
    HIP_TRY(c, hipSetDevice(c->device));
    (void)take_flags(c, kFlagWindow);     // a stale report must not label a later overflow of THIS call
    std::memcpy(c->h_stage, v, (size_t)c->n * sizeof(double));
    if (materialize) {
        c->trade_v.assign(v, v + c->n);   // the prices the device trades belong to (update_reserves!)
        // find_arb!(r, v) is a function of v alone (test/arb.jl:16 compares its netflows exactly): a materialising host call
        // always walks forwards, whatever ran before; the alternation (option "alternate") restarts behind it.
        c->sweep_count = 0;
    }
    double* h_out = c->h_stage + c->n;
    const bool zero_copy = c->opt_zero_copy != 0 && c->d_stage != nullptr;
    // v: small vectors are read by every block straight from the mapped pinned buffer (the PCIe
    // round trip hides behind the first tile's pool loads); larger ones go through one H2D copy.
    const double* v_src = c->d_stage;
    if (!zero_copy || c->n > 1024 || global_bins(c)) {
        HIP_TRY(c, hipMemcpyAsync(c->d_v, c->h_stage, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        v_src = c->d_v;
    }
    // {Ψ, acc}: as output granules in the mapped pinned buffer when it can (polled by host_sweep_end), else d_out + a copy
    const bool want_host_out = zero_copy && c->opt_host_flag != 0;
    int rc = enqueue_sweep(c, v_src, c->d_out, materialize, want_host_out, 0,
                           prices_in_fast_window(v, c->n) ? kPricesInWindow : kPricesOutside);
    if (rc != CFMM_OK) return rc;
    if (!c->last_host_out)
        HIP_TRY(c, hipMemcpyAsync(h_out, c->d_out, (size_t)(c->n + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return CFMM_OK;
}

namespace {

// Output granules of the sweep with sequence number `seq` (fold_finish): true once all 2(n+1) carry its tag; the
// doubles are then reassembled into the {Ψ, acc} slots of the staging buffer.
bool granules_arrived(cfmm_ctx* c, uint64_t seq)
{
    const unsigned long long tag = seq % 0xffffffffull + 1ull;
    const volatile unsigned long long* g = reinterpret_cast<const volatile unsigned long long*>(c->h_stage + c->gran_off);
    const int n1 = c->n + 1;
    if ((g[2 * n1 - 1] >> 32) != tag || (g[0] >> 32) != tag) return false;   // cheap rejects: last and first granule
    double* h_out = c->h_stage + c->n;
    for (int j = 0; j < n1; ++j) {
        const unsigned long long a = g[2 * j], b = g[2 * j + 1];
        if ((a >> 32) != tag || (b >> 32) != tag) return false;
        const unsigned long long bits = (a & 0xffffffffull) | (b << 32);
        std::memcpy(h_out + j, &bits, sizeof(double));
    }
    return true;
}

// {Ψ, acc} have arrived in the pinned staging buffer: take them over into last_out.
int take_host_out(cfmm_ctx* c)
{
    const double* h_out = c->h_stage + c->n;
    c->last_out.assign(h_out, h_out + c->n + 1);
    for (int j = 0; j <= c->n; ++j)
        if (!std::isfinite(c->last_out[(size_t)j])) {
            c->have_out = false;
            (void)hipStreamSynchronize(c->stream);
            const unsigned long long why = take_flags(c, kFlagWindow);
            if (why & kFlagWindow)     // (a fast kernel met prices the host had vouched for: cannot happen unless v changed under the call)
                return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}: a price lies outside the window of the fast arithmetic "
                                               "(the price vector changed while the call ran?)");
            if (c->rccl_comm)
                return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}[%d] after the RCCL all-reduce: a shard overflowed (on some rank)", j);
            if (!c->peers.empty())
                return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}[%d]: the peer all-reduce timed out (a rank did not "
                                               "publish within CFMM_AMD_PEER_TIMEOUT_S) or a shard overflowed", j);
            return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}[%d]: pool arithmetic overflowed", j);
        }
    c->have_out = true;
    return CFMM_OK;
}

} // namespace

// Second half: wait for {Ψ, acc} to be on the host and take them over into last_out.
int host_sweep_end(cfmm_ctx* c)
{
    bool seen = false;
    if (c->last_host_out) {
        // the fold blocks wrote {Ψ, acc} as self-validating granules into this pinned buffer: poll them instead of
        // waiting for the kernel's end-of-pipe processing and its completion signal.  Bounded; falls back to a stream wait.
        const uint64_t want = c->out_seq;
        for (long spins = 0; spins < 400000000L; ++spins) {
            if (granules_arrived(c, want)) { seen = true; break; }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->last_host_out && !granules_arrived(c, c->out_seq)) {
            c->have_out = false;
            return fail(c, CFMM_ERR_STATE, "the sweep retired without delivering its outputs");
        }
    }
    return take_host_out(c);
}

int single_host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    int rc = host_sweep_begin(c, v, materialize);
    return rc != CFMM_OK ? rc : host_sweep_end(c);
}

int host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    int rc = check_prices(c, v);
    if (rc != CFMM_OK) return rc;
    return c->shards.empty() ? single_host_sweep(c, v, materialize) : multi_host_sweep(c, v, materialize);
}

// ---- pre-armed evaluations (cfmm_route; sweep.h SweepArgs::arm_word) ---------------------------
// cfmm_route's evaluations are strictly sequential (the solver needs {Ψ, acc} of v_k to choose v_k+1), so the ~5 us
// between "v is ready" and "the kernel's first wavefront runs" (launch call, doorbell, command processor, dispatch)
// sit on the critical path of every evaluation.  Armed operation takes them off it: evaluation k+1 -- sweep and
// fold -- is enqueued right after evaluation k has been signalled, becomes resident when k's fold retires, issues its
// first pool loads, clears its LDS bins and then polls a word in device memory; the host writes v_k+1 and the word
// straight into (fine-grained) device memory through the PCIe BAR.  Measured on the handoff alone
// (scripts/native/handoff.hip, profiles/r02_handoff.txt): 11.2 us launch-when-ready vs 6.3 us armed.
// The one launch left over when the solver stops is cancelled through the same word.
//
// Sharded contexts arm as well (round 3).  cfmm_set_peers: every rank arms its own launches; the ranks run the same
// solver on bit-identical {Ψ, acc}, so they signal and cancel the same evaluations, and a cancelled fold + gather
// publishes nothing (its sequence number is reused).  Multi-device parents: the calling thread writes v into every
// shard's BAR window, then polls every shard's output granules and sums them in device order -- no worker threads on
// this path.  NOT armed: parents whose shards share a device, and therefore the ranks-on-one-GPU test set-ups (they
// switch the option off): a polling launch occupies its CUs, and a shard of the SAME evaluation queued behind it on
// the same device would wait for an evaluation that waits for it.
namespace {

bool can_arm_single(cfmm_ctx* c)
{
    if (c->opt_armed == 0 || !c->d_arm || c->opt_zero_copy == 0 || !c->d_stage || c->opt_host_flag == 0 ||
        c->opt_time_kernels != 0 || c->n > 1024 || c->stream != c->own_stream || global_bins(c) || c->rccl_comm != nullptr)
        return false;
    return ensure_geometry(c) == CFMM_OK && c->rows_total > 0;
}

void armed_write(cfmm_ctx* c, const double* v, uint64_t word)
{
    if (v) std::memcpy(c->d_arm, v, (size_t)c->n * sizeof(double));   // write-combining stores through the BAR
    __builtin_ia32_sfence();                                            // v before the word (WC buffers flush out of order)
    *reinterpret_cast<volatile unsigned long long*>(c->d_arm + c->n_pad) = word;
    __builtin_ia32_sfence();                                            // and out now
}

int armed_enqueue(cfmm_ctx* c)
{
    const uint64_t seq = ++c->arm_seq;
    int rc = enqueue_sweep(c, c->d_arm, c->d_out, false, true, seq);
    if (rc != CFMM_OK) return rc;
    c->arm_tag = c->out_seq;
    c->arm_pending = true;
    return CFMM_OK;
}

void armed_cancel_single(cfmm_ctx* c)
{
    if (!c->arm_pending) return;
    armed_write(c, nullptr, c->arm_seq | kArmCancel);
    c->arm_pending = false;
    --c->sweep_count;   // the cancelled launch swept nothing: later sweeps keep the tile directions of an unarmed run
    if (!c->peers.empty()) --c->peer_seq;   // ... and the peers' sequence number is reused (every rank cancels the same launch)
}

// Signal the waiting launch (enqueuing it first if none is waiting) and enqueue the next one behind it.
// `signalled` = an evaluation is now running whose outputs carry the sequence number `want` (a failure after that point
// concerns the NEXT launch only).
int armed_signal(cfmm_ctx* c, const double* v, uint64_t& want, bool& signalled)
{
    signalled = false;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->arm_pending) {
        int rc = armed_enqueue(c);
        if (rc != CFMM_OK) return rc;
    }
    want = c->arm_tag;
    armed_write(c, v, c->arm_seq);
    c->arm_pending = false;
    signalled = true;
    return armed_enqueue(c);   // evaluation k+1 goes out while k runs
}

// The signalled evaluation is lost: cancel the launch queued behind it, drain the stream, and put the bookkeeping back
// to where an unarmed run would be (the lost launch swept nothing useful: the retry takes its tile direction).  On a
// cfmm_set_peers context a lost evaluation is fatal -- the other ranks have moved on with it.
void armed_lost(cfmm_ctx* c, bool& retry)
{
    armed_cancel_single(c);
    (void)hipStreamSynchronize(c->stream);
    c->have_out = false;
    retry = c->peers.empty();
    if (retry) --c->sweep_count;
}

// Wait for the signalled evaluation's granules.  CFMM_ERR_STATE with `lost` set: the device never delivered (host
// stalled past the arm timeout between two evaluations, or the device never saw the word): the caller may retry unarmed.
// The bound: the device-side wait of the launch (arm_timeout_ms; on cfmm_set_peers contexts at least the peer timeout,
// see enqueue_sweep) plus, on peers contexts, the time its fold + gather may legitimately wait for a slower rank.
int armed_wait(cfmm_ctx* c, uint64_t want, bool& lost)
{
    lost = false;
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    const double arm_s = 1e-3 * (double)std::min<int64_t>(std::max<int64_t>(c->opt_arm_timeout_ms, 1), 10000);
    const double peer_s = 1e-8 * (double)c->peer_timeout_ticks;
    const double limit = c->peers.empty() ? 2.0 * arm_s + 1.0 : std::max(arm_s, peer_s) + peer_s + 2.0;
    for (long spins = 0;; ++spins) {
        if (granules_arrived(c, want)) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) {
        armed_lost(c, lost);
        (void)take_flags(c, kFlagGaveUp);
        return fail(c, CFMM_ERR_STATE, c->peers.empty()
                        ? "armed evaluation did not complete (the device never saw its price vector)"
                        : "armed evaluation did not complete within the arm + peer timeouts (this rank's launch never saw its "
                          "price vector, or a peer never published)");
    }
    // NaN in the dual column: the blocks say why through the sticky report word.  A launch that gave up waiting for its
    // prices is a LOST evaluation (retried unarmed by the caller); anything else -- arithmetic overflow, a peer gather that
    // timed out -- is reported by take_host_out as what it is.
    const double acc = c->h_stage[2 * c->n];
    if (acc != acc) {
        const unsigned long long why = take_flags(c, kFlagGaveUp);
        if (why & kFlagGaveUp) {
            armed_lost(c, lost);
            return fail(c, CFMM_ERR_STATE, "armed evaluation gave up waiting for its price vector (host stalled longer than the arm timeout)");
        }
    }
    int rc = take_host_out(c);
    if (rc != CFMM_OK) armed_cancel_single(c);
    return rc;
}

} // namespace

bool can_arm(cfmm_ctx* c)
{
    if (!is_parent(c)) return can_arm_single(c);
    if (c->opt_armed == 0 || !c->shards_distinct) return false;
    for (cfmm_ctx* child : c->shards)
        if (!child->segs.empty() && !can_arm_single(child)) return false;
    return true;
}

void armed_cancel(cfmm_ctx* c)
{
    if (!c) return;
    if (!is_parent(c)) return armed_cancel_single(c);
    for (cfmm_ctx* child : c->shards) armed_cancel_single(child);
}

// One fused evaluation at v through the armed launches.  A lost evaluation (see armed_wait) is retried ONCE through the
// launch-when-ready path before the call fails: a host that was paused between two evaluations (debugger, SIGSTOP,
// oversubscription) costs a retry, not the route.  *lost_out tells the caller (cfmm_route stops arming for the rest of
// the call: whatever stalled the hand-over once -- e.g. another process's launches holding this GPU's CUs -- may do so again).
int armed_eval(cfmm_ctx* c, const double* v, bool* lost_out)
{
    if (lost_out) *lost_out = false;
    int rc = check_prices(c, v);
    if (rc != CFMM_OK) return rc;
    if (!prices_in_fast_window(v, c->n)) {
        // the waiting launch runs the fast arithmetic (its kernel was chosen before these prices existed): cancel it and
        // evaluate launch-when-ready on the full-range kernels; the caller stops arming (prices this extreme stay extreme)
        armed_cancel(c);
        if (lost_out) *lost_out = true;
        return is_parent(c) ? multi_host_sweep(c, v, false) : single_host_sweep(c, v, false);
    }
#ifdef CFMM_TEST_HOOKS
    if (c->opt_debug_stall_ms > 0 && c->arm_pending) {   // test hook: the host "stalls" once while a launch waits for its prices
        std::this_thread::sleep_for(std::chrono::milliseconds(c->opt_debug_stall_ms));
        c->opt_debug_stall_ms = 0;
    }
#endif
    if (!is_parent(c)) {
        uint64_t want = 0;
        bool signalled = false;
        const int rc_next = armed_signal(c, v, want, signalled);
        if (!signalled) return rc_next;
        bool lost = false;
        rc = armed_wait(c, want, lost);
        if (rc != CFMM_OK && lost) {
            armed_cancel_single(c);
            if (lost_out) *lost_out = true;
            return single_host_sweep(c, v, false);
        }
        if (rc != CFMM_OK) return rc;
        return rc_next;
    }
    const int nd = (int)c->shards.size();
    std::vector<uint64_t> want((size_t)nd, 0);
    std::vector<int> rcs((size_t)nd, CFMM_OK);
    std::vector<char> running((size_t)nd, 0);
    for (int d = 0; d < nd; ++d)
        if (!c->shards[(size_t)d]->segs.empty()) {
            bool signalled = false;
            rcs[(size_t)d] = armed_signal(c->shards[(size_t)d], v, want[(size_t)d], signalled);
            running[(size_t)d] = signalled ? 1 : 0;
        }
    bool any_lost = false;
    int first_err = CFMM_OK, err_shard = -1;
    for (int d = 0; d < nd; ++d) {
        cfmm_ctx* child = c->shards[(size_t)d];
        if (child->segs.empty()) continue;
        if (!running[(size_t)d]) {
            if (first_err == CFMM_OK) { first_err = rcs[(size_t)d]; err_shard = d; }
            continue;
        }
        bool lost = false;
        const int rw = armed_wait(child, want[(size_t)d], lost);
        any_lost = any_lost || lost;
        running[(size_t)d] = (rw == CFMM_OK && !lost) ? 2 : 1;   // 2: this shard delivered the evaluation
        const int r = rw != CFMM_OK ? rw : rcs[(size_t)d];
        if (r != CFMM_OK && first_err == CFMM_OK) { first_err = r; err_shard = d; }
    }
    if (any_lost) {
        armed_cancel(c);
        for (int d = 0; d < nd; ++d)   // shards that did deliver repeat the evaluation too: it takes the same tile direction
            if (running[(size_t)d] == 2) --c->shards[(size_t)d]->sweep_count;
        if (lost_out) *lost_out = true;
        return multi_host_sweep(c, v, false);
    }
    if (first_err != CFMM_OK) {
        armed_cancel(c);
        c->have_out = false;
        return fail(c, first_err, "shard %d: %s", err_shard, c->shards[(size_t)err_shard]->err.c_str());
    }
    c->last_out.assign((size_t)c->n + 1, 0.0);   // the all-reduce: shard order, on the host
    for (int d = 0; d < nd; ++d) {
        const cfmm_ctx* child = c->shards[(size_t)d];
        if (child->segs.empty()) continue;
        for (int j = 0; j <= c->n; ++j) c->last_out[(size_t)j] += child->last_out[(size_t)j];
    }
    c->have_out = true;
    c->have_trades = false;
    return CFMM_OK;
}

} // namespace cfmm

extern "C" {

int cfmm_find_arb(cfmm_ctx* c, const double* v)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    return host_sweep(c, v, true);
}

int cfmm_eval(cfmm_ctx* c, const double* v, double* psi_out, double* acc_out)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int rc = host_sweep(c, v, false);
    if (rc != CFMM_OK) return rc;
    c->have_trades = false; // trades on the device no longer correspond to the latest v
    if (psi_out) std::memcpy(psi_out, c->last_out.data(), (size_t)c->n * sizeof(double));
    if (acc_out) *acc_out = c->last_out[(size_t)c->n];
    return CFMM_OK;
}

int cfmm_netflows(cfmm_ctx* c, double* psi)
{
    if (!c || !psi) return CFMM_ERR_INVALID_ARG;
    if (!c->have_out) return fail(c, CFMM_ERR_STATE, "no sweep has been run yet");
    std::memcpy(psi, c->last_out.data(), (size_t)c->n * sizeof(double));
    return CFMM_OK;
}

int cfmm_dual_value(cfmm_ctx* c, double* acc)
{
    if (!c || !acc) return CFMM_ERR_INVALID_ARG;
    if (!c->have_out) return fail(c, CFMM_ERR_STATE, "no sweep has been run yet");
    *acc = c->last_out[(size_t)c->n];
    return CFMM_OK;
}

int cfmm_sweep_dev(cfmm_ctx* c, const double* d_v, double* d_out, int materialize)
{
    if (!c || !d_v || !d_out) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_sweep_dev");
    c->have_out = false; // results live on the device; the host copy is stale
    if (materialize) c->trade_v.clear();   // the library has not seen these prices
    // The library cannot see these prices: the launch carries both arithmetics and every block picks from the prices it
    // stages (sweep_kernels.hip kArithAuto) -- prices outside [2^-150, 2^150], NaN included, take the full-range loop and
    // propagate like the reference's arithmetic.  (Round 4 launched the fast kernels on trust and reported a refusal on a
    // LATER call: ADVICE r4.)
    return enqueue_sweep(c, d_v, d_out, materialize != 0, false, 0, kPricesUnknown);
}

} // extern "C"
