// cfmm_abi.hip -- implementation of the C ABI declared in include/cfmm_amd.h.
//
// Host-side only: owns the HBM-resident pool store (SoA-of-pairs per pool family), the trade
// buffers, the partial-row scratch and a pinned staging area, validates what the reference's
// constructors validate (src/cfmms.jl:76-90) plus what its kernels silently assume
// (src/cfmms.jl:129 "Assumes that v > 0 and γ > 0"), and launches sweep_kernels.hip.
// There is no CPU fallback anywhere in this file: without a gfx950 device every entry point
// that needs one fails with CFMM_ERR_HIP.

#include "../../include/cfmm_amd.h"
#include "sweep.h"
#include "lbfgsb.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace cfmm;

namespace {

thread_local std::string g_create_error = "";

// Kernel arguments in device memory: measured 22.2 vs 24.9 us per config-3 step and 7.0 vs 9.1 us per config-2
// step against host-memory kernargs (r01).  The HIP runtime reads the variable when it initialises, so it is
// set when this library is loaded -- unless the caller has decided otherwise (an existing value is kept).
__attribute__((constructor)) void cfmm_default_environment() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

struct Segment {
    int kind = 0;
    int64_t m = 0;
    int64_t trade_off = 0; // first row of this segment in the trade buffers
    int64_t n_ticks_total = 0;
    int deep = 0; // univ3: longest walk list exceeds kDeepWalk -> wavefront-cooperative kernel
    // device arrays (owned)
    double2* R = nullptr;
    double2* w = nullptr;
    double* eta = nullptr;   // geomean: η = w1/w2
    double2* lR = nullptr;   // geomean: {Q1, Q2}, the v-independent constants of the log-space exponents (GeoMeanLogOps)
    double* gamma = nullptr;
    int2* Ai = nullptr;
    double2* pg = nullptr;
    double* cp = nullptr;    // univ3: current_price alone (packed records)
    int has_walk = 1;        // univ3: some pool has a tick beyond its current one
    double2* cur_a = nullptr;
    double2* cur_b = nullptr;
    double* cur_c = nullptr;
    double2* curR = nullptr;
    int4* walk = nullptr;
    double2* ks = nullptr;
    double2* dt = nullptr;
    double* rout = nullptr;
    // launch geometry (decided at upload)
    int block = kSmallBlock;
    int grid = 0;
    int unroll = 1;
    int64_t row_off = 0; // first partial row
    PackedFeeTok* pk = nullptr;     // two-coin families: {i1 | i2 << 16, fee-table index} per pool, or null (too many distinct fees)
    std::vector<double> gvals;      // the segment's distinct fees, in order of first appearance (index = PackedFeeTok::gidx)
    int gbase = 0;                  // first entry of this segment in its launch's fee table (ensure_geometry)
    std::vector<int32_t> h_ai; // host copy of Ai: large-market mode (incidence build) and UniV3 segments
    // UniV3 only: the pool definitions as uploaded (update_reserves! moves current_price and re-derives the constants)
    std::vector<double> h_cp, h_gamma, h_lt, h_liq;
    std::vector<int64_t> h_tick_off;
};

// A launch: either one segment (sweep_kernel) or up to kMaxMulti segments fused (sweep_multi).
struct Group {
    int first = 0, nseg = 1;
    bool multi = false;
    int block = kSmallBlock;
    int grid = 0;       // total blocks of the launch
    int64_t row_off = 0;
    int gtab_n = 0;     // entries of this launch's fee table (0: its segments use the plain gamma / Ai arrays)
    // XCD-aware weighted block -> segment map of a fused launch (see sweep_multi); xcd_map == false: block b -> segment b % nseg
    bool xcd_map = false;
    unsigned char pattern[32] = {0}, rank[32] = {0};
    int seg_w[kMaxMulti] = {0};
};

} // namespace

namespace {
struct Workers {
    // One persistent thread per shard >= 1 (shard 0 runs on the calling thread).  A call publishes
    // {v, materialize} and bumps `go`; workers spin briefly on it (an L-BFGS-B evaluation follows
    // the previous one within microseconds), then sleep on the condition variable.
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> go{0};
    std::atomic<int> pending{0};
    std::atomic<int> sleepers{0};
    std::atomic<bool> quit{false};
    const double* v = nullptr;
    bool materialize = false;
    std::vector<int> rc;
};

} // namespace

struct cfmm_ctx {
    int device = 0;
    int n = 0;
    int n_pad = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::vector<Segment> segs;
    std::vector<Group> groups;
    int64_t m_total = 0;
    int64_t rows_total = 0;

    double* d_v = nullptr;        // [n]
    double* d_out = nullptr;      // [n+1]
    double* d_partials = nullptr; // [rows_cap][n+1]
    int64_t rows_cap = 0;
    // trade buffers [trade_cap] each.  Compact layout (option "compact_trades", default): d_delta holds ONE 16-byte
    // record per pool, d_lambda / d_over the four values of the rare pools that trade in both directions (sweep.h
    // SweepArgs); plain layout: d_delta = {Δ₁, Δ₂}, d_lambda = {Λ₁, Λ₂}.  d_xdelta / d_xlambda: expanded copies, only for
    // cfmm_trades_dev.
    double2* d_delta = nullptr;
    double2* d_lambda = nullptr;
    double2* d_over = nullptr;
    double2* d_xdelta = nullptr;
    double2* d_xlambda = nullptr;
    int64_t x_cap = 0;
    int trades_compact = 0;       // layout of the trades currently on the device
    std::vector<double2> h_rec, h_ovA, h_ovB;   // host scratch of cfmm_get_trades*
    int64_t trade_cap = 0;
    // large-market mode (n > kMaxLdsTokens): token -> (pool, side) incidence and flow scratch
    double2* d_flow = nullptr;    // [m_total] {Λ₁−Δ₁, Λ₂−Δ₂}
    int* d_entries = nullptr;     // [2·m_total] flat flow indices grouped by token
    int2* d_chunks = nullptr;     // [n_chunks] {begin, end} into d_entries
    int* d_tok_chunk_off = nullptr; // [n+1]
    double* d_chunk_sums = nullptr; // [n_chunks]
    int n_chunks = 0;
    // sharded operation (cfmm_set_peers): every host-pointer sweep ends with the one-shot peer
    // all-reduce of peer_allreduce.hip, so eval / find_arb / route return GLOBAL {Ψ, acc}
    std::vector<uint64_t> peers;  // device addresses of all ranks' symmetric buffers
    int peer_rank = 0;
    uint64_t peer_seq = 0;
    long long peer_timeout_ticks = 3000000000ll;   // 30 s of wall_clock64() at 100 MHz (CFMM_AMD_PEER_TIMEOUT_S)
    double* h_stage = nullptr;    // pinned + device-mapped: [n] v in, [n+1] out, then one uint64 completion flag
    double* d_stage = nullptr;    // device address of h_stage
    double* d_gtab = nullptr;     // [groups][kMaxFeeTable] fee tables of the launches (packed pool records)
    size_t gtab_cap = 0;
    unsigned* d_sync = nullptr;   // [kSyncWords] arrival counters of the in-launch fold (zero between launches)
    // pre-armed evaluations of cfmm_route (sweep.h SweepArgs::arm_word): [n_pad] v, then the word, in FINE-GRAINED
    // device memory that the host writes through the PCIe BAR (null: no large BAR, or the self-check failed)
    double* d_arm = nullptr;
    uint64_t arm_seq = 0;         // sequence number of the latest armed launch
    bool arm_pending = false;     // an armed launch is enqueued and has not been signalled or cancelled yet
    uint64_t arm_flag = 0;        // completion-flag value that launch will raise
    uint64_t flag_seq = 0;        // host-visible completion flag: value the NEXT flagged sweep will raise
    bool last_inline = false;     // the latest enqueue_sweep folded inside the sweep launch
    bool last_flagged = false;    // ... and raises the host flag (the caller may poll it instead of the stream)
    size_t gran_off = 0;          // first output granule in h_stage / d_stage (doubles)
    bool last_granules = false;   // ... in the form of self-validating output granules (no flag word: the data carries flag_seq)
    std::vector<double> last_out; // psi..., acc of the latest host-pointer sweep
    std::vector<double> trade_v;  // v of the latest MATERIALISING host-pointer sweep (empty: none / device-pointer sweep)
    bool have_out = false;
    bool have_trades = false;
    bool geometry_dirty = true;

    // options
    int64_t opt_max_grid = 0;    // 0 = auto (512 fat blocks / 2048 small blocks)
    int64_t opt_block = 0;       // 0 = auto, else kSmallBlock or kBigBlock
    int64_t opt_unroll = 0;      // 0 = auto
    int64_t opt_bin_copies = 0;  // 0 = auto, 1 = one shared copy, 2 = one copy per wavefront
    int64_t opt_time_kernels = 0;
    int64_t opt_nt_stores = 2;     // trade stores: 0 plain, 1 non-temporal, 2 write-through (default; -1..-2 us per 1M-pool sweep)
    int64_t opt_geomean_exact = 0; // 1: pow-based reference-order forms instead of log-space
    int64_t opt_fuse_segments = 1; // 1: sweep all pool families in one launch (sweep_multi)
    int64_t opt_univ3_coop = -1;   // -1 auto (by walk-list length), 0 lane-per-pool only, 1 wavefront-cooperative
    int64_t opt_zero_copy = 1;     // 1: host-pointer calls read v / write Ψ through mapped pinned memory
    int64_t opt_spin_wait = 0;     // 1: host-pointer calls busy-poll the stream (measured: no gain over hipStreamSynchronize)
    int64_t opt_wave_split = 0;    // 1: fused launches deal each block's wavefronts to the pool families (every block sweeps every segment)
    int64_t opt_xcd_map = 1;       // fused launches: 1 = XCD-aware block -> segment map weighted by pools x cost per pool,
                                   // 2 = XCD-aware with equal cost per pool, 0 = block b -> segment b % nseg
    int64_t opt_cost_geomean = 10; // cost of a GeometricMean / UniV3 evaluation in tenths of a ProductTwoCoin one (10 = blocks in
    int64_t opt_cost_univ3 = 10;   // proportion to pool counts: measured best once sweeps alternate direction; 18 / 14 before)
    int64_t opt_compact_trades = 1; // 1: a materialising sweep writes one 16-byte trade record per pool (+ overflow rows for the rare
                                   //    pools trading in both directions) instead of 32 bytes; lossless, see sweep.h SweepArgs
    int64_t opt_pack = 1;          // 1: sweeps read the packed fee + token record (24 / 48 B per pool instead of 32 / 56) when the
                                   //    launch's distinct fees fit the LDS fee table
    int64_t opt_alternate = 1;     // 1: consecutive sweeps walk the tiles in alternating directions (L2 reuse across sweeps);
                                   //    results of two sweeps at the same v then agree to rounding, not bit for bit
    uint64_t sweep_count = 0;
    int64_t opt_inline_fold = 0;   // 1: partial rows are folded inside the sweep launch (single-launch evaluations, n <= kMaxFoldTokens);
                                   //    measured 1-3 us per step SLOWER than the separate fold launch, kept as an option
    int64_t opt_host_granules = 1; // 1: a flagged sweep returns {psi, acc} to the host as self-validating 8-byte granules (tag + half
                                   //    of a double) that the host re-reads until complete, instead of outputs + drain + ticket + flag
    int64_t opt_armed = 1;         // 1: cfmm_route enqueues evaluation k+1 while evaluation k runs; its blocks wait on the device for
                                   //    the host to write v through the PCIe BAR (hides the launch latency: -4..-5 us per evaluation)
    int64_t opt_arm_timeout_ms = 2000; // bound of that wait
    int64_t opt_host_flag = 1;     // 1: zero-copy host-pointer sweeps end by raising a flag in mapped host memory that the
                                   //    caller polls, instead of waiting for the stream (saves the end-of-kernel + signal path)

    // kernel timing
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Pending { hipEvent_t a, b; int what; };
    std::vector<Pending> pending;
    int64_t t_sweep_n = 0, t_reduce_n = 0;
    double t_sweep_ms = 0, t_reduce_ms = 0;

    // single-process multi-device parent (cfmm_ctx_create_multi): shards non-empty, no device state of its own
    std::vector<cfmm_ctx*> shards;
    struct ParentSeg { int kind; int64_t m; int64_t trade_off; };
    std::vector<ParentSeg> psegs;          // one per cfmm_pools_add_* call with m > 0
    std::unique_ptr<Workers> workers;
    int64_t opt_multi_threads = 1;

    mutable std::string err = "";
};

namespace {

int fail(const cfmm_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(ctx, CFMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));            \
    } while (0)

template <class T>
int upload(cfmm_ctx* c, T** dst, const void* src, size_t count)
{
    *dst = nullptr;
    if (count == 0) return CFMM_OK;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(dst), count * sizeof(T)));
    HIP_TRY(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return CFMM_OK;
}

void free_segment(Segment& s)
{
    (void)hipFree(s.R); (void)hipFree(s.w); (void)hipFree(s.gamma); (void)hipFree(s.Ai);
    (void)hipFree(s.eta); (void)hipFree(s.lR); (void)hipFree(s.pk);
    (void)hipFree(s.cur_a); (void)hipFree(s.cur_b); (void)hipFree(s.cur_c); (void)hipFree(s.curR);
    (void)hipFree(s.pg); (void)hipFree(s.cp); (void)hipFree(s.walk); (void)hipFree(s.ks); (void)hipFree(s.dt); (void)hipFree(s.rout);
    s = Segment{};
}

bool finite_pos(double x) { return std::isfinite(x) && x > 0.0; }

bool global_bins(const cfmm_ctx* c);

// Packed fee + token record of a two-coin segment (sweep.h PackedFeeTok): built when the token ids fit 16 bits and the
// segment has at most kMaxFeeTable distinct fees; otherwise the segment keeps pk == null and sweeps read gamma / Ai.
int build_packed(cfmm_ctx* c, Segment& s, int64_t m, const double* gamma, const int32_t* Ai)
{
    s.pk = nullptr;
    s.gvals.clear();
    if (global_bins(c) || c->n > 65536 || m == 0) return CFMM_OK;
    std::vector<PackedFeeTok> pk((size_t)m);
    std::vector<double> vals;
    uint64_t last_bits = 0;
    uint32_t last_idx = 0;
    bool have_last = false;
    for (int64_t i = 0; i < m; ++i) {
        uint64_t bits;
        std::memcpy(&bits, &gamma[i], sizeof bits);
        uint32_t idx;
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
                if ((int)vals.size() == kMaxFeeTable) return CFMM_OK;   // too many fee tiers: stay unpacked
                vals.push_back(gamma[i]);
            }
            last_bits = bits; last_idx = idx; have_last = true;
        }
        pk[(size_t)i].tok = (uint32_t)Ai[2 * i] | ((uint32_t)Ai[2 * i + 1] << 16);
        pk[(size_t)i].gidx = idx;
    }
    int rc = upload(c, &s.pk, pk.data(), (size_t)m);
    if (rc != CFMM_OK) return rc;
    s.gvals.swap(vals);
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

// Launch geometry for a segment of m pools.  Small markets: 256-thread blocks, one tile each
// (enough blocks to cover 256 CUs).  Large markets: 1024-thread blocks, at most two per CU, each
// striding over many tiles -- this keeps the number of partial rows (and the fold kernel) small.
void plan_segment(const cfmm_ctx* c, Segment& s)
{
    int U = (int)c->opt_unroll;
    if (U != 1 && U != 2 && U != 4) {
        U = 1; // measured on MI355X (scripts/tune.py): one pool per lane per tile is fastest for
               // every family; 16 wavefronts per CU already cover the HBM latency
    }
    s.unroll = U;
    const int64_t tiles_small = std::max<int64_t>(1, (s.m + (int64_t)kSmallBlock * U - 1) / ((int64_t)kSmallBlock * U));
    const bool small = c->opt_block == kSmallBlock || (c->opt_block == 0 && tiles_small <= 512);
    if (small) {
        // one tile per block; from ~64k pools on, 512-thread blocks (half the partial rows) measured
        // faster than 256-thread ones (config2: 6.6 vs 7.2 us per step)
        s.block = (c->opt_block == 0 && s.m >= 65536) ? kMidBlock : kSmallBlock;
        const int64_t tiles = std::max<int64_t>(1, (s.m + (int64_t)s.block * U - 1) / ((int64_t)s.block * U));
        s.grid = (int)std::min<int64_t>(tiles, c->opt_max_grid > 0 ? c->opt_max_grid : 2048);
    } else {
        s.block = c->opt_block == kMidBlock ? kMidBlock : kBigBlock;
        const int64_t tiles = std::max<int64_t>(1, (s.m + (int64_t)s.block * U - 1) / ((int64_t)s.block * U));
        s.grid = (int)std::min<int64_t>(tiles, c->opt_max_grid > 0 ? c->opt_max_grid : fat_grid_cap(c, s.block));
    }
}

bool global_bins(const cfmm_ctx* c) { return c->n > kMaxLdsTokens; }

// Relative cost of one pool evaluation per family, in tenths of a ProductTwoCoin evaluation (options
// "cost_geomean" / "cost_univ3"; measured on config3 / mixed markets, see DESIGN).  Used only to divide
// the blocks of a fused launch among its segments so that they finish together.
int64_t family_cost(const cfmm_ctx* c, const Segment& s)
{
    switch (s.kind) {
    case CFMM_KIND_PRODUCT: return 10;
    case CFMM_KIND_GEOMEAN: return c->opt_cost_geomean;
    default: return c->opt_cost_univ3 + (s.m > 0 && s.n_ticks_total / s.m > 2 ? 2 * (s.n_ticks_total / s.m) : 0);   // deeper ladders walk longer
    }
}

// XCD-aware, cost-weighted map of a fused launch (grid a multiple of 256 blocks): 32-deal pattern in
// which segment s appears seg_w[s] times, spread evenly (largest-remainder weights, Bresenham order).
void plan_xcd_map(cfmm_ctx* c, Group& g)
{
    g.xcd_map = false;
    if (!g.multi || c->opt_xcd_map == 0 || g.grid % 256 != 0 || global_bins(c)) return;
    double cost[kMaxMulti], total = 0.0;
    for (int k = 0; k < g.nseg; ++k) {
        const Segment& s = c->segs[(size_t)g.first + k];
        cost[k] = (double)s.m * (double)(c->opt_xcd_map == 2 ? 10 : family_cost(c, s));   // 2: equal cost per pool
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
int row_width(const cfmm_ctx* c) { return global_bins(c) ? 1 : c->n + 1; }

int bin_copies(const cfmm_ctx* c, int block)
{
    if (global_bins(c)) return 1;
    const int waves = block / 64;
    if (c->opt_bin_copies == 1) return 1;
    const size_t per_wave = sweep_lds_bytes(c->n_pad, waves, block, 1, kMaxFeeTable);   // incl. the log-price row and the fee table a launch may stage
    if (c->opt_bin_copies == 2) return per_wave <= 160 * 1024 ? waves : 1;
    // auto: one private copy per wavefront while two blocks still fit a CU's 160 KiB of LDS
    return per_wave <= (block == kBigBlock ? 80 : (block == kMidBlock ? 48 : 32)) * 1024 ? waves : 1;
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
        fusable = fusable && s.unroll == 1;
        // the fused kernel always carries the wavefront-cooperative UniV3 walk; "univ3_coop" = 0 (lane-per-pool
        // walks only) is honoured by sweeping such routers with per-segment launches
        if (s.kind == CFMM_KIND_UNIV3 && c->opt_univ3_coop == 0) fusable = false;
        any_big = any_big || s.block != kSmallBlock;
    }
    c->groups.clear();
    if (fusable) {
        // fused launches use 512-thread blocks: at ~75 VGPRs three of them fit a CU (24 wavefronts vs 16
        // for one 1024-thread block) and Product / GeoMean blocks interleave on every CU
        const int block = !any_big ? kSmallBlock : (c->opt_block == kBigBlock ? kBigBlock : kMidBlock);
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
                1, (c->opt_max_grid > 0 ? c->opt_max_grid : (block == kSmallBlock ? 2048 : fused_grid_cap(c, block))) / g.nseg);
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
                if (!sg.pk) ok = false;
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
    if (rows > c->rows_cap) {
        (void)hipFree(c->d_partials);
        c->d_partials = nullptr;
        c->rows_cap = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_partials), 2 * (size_t)rows * row_width(c) * sizeof(double)));   // x2: self-validating rows (inline_fold = 3) hold 16 bytes per entry
        HIP_TRY(c, hipMemset(c->d_partials, 0, 2 * (size_t)rows * row_width(c) * sizeof(double)));
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
    c->have_out = false;
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

// Enqueue one full evaluation on c->stream: every segment's sweep, then the row fold.
int enqueue_sweep(cfmm_ctx* c, const double* d_v, double* d_out, bool materialize, bool want_host_flag = false,
                  uint64_t arm_seq = 0)
{
    int rc = ensure_geometry(c);
    if (rc != CFMM_OK) return rc;
    const bool timed = c->opt_time_kernels != 0 && c->pending.size() < (1u << 20); // harvest with cfmm_kernel_times
    const bool gb = global_bins(c);
    // One launch per evaluation: the partial rows are folded by extra blocks of that launch.
    const bool sharded = !c->peers.empty();   // fold + all-reduce over the peer mappings in one launch
    // One launch per evaluation (option "inline_fold"): the partial rows are folded by extra blocks of that launch.
    const bool inline_fold = c->opt_inline_fold != 0 && !gb && !sharded && c->groups.size() == 1 && c->rows_total > 0 &&
                             c->n <= kMaxFoldTokens && c->d_sync != nullptr && arm_seq == 0;
    const unsigned long long* arm_word = arm_seq ? reinterpret_cast<const unsigned long long*>(c->d_arm + c->n_pad) : nullptr;
    // host-visible completion flag: raised by the last fold block (of the sweep launch, or of reduce_partials)
    const bool flagged = want_host_flag && !gb && (sharded || c->rows_total > 0) && c->d_sync != nullptr;
    const bool granules = flagged && !sharded && c->opt_host_granules != 0;
    // where the fold signals completion, and with what: the flag word + sequence number, or the granule array + tag
    unsigned long long* const flag_ptr = !flagged ? nullptr
                                         : reinterpret_cast<unsigned long long*>(c->d_stage + (granules ? c->gran_off : (size_t)(2 * c->n + 1)));
    auto next_seq = [&]() -> unsigned long long {
        ++c->flag_seq;
        return granules ? (kHostGranules | (c->flag_seq % 0xffffffffull + 1ull)) : c->flag_seq;
    };
    HIP_TRY(c, hipSetDevice(c->device));
    size_t group_index = 0;
    for (const Group& g : c->groups) {
        const size_t gi = group_index++;
        SweepArgs a;
        a.v = d_v;
        a.n = c->n;
        a.n_pad = c->n_pad;
        a.gtab = c->d_gtab ? c->d_gtab + gi * kMaxFeeTable : nullptr;
        a.gtab_n = a.gtab ? g.gtab_n : 0;
        a.need_logv = 0;
        if (!gb && c->opt_geomean_exact == 0)
            for (int k = 0; k < g.nseg; ++k)
                if (c->segs[(size_t)g.first + k].kind == CFMM_KIND_GEOMEAN) a.need_logv = 1;
        a.copies = bin_copies(c, g.block);
        a.m = 0;
        a.Delta = a.Lambda = a.Over = nullptr;
        a.compact = (c->opt_compact_trades != 0 && !gb) ? 1 : 0;
        a.partials = c->d_partials + (size_t)g.row_off * row_width(c);
        a.gflow = nullptr;
        a.nt_stores = (int)c->opt_nt_stores;
        a.reverse = c->opt_alternate != 0 ? (int)(c->sweep_count & 1) : 0;
        a.fold_blocks = inline_fold ? (c->n + 1 + kReduceCols - 1) / kReduceCols : 0;
        a.sync = c->d_sync;
        a.fold_out = d_out;
        a.host_flag = flagged && inline_fold ? flag_ptr : nullptr;
        a.host_seq = flagged && inline_fold ? next_seq() : 0;
        a.arm_word = arm_word;
        a.arm_seq = arm_seq;
        a.arm_timeout = std::min<long long>(std::max<long long>(c->opt_arm_timeout_ms, 1), 10000) * 100000ll;   // ms -> ticks of the 100 MHz wall clock, at most 10 s
        const size_t lds = gb ? (size_t)(g.block / 64) * sizeof(double) : sweep_lds_bytes(c->n_pad, a.copies, g.block, a.need_logv, a.gtab_n);
        hipEvent_t ea = nullptr, eb = nullptr;
        if (timed) { // start/stop written by the command processor around this launch (hipExtLaunchKernel)
            ea = take_event(c);
            eb = take_event(c);
            if (!ea || !eb) ea = eb = nullptr;
        }
        hipError_t e = hipSuccess;
        if (g.multi) {
            MultiArgs ma;
            std::memset(&ma, 0, sizeof ma);
            ma.nseg = g.nseg;
            ma.xcd_map = g.xcd_map ? 1 : 0;
            std::memcpy(ma.pattern, g.pattern, sizeof ma.pattern);
            std::memcpy(ma.rank, g.rank, sizeof ma.rank);
            for (int k = 0; k < kMaxMulti; ++k) ma.seg_w[k] = g.seg_w[k];
            ma.wave_split = (c->opt_wave_split != 0 && (g.block / 64) % g.nseg == 0 && !gb) ? 1 : 0;
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
                case CFMM_KIND_PRODUCT: ms.pools.p = ProductPools{s.R, s.gamma, s.Ai, a.gtab_n ? s.pk : nullptr, s.gbase}; break;
                case CFMM_KIND_GEOMEAN: ms.pools.g = GeoMeanPools{s.R, s.w, s.gamma, s.Ai, s.eta, s.lR, (int)c->opt_geomean_exact, a.gtab_n ? s.pk : nullptr, s.gbase}; break;
                default: ms.pools.u = UniV3Pools{s.pg, s.Ai, s.cur_a, s.cur_b, s.cur_c, s.curR, s.walk, s.ks, s.dt, s.rout, c->opt_univ3_coop < 0 ? s.deep : (int)(c->opt_univ3_coop != 0), s.has_walk, s.cp, a.gtab_n ? s.pk : nullptr, s.gbase}; break;
                }
            }
            LaunchCfg cfg{g.block, g.grid, 1, lds, ea, eb};
            e = launch_multi(ma, cfg, materialize, c->stream);
        } else {
            const Segment& s = c->segs[(size_t)g.first];
            a.m = s.m;
            a.Delta = materialize ? c->d_delta + s.trade_off : nullptr;
            a.Lambda = materialize ? c->d_lambda + s.trade_off : nullptr;
            a.Over = materialize ? c->d_over + s.trade_off : nullptr;
            a.gflow = gb ? c->d_flow + s.trade_off : nullptr;
            LaunchCfg cfg{g.block, g.grid, s.unroll, lds, ea, eb};
            switch (s.kind) {
            case CFMM_KIND_PRODUCT: e = launch_sweep(ProductPools{s.R, s.gamma, s.Ai, a.gtab_n ? s.pk : nullptr, s.gbase}, a, cfg, materialize, c->stream); break;
            case CFMM_KIND_GEOMEAN:
                e = launch_sweep(GeoMeanPools{s.R, s.w, s.gamma, s.Ai, s.eta, s.lR, (int)c->opt_geomean_exact,
                                              a.gtab_n && c->opt_geomean_exact == 0 ? s.pk : nullptr, s.gbase}, a, cfg, materialize, c->stream);
                break;
            default:
                e = launch_sweep(UniV3Pools{s.pg, s.Ai, s.cur_a, s.cur_b, s.cur_c, s.curR, s.walk, s.ks, s.dt, s.rout, c->opt_univ3_coop < 0 ? s.deep : (int)(c->opt_univ3_coop != 0), s.has_walk, s.cp, a.gtab_n ? s.pk : nullptr, s.gbase}, a, cfg, materialize, c->stream);
                break;
            }
        }
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "sweep launch failed: %s", hipGetErrorString(e));
        if (ea && eb) c->pending.push_back({ea, eb, 0});
    }
    c->last_inline = inline_fold;
    c->last_flagged = flagged;
    c->last_granules = granules;
    hipEvent_t ra = nullptr, rb = nullptr;
    if (timed) {
        ra = take_event(c);
        rb = take_event(c);
        if (!ra || !rb) ra = rb = nullptr;
        if (ra && !inline_fold && (gb || (c->rows_total == 0 && !sharded))) HIP_TRY(c, hipEventRecord(ra, c->stream)); // several launches: bracket them
    }
    if (inline_fold) {
        // nothing to launch: the sweep launch has already produced d_out
    } else if (sharded) {
        PeerSet ps;
        const int64_t count = c->n + 1;
        for (size_t p = 0; p < c->peers.size(); ++p)
            ps.gran[p] = reinterpret_cast<unsigned long long*>(c->peers[p] + (uint64_t)(2 * count + 2) * sizeof(double));
        ps.world = (int)c->peers.size();
        ps.rank = c->peer_rank;
        ps.count = count;
        ps.seq = ++c->peer_seq;
        ps.timeout_ticks = c->peer_timeout_ticks;
        ps.sync = c->d_sync;
        ps.host_flag = flagged ? reinterpret_cast<unsigned long long*>(c->d_stage + 2 * c->n + 1) : nullptr;
        ps.host_seq = flagged ? ++c->flag_seq : 0;
        hipError_t e = launch_reduce_gather(c->d_partials, (int)c->rows_total, c->n + 1, d_out, c->stream,
                                            c->groups.empty() ? kSmallBlock : c->groups.back().block, ps, ra, rb);
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "fold + gather launch failed: %s", hipGetErrorString(e));
    } else if (c->rows_total > 0) {
        hipError_t e;
        if (gb) { // pull Ψ per token over the incidence list, then fold the dual-scalar column
            e = launch_gather(c->d_chunks, c->d_entries, reinterpret_cast<const double*>(c->d_flow), c->d_chunk_sums,
                              c->n_chunks, c->d_tok_chunk_off, d_out, c->n, c->d_partials, (int)c->rows_total, c->stream);
        } else {
            e = launch_reduce(c->d_partials, (int)c->rows_total, c->n + 1, d_out, c->stream, c->groups.back().block, ra, rb,
                              c->d_sync, flag_ptr, flagged ? next_seq() : 0, ArmWord{arm_word, arm_seq});
        }
        if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "reduce launch failed: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(c, hipMemsetAsync(d_out, 0, (size_t)(c->n + 1) * sizeof(double), c->stream));
    }
    if (ra && rb && !inline_fold) {
        if (gb || (c->rows_total == 0 && !sharded)) HIP_TRY(c, hipEventRecord(rb, c->stream));
        c->pending.push_back({ra, rb, 1});
    }
    if (materialize) {
        c->have_trades = true;
        c->trades_compact = (c->opt_compact_trades != 0 && !gb) ? 1 : 0;
    }
    ++c->sweep_count;
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

// First half of a host-pointer sweep: stage v, enqueue the evaluation (asynchronous).
int host_sweep_begin(cfmm_ctx* c, const double* v, bool materialize)
{
    HIP_TRY(c, hipSetDevice(c->device));
    std::memcpy(c->h_stage, v, (size_t)c->n * sizeof(double));
    if (materialize) c->trade_v.assign(v, v + c->n);   // the prices the device trades belong to (update_reserves!)
    double* h_out = c->h_stage + c->n;
    const bool zero_copy = c->opt_zero_copy != 0 && c->d_stage != nullptr;
    // v: small vectors are read by every block straight from the mapped pinned buffer (the PCIe
    // round trip hides behind the first tile's pool loads); larger ones go through one H2D copy.
    const double* v_src = c->d_stage;
    if (!zero_copy || c->n > 1024 || global_bins(c)) {
        HIP_TRY(c, hipMemcpyAsync(c->d_v, c->h_stage, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        v_src = c->d_v;
    }
    // {Ψ, acc}: the last kernel of the evaluation writes into the mapped pinned buffer when it can
    double* out_dst = zero_copy ? c->d_stage + c->n : c->d_out;
    {   // sharded contexts (cfmm_set_peers): the fold launch also gathers the peers' {Ψ, acc} over xGMI
        int rc = enqueue_sweep(c, v_src, out_dst, materialize, zero_copy && c->opt_host_flag != 0);
        if (rc != CFMM_OK) return rc;
    }
    if (!zero_copy)
        HIP_TRY(c, hipMemcpyAsync(h_out, c->d_out, (size_t)(c->n + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return CFMM_OK;
}

// Output granules of the flagged sweep with sequence number `seq` (fold_finish, kHostGranules): true once all 2(n+1)
// carry its tag; the doubles are then reassembled into the {Ψ, acc} slots of the staging buffer.
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
            // a fold that gave up may have left its arrival / ticket words non-zero: clean them for the next sweep
            (void)hipStreamSynchronize(c->stream);
            if (c->d_sync) (void)hipMemset(c->d_sync, 0, (size_t)kSyncWords * sizeof(unsigned));
            if (!c->peers.empty())
                return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}[%d]: the peer all-reduce timed out (a rank did not "
                                               "publish) or a shard overflowed", j);
            return fail(c, CFMM_ERR_STATE, "non-finite {psi, acc}[%d]: pool arithmetic overflowed, or the in-launch fold "
                                           "timed out", j);
        }
    c->have_out = true;
    return CFMM_OK;
}

// Second half: wait for {Ψ, acc} to be on the host and take them over into last_out.
int host_sweep_end(cfmm_ctx* c)
{
    bool flag_seen = false;
    if (c->last_flagged) {
        // The last fold block wrote {Ψ, acc} through to this pinned buffer and then raised the flag
        // (PCIe posted writes stay ordered): poll it instead of waiting for the kernel's end-of-pipe
        // processing and its completion signal.  Bounded; falls back to a stream wait.
        volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_stage + 2 * c->n + 1);
        const unsigned long long want = c->flag_seq;
        for (long spins = 0; spins < 400000000L; ++spins) {
            if (c->last_granules ? granules_arrived(c, want) : *flag == want) { flag_seen = true; break; }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (flag_seen) {
        // results are on the host; the kernel itself retires in stream order behind us
    } else if (c->opt_spin_wait != 0) {
        // busy-poll the stream instead of a blocking wait
        hipError_t q;
        while ((q = hipStreamQuery(c->stream)) == hipErrorNotReady) {}
        if (q != hipSuccess) return fail(c, CFMM_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
    } else {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (!flag_seen && c->last_flagged && c->last_granules && !granules_arrived(c, c->flag_seq)) {
        c->have_out = false;
        return fail(c, CFMM_ERR_STATE, "the sweep retired without delivering its outputs");
    }
    return take_host_out(c);
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
bool can_arm(cfmm_ctx* c)
{
    if (c->opt_armed == 0 || !c->d_arm || !c->shards.empty() || !c->peers.empty() || c->opt_zero_copy == 0 || !c->d_stage ||
        c->opt_host_flag == 0 || c->opt_time_kernels != 0 || c->opt_wave_split != 0 || c->n > 1024 || !c->d_sync ||
        c->stream != c->own_stream || global_bins(c))
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
    int rc = enqueue_sweep(c, c->d_arm, c->d_stage + c->n, false, true, seq);
    if (rc != CFMM_OK) return rc;
    c->arm_flag = c->flag_seq;
    c->arm_pending = true;
    return CFMM_OK;
}

void armed_cancel(cfmm_ctx* c)
{
    if (!c->arm_pending) return;
    armed_write(c, nullptr, c->arm_seq | kArmCancel);
    c->arm_pending = false;
    --c->sweep_count;   // the cancelled launch swept nothing: later sweeps keep the tile directions of an unarmed run
}

// One fused evaluation at v through the armed launch (enqueuing it first if none is waiting), with the next one
// enqueued behind it while it runs.
int armed_eval(cfmm_ctx* c, const double* v)
{
    int rc = check_prices(c, v);
    if (rc != CFMM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->arm_pending) {
        rc = armed_enqueue(c);
        if (rc != CFMM_OK) return rc;
    }
    const uint64_t want = c->arm_flag;
    armed_write(c, v, c->arm_seq);
    c->arm_pending = false;
    const int rc_next = armed_enqueue(c);   // evaluation k+1 goes out while k runs
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_stage + 2 * c->n + 1);
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    const bool by_granules = c->last_granules;
    for (long spins = 0;; ++spins) {
        if (by_granules ? granules_arrived(c, want) : *flag == want) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >
                2e-3 * (double)std::min<int64_t>(std::max<int64_t>(c->opt_arm_timeout_ms, 1), 10000) + 1.0)
            break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) {
        armed_cancel(c);
        (void)hipStreamSynchronize(c->stream);
        if (c->d_sync) (void)hipMemset(c->d_sync, 0, (size_t)kSyncWords * sizeof(unsigned));
        c->have_out = false;
        return fail(c, CFMM_ERR_STATE, "armed evaluation did not complete (the device never saw its price vector)");
    }
    rc = take_host_out(c);
    if (rc != CFMM_OK) { armed_cancel(c); return rc; }
    if (rc_next != CFMM_OK) return rc_next;
    return CFMM_OK;
}

int single_host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    int rc = host_sweep_begin(c, v, materialize);
    return rc != CFMM_OK ? rc : host_sweep_end(c);
}

// ---- single-process multi-device context (cfmm_ctx_create_multi) ------------------------------
// The parent context owns one ordinary single-device context per shard.  Pools of every batch are
// split into contiguous blocks over the shards (no pool is replicated); a host-pointer sweep stages
// the same v on every device, runs the shard sweeps concurrently (one host worker thread per shard,
// or -- option "multi_threads" = 0 -- all launches from the calling thread, then all waits) and sums
// the shards' {Ψ, acc} ON THE HOST in shard order: v comes from the host and Ψ returns to it on
// every evaluation anyway, so the "all-reduce" of SURVEY 8e degenerates to N·(n+1) additions --
// no peer access, no IPC, no torch.  One L-BFGS-B (cfmm_route) drives all shards.

void shard_range(int64_t m, int d, int nd, int64_t& lo, int64_t& hi)
{
    const int64_t base = m / nd, rem = m % nd;
    lo = d * base + std::min<int64_t>(d, rem);
    hi = lo + base + (d < rem ? 1 : 0);
}

void worker_main(cfmm_ctx* parent, int d)
{
    Workers& w = *parent->workers;
    cfmm_ctx* child = parent->shards[(size_t)d];
    (void)hipSetDevice(child->device);
    uint64_t seen = 0;
    for (;;) {
        int spins = 0;
        while (w.go.load(std::memory_order_acquire) == seen && !w.quit.load(std::memory_order_relaxed)) {
            if (++spins < 20000) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(w.mu);
            w.sleepers.fetch_add(1);
            w.cv.wait(lk, [&] { return w.go.load(std::memory_order_acquire) != seen || w.quit.load(); });
            w.sleepers.fetch_sub(1);
        }
        if (w.quit.load()) return;
        seen = w.go.load(std::memory_order_acquire);
        w.rc[(size_t)d] = single_host_sweep(child, w.v, w.materialize);
        w.pending.fetch_sub(1, std::memory_order_release);
    }
}

int multi_host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    const int nd = (int)c->shards.size();
    std::vector<int>& rcs = c->workers->rc;
    std::fill(rcs.begin(), rcs.end(), CFMM_OK);
    if (c->opt_multi_threads != 0 && nd > 1) {
        Workers& w = *c->workers;
        if (w.threads.empty())
            for (int d = 1; d < nd; ++d) w.threads.emplace_back(worker_main, c, d);
        w.v = v;
        w.materialize = materialize;
        w.pending.store(nd - 1, std::memory_order_relaxed);
        w.go.fetch_add(1, std::memory_order_release);
        if (w.sleepers.load() > 0) {
            std::lock_guard<std::mutex> lk(w.mu);
            w.cv.notify_all();
        }
        rcs[0] = single_host_sweep(c->shards[0], v, materialize);
        while (w.pending.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
    } else {
        for (int d = 0; d < nd; ++d) rcs[(size_t)d] = host_sweep_begin(c->shards[(size_t)d], v, materialize);
        for (int d = 0; d < nd; ++d)
            if (rcs[(size_t)d] == CFMM_OK) rcs[(size_t)d] = host_sweep_end(c->shards[(size_t)d]);
    }
    for (int d = 0; d < nd; ++d)
        if (rcs[(size_t)d] != CFMM_OK) {
            c->have_out = false;
            return fail(c, rcs[(size_t)d], "shard %d (device %d): %s", d, c->shards[(size_t)d]->device,
                        c->shards[(size_t)d]->err.c_str());
        }
    // the all-reduce: shard order, on the host
    c->last_out.assign((size_t)c->n + 1, 0.0);
    for (int d = 0; d < nd; ++d) {
        const std::vector<double>& o = c->shards[(size_t)d]->last_out;
        for (int j = 0; j <= c->n; ++j) c->last_out[(size_t)j] += o[(size_t)j];
    }
    c->have_out = true;
    c->have_trades = materialize;
    return CFMM_OK;
}


// ---- multi-device parents: pool upload and trade download across the shards -------------------
void pop_last_segment(cfmm_ctx* child)
{
    (void)hipSetDevice(child->device);
    (void)hipStreamSynchronize(child->stream);
    free_segment(child->segs.back());
    child->segs.pop_back();
    child->geometry_dirty = true;
    child->have_out = child->have_trades = false;
}

// Run add(child, d, lo, hi) on every shard with a non-empty block [lo, hi) of the m pools; all or nothing.
template <class F>
int multi_add(cfmm_ctx* c, int kind, int64_t m, F add)
{
    if (m < 0) return fail(c, CFMM_ERR_INVALID_ARG, "negative pool count");
    if (m == 0) return CFMM_OK;
    const int nd = (int)c->shards.size();
    std::vector<int> added;
    for (int d = 0; d < nd; ++d) {
        int64_t lo, hi;
        shard_range(m, d, nd, lo, hi);
        if (hi == lo) continue;
        cfmm_ctx* child = c->shards[(size_t)d];
        const int rc = add(child, lo, hi);
        if (rc != CFMM_OK) {
            fail(c, rc, "pools [%lld, %lld) -> shard %d (device %d): %s", (long long)lo, (long long)hi, d, child->device,
                 child->err.c_str());
            for (int a : added) pop_last_segment(c->shards[(size_t)a]);
            return rc;
        }
        added.push_back(d);
    }
    c->psegs.push_back({kind, m, c->m_total});
    c->m_total += m;
    c->have_out = c->have_trades = false;
    return CFMM_OK;
}

// child segment index that holds shard d's block of parent segment `pseg` (-1: that block is empty)
int child_segment(const cfmm_ctx* c, int pseg, int d)
{
    const int nd = (int)c->shards.size();
    int idx = 0;
    for (int k = 0; k <= pseg; ++k) {
        int64_t lo, hi;
        shard_range(c->psegs[(size_t)k].m, d, nd, lo, hi);
        if (k == pseg) return hi > lo ? idx : -1;
        if (hi > lo) ++idx;
    }
    return -1;
}

int multi_get_trades_range(cfmm_ctx* c, int32_t seg, int64_t first, int64_t count, double* Delta, double* Lambda)
{
    if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb first");
    if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    const int64_t m = c->psegs[(size_t)seg].m;
    if (first < 0 || count < 0 || first + count > m) return fail(c, CFMM_ERR_INVALID_ARG, "row range out of bounds");
    const int nd = (int)c->shards.size();
    for (int d = 0; d < nd; ++d) {
        int64_t lo, hi;
        shard_range(m, d, nd, lo, hi);
        const int64_t a = std::max(lo, first), b = std::min(hi, first + count);
        if (b <= a) continue;
        cfmm_ctx* child = c->shards[(size_t)d];
        const int rc = cfmm_get_trades_range(child, child_segment(c, seg, d), a - lo, b - a,
                                             Delta ? Delta + 2 * (a - first) : nullptr,
                                             Lambda ? Lambda + 2 * (a - first) : nullptr);
        if (rc != CFMM_OK) return fail(c, rc, "shard %d: %s", d, child->err.c_str());
    }
    return CFMM_OK;
}

int host_sweep(cfmm_ctx* c, const double* v, bool materialize)
{
    int rc = check_prices(c, v);
    if (rc != CFMM_OK) return rc;
    return c->shards.empty() ? single_host_sweep(c, v, materialize) : multi_host_sweep(c, v, materialize);
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

// Validates m UniV3 pools and prepares + uploads the find_arb_pos constants (see UniV3Ops) into `s`.
int univ3_build(cfmm_ctx* c, Segment& s, int64_t m, const double* current_price, const double* gamma, const int32_t* Ai,
                const int64_t* tick_off, const double* lower_ticks, const double* liquidity)
{
    const int64_t T = m > 0 ? tick_off[m] : 0;
    if (T < 0 || 2 * T > (int64_t)0x3fffffff) return fail(c, CFMM_ERR_UNSUPPORTED, "too many ticks in one segment");
    std::vector<double2> pg((size_t)m), ks, dt, cur_a((size_t)m), cur_b((size_t)m), curR((size_t)m);
    std::vector<double> rout, cur_c((size_t)m);
    std::vector<int4> walk((size_t)m);
    int longest = 0;
    ks.reserve((size_t)T + (size_t)m);
    dt.reserve((size_t)T + (size_t)m);
    rout.reserve((size_t)T + (size_t)m);
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
        int4 w;
        w.x = (int)ks.size();
        int cnt = 0;
        for (int64_t idx = ct + 1; idx <= nt; ++idx) {        // get_upper_pools beyond the current tick, :316
            double k, al, be, R1, R2;
            at_tick(idx, k, al, be, R1, R2);
            if (k == 0) continue;                             // is_empty_pool, :288
            const double s_in = R1 + al;
            ks.push_back(make_double2(k, s_in));
            dt.push_back(make_double2(k / be - s_in, R2 + be)); // :329, :334
            rout.push_back(R2);
            ++cnt;
        }
        w.y = cnt;
        w.z = (int)ks.size();
        cnt = 0;
        for (int64_t idx = ct - 1; idx >= 1; --idx) {         // flip_sides.(get_lower_pools), :317,:289
            double k, al, be, R1, R2;
            at_tick(idx, k, al, be, R1, R2);
            if (k == 0) continue;
            const double s_in = R2 + be;
            ks.push_back(make_double2(k, s_in));
            dt.push_back(make_double2(k / al - s_in, R1 + al));
            rout.push_back(R1);
            ++cnt;
        }
        w.w = cnt;
        longest = std::max(longest, std::max(w.y, w.w));
        walk[(size_t)i] = w;
        pg[(size_t)i] = make_double2(cp, gamma[i]);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    s.kind = CFMM_KIND_UNIV3;
    s.m = m;
    s.n_ticks_total = T;
    s.deep = longest > 8 ? 1 : 0; // short ladders: a lane walks its own pool; long: the wavefront helps
    s.has_walk = longest > 0 ? 1 : 0;
    int rc;
    if ((rc = upload(c, &s.pg, pg.data(), (size_t)m)) || (rc = upload(c, &s.Ai, Ai, (size_t)m)) ||
        (rc = upload(c, &s.cur_a, cur_a.data(), (size_t)m)) || (rc = upload(c, &s.cur_b, cur_b.data(), (size_t)m)) ||
        (rc = upload(c, &s.cur_c, cur_c.data(), (size_t)m)) || (rc = upload(c, &s.curR, curR.data(), (size_t)m)) ||
        (rc = upload(c, &s.walk, walk.data(), (size_t)m)) || (rc = upload(c, &s.ks, ks.data(), ks.size())) ||
        (rc = upload(c, &s.dt, dt.data(), dt.size())) || (rc = upload(c, &s.rout, rout.data(), rout.size())) ||
        (rc = upload(c, &s.cp, current_price, (size_t)m)) || (rc = build_packed(c, s, m, gamma, Ai))) {
        free_segment(s);
        return rc;
    }
    return CFMM_OK;

}

} // namespace

extern "C" {

const char* cfmm_version(void) { return "cfmm_amd 0.1.0 (gfx950)"; }

const char* cfmm_last_error(const cfmm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int cfmm_ctx_create(int device_id, int32_t n_tokens, cfmm_ctx** out)
{
    if (!out) return fail(nullptr, CFMM_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_tokens < 1) return fail(nullptr, CFMM_ERR_INVALID_ARG, "n_tokens must be >= 1");
    if (n_tokens > (1 << 26))
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "n_tokens %d exceeds the supported maximum %d", n_tokens, 1 << 26);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, CFMM_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, CFMM_ERR_INVALID_ARG, "device_id %d out of range [0, %d)", device_id, count);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device_id,
                    prop.gcnArchName);
    HIP_TRY(nullptr, hipSetDevice(device_id));

    cfmm_ctx* c = new cfmm_ctx();
    c->device = device_id;
    c->n = n_tokens;
    c->n_pad = (n_tokens + 1) & ~1;
    auto bail = [&](int code) {
        g_create_error = c->err;
        cfmm_ctx_destroy(c);
        return code;
    };
#define HIP_TRY_C(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            fail(c, CFMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));                     \
            return bail(CFMM_ERR_HIP);                                                                \
        }                                                                                             \
    } while (0)
    HIP_TRY_C(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&c->d_v), (size_t)c->n * sizeof(double)));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&c->d_out), (size_t)(c->n + 1) * sizeof(double)));
    // [n] v, [n+1] {psi, acc}, completion flag, padding to a 128-byte boundary, then the output granules: 16 per fold
    // block = 2 per column, columns padded to a multiple of 8 (see fold_finish / kHostGranules)
    c->gran_off = (size_t)((2 * c->n + 2 + 15) & ~15);
    const size_t stage_words = c->gran_off + 2 * (size_t)((c->n + 1 + 7) & ~7);
    HIP_TRY_C(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), stage_words * sizeof(double), hipHostMallocMapped));
    std::memset(c->h_stage, 0, stage_words * sizeof(double));
    HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&c->d_sync), (size_t)kSyncWords * sizeof(unsigned)));
    HIP_TRY_C(hipMemset(c->d_sync, 0, (size_t)kSyncWords * sizeof(unsigned)));
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_stage), c->h_stage, 0) != hipSuccess) {
        (void)hipGetLastError();
        c->d_stage = nullptr; // fall back to explicit copies
    }
    {   // armed evaluations: fine-grained device memory the host can write through the PCIe BAR (optional)
        int large_bar = 0;
        const char* env = std::getenv("CFMM_AMD_ARMED");
        if (!(env && env[0] == '0') &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar != 0) {
            const size_t words = (size_t)c->n_pad + 8;
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&c->d_arm), words * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess) {
                // self-check: what the host stores must be what the device holds
                std::vector<double> probe(words), back(words, 0.0);
                for (size_t j = 0; j < words; ++j) probe[j] = 1.0 + (double)j;
                std::memcpy(c->d_arm, probe.data(), words * sizeof(double));
                __builtin_ia32_sfence();
                if (hipMemcpy(back.data(), c->d_arm, words * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess || back != probe) {
                    (void)hipGetLastError();
                    (void)hipFree(c->d_arm);
                    c->d_arm = nullptr;
                } else {
                    std::memset(c->d_arm, 0, words * sizeof(double));
                    __builtin_ia32_sfence();
                }
            } else {
                (void)hipGetLastError();
                c->d_arm = nullptr;
            }
        }
    }
    // the dynamic-LDS ceiling is a per-function, process-wide attribute: always raise it to the
    // full 160 KiB so that contexts with different n_tokens cannot shrink each other's limit
    HIP_TRY_C(prepare_kernels(160 * 1024));
#undef HIP_TRY_C
    *out = c;
    return CFMM_OK;
}

int cfmm_ctx_create_multi(int32_t n_devices, const int32_t* device_ids, int32_t n_tokens, cfmm_ctx** out)
{
    if (!out) return fail(nullptr, CFMM_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !device_ids)
        return fail(nullptr, CFMM_ERR_INVALID_ARG, "n_devices must be in [1, 64] and device_ids non-null");
    if (n_tokens > kMaxLdsTokens)
        return fail(nullptr, CFMM_ERR_UNSUPPORTED, "multi-device contexts are limited to n_tokens <= %d", kMaxLdsTokens);
    cfmm_ctx* c = new cfmm_ctx();
    c->device = -1;
    c->n = n_tokens;
    c->n_pad = (n_tokens + 1) & ~1;
    c->workers.reset(new Workers());
    c->workers->rc.assign((size_t)n_devices, CFMM_OK);
    for (int d = 0; d < n_devices; ++d) {
        cfmm_ctx* child = nullptr;
        int rc = cfmm_ctx_create(device_ids[d], n_tokens, &child);   // validates the ordinal, the arch, n_tokens
        if (rc != CFMM_OK) {
            cfmm_ctx_destroy(c);
            return rc;   // g_create_error already holds the message
        }
        c->shards.push_back(child);
    }
    *out = c;
    return CFMM_OK;
}

int32_t cfmm_device_count(const cfmm_ctx* c) { return c ? (c->shards.empty() ? 1 : (int32_t)c->shards.size()) : 0; }

void cfmm_ctx_destroy(cfmm_ctx* c)
{
    if (!c) return;
    if (c->workers) {
        Workers& w = *c->workers;
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.quit.store(true);
            w.cv.notify_all();
        }
        for (auto& t : w.threads) t.join();
    }
    if (!c->shards.empty() || c->device < 0) {
        for (cfmm_ctx* child : c->shards) cfmm_ctx_destroy(child);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream && c->stream != c->own_stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto& s : c->segs) free_segment(s);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipFree(c->d_v); (void)hipFree(c->d_out); (void)hipFree(c->d_partials); (void)hipFree(c->d_sync); (void)hipFree(c->d_gtab);
    (void)hipFree(c->d_delta); (void)hipFree(c->d_lambda); (void)hipFree(c->d_over);
    (void)hipFree(c->d_xdelta); (void)hipFree(c->d_xlambda);
    (void)hipFree(c->d_flow); (void)hipFree(c->d_entries); (void)hipFree(c->d_chunks);
    (void)hipFree(c->d_tok_chunk_off); (void)hipFree(c->d_chunk_sums);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_arm) (void)hipFree(c->d_arm);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

#define CFMM_SINGLE_ONLY(c, what)                                                                     \
    if (!(c)->shards.empty() || (c)->device < 0)                                                      \
        return fail(c, CFMM_ERR_UNSUPPORTED, what " is not available on a multi-device context (host-pointer calls only)")

int cfmm_set_stream(cfmm_ctx* c, void* hip_stream)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_set_stream");
    c->stream = static_cast<hipStream_t>(hip_stream); // NULL is HIP's default (null) stream
    return CFMM_OK;
}

int cfmm_reset_stream(cfmm_ctx* c)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_reset_stream");
    c->stream = c->own_stream;
    return CFMM_OK;
}

static int64_t* option_slot(cfmm_ctx* c, const char* key)
{
    if (!key) return nullptr;
    if (!std::strcmp(key, "max_grid")) return &c->opt_max_grid;
    if (!std::strcmp(key, "unroll")) return &c->opt_unroll;
    if (!std::strcmp(key, "block")) return &c->opt_block;
    if (!std::strcmp(key, "bin_copies")) return &c->opt_bin_copies;
    if (!std::strcmp(key, "time_kernels")) return &c->opt_time_kernels;
    if (!std::strcmp(key, "nt_stores")) return &c->opt_nt_stores;
    if (!std::strcmp(key, "geomean_exact")) return &c->opt_geomean_exact;
    if (!std::strcmp(key, "fuse_segments")) return &c->opt_fuse_segments;
    if (!std::strcmp(key, "zero_copy")) return &c->opt_zero_copy;
    if (!std::strcmp(key, "univ3_coop")) return &c->opt_univ3_coop;
    if (!std::strcmp(key, "spin_wait")) return &c->opt_spin_wait;
    if (!std::strcmp(key, "inline_fold")) return &c->opt_inline_fold;
    if (!std::strcmp(key, "alternate")) return &c->opt_alternate;
    if (!std::strcmp(key, "pack")) return &c->opt_pack;
    if (!std::strcmp(key, "compact_trades")) return &c->opt_compact_trades;
    if (!std::strcmp(key, "armed")) return &c->opt_armed;
    if (!std::strcmp(key, "host_granules")) return &c->opt_host_granules;
    if (!std::strcmp(key, "arm_timeout_ms")) return &c->opt_arm_timeout_ms;
    if (!std::strcmp(key, "xcd_map")) return &c->opt_xcd_map;
    if (!std::strcmp(key, "cost_geomean")) return &c->opt_cost_geomean;
    if (!std::strcmp(key, "cost_univ3")) return &c->opt_cost_univ3;
    if (!std::strcmp(key, "wave_split")) return &c->opt_wave_split;
    if (!std::strcmp(key, "host_flag")) return &c->opt_host_flag;
    if (!std::strcmp(key, "multi_threads")) return &c->opt_multi_threads;
    return nullptr;
}

int cfmm_set_option(cfmm_ctx* c, const char* key, int64_t value)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int64_t* slot = option_slot(c, key);
    if (!slot) return fail(c, CFMM_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    if (slot == &c->opt_max_grid && value < 0) return fail(c, CFMM_ERR_INVALID_ARG, "max_grid must be >= 0 (0 = auto)");
    if (slot == &c->opt_unroll && !(value == 0 || value == 1 || value == 2 || value == 4))
        return fail(c, CFMM_ERR_INVALID_ARG, "unroll must be 0 (auto), 1, 2 or 4");
    if (slot == &c->opt_block && !(value == 0 || value == kSmallBlock || value == kMidBlock || value == kBigBlock))
        return fail(c, CFMM_ERR_INVALID_ARG, "block must be 0 (auto), %d, %d or %d", kSmallBlock, kMidBlock, kBigBlock);
    if (slot == &c->opt_bin_copies && !(value == 0 || value == 1 || value == 2))
        return fail(c, CFMM_ERR_INVALID_ARG, "bin_copies must be 0 (auto), 1 (shared) or 2 (per wavefront)");
    *slot = value;
    if (slot != &c->opt_multi_threads)
        for (cfmm_ctx* child : c->shards) {
            int rc = cfmm_set_option(child, key, value);
            if (rc != CFMM_OK) return fail(c, rc, "%s", child->err.c_str());
        }
    if (slot == &c->opt_max_grid || slot == &c->opt_unroll || slot == &c->opt_block || slot == &c->opt_fuse_segments ||
        slot == &c->opt_geomean_exact || slot == &c->opt_univ3_coop || slot == &c->opt_pack || slot == &c->opt_xcd_map || slot == &c->opt_cost_geomean || slot == &c->opt_cost_univ3)
        c->geometry_dirty = true;
    return CFMM_OK;
}

int cfmm_get_option(const cfmm_ctx* c, const char* key, int64_t* value)
{
    if (!c || !value) return CFMM_ERR_INVALID_ARG;
    int64_t* slot = option_slot(const_cast<cfmm_ctx*>(c), key);
    if (!slot) return fail(c, CFMM_ERR_INVALID_ARG, "unknown option '%s'", key ? key : "(null)");
    *value = *slot;
    return CFMM_OK;
}

int cfmm_pools_add_product(cfmm_ctx* c, int64_t m, const double* R, const double* gamma, const int32_t* Ai)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    int rc = check_two_coin(c, m, R, gamma, Ai);
    if (rc != CFMM_OK) return rc;
    if (!c->shards.empty())
        return multi_add(c, CFMM_KIND_PRODUCT, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) {
            return cfmm_pools_add_product(child, hi - lo, R + 2 * lo, gamma + lo, Ai + 2 * lo);
        });
    HIP_TRY(c, hipSetDevice(c->device));
    Segment s;
    s.kind = CFMM_KIND_PRODUCT;
    s.m = m;
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
        return multi_add(c, CFMM_KIND_GEOMEAN, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) {
            return cfmm_pools_add_geomean(child, hi - lo, R + 2 * lo, w + 2 * lo, gamma + lo, Ai + 2 * lo);
        });
    // v-independent pieces of the log-space closed forms (sweep_kernels.hip, GeoMeanLogOps)
    std::vector<double2> lR((size_t)m);
    std::vector<double> etas((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        const double e = w[2 * i] / w[2 * i + 1]; // src/cfmms.jl:188
        const double lg = std::log(gamma[i]), le = std::log(e), l1 = std::log(R[2 * i]), l2 = std::log(R[2 * i + 1]);
        etas[(size_t)i] = e;
        lR[(size_t)i] = make_double2(((lg + le) + l2) + e * l1, e * ((lg + l1) - le) + l2);   // {Q1, Q2}
    }
    HIP_TRY(c, hipSetDevice(c->device));
    Segment s;
    s.kind = CFMM_KIND_GEOMEAN;
    s.m = m;
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
        return multi_add(c, CFMM_KIND_UNIV3, m, [&](cfmm_ctx* child, int64_t lo, int64_t hi) {
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

int64_t cfmm_pools_count(const cfmm_ctx* c)
{
    if (!c) return 0;
    int64_t m = 0;
    for (auto& s : c->segs) m += s.m;
    for (auto& ps : c->psegs) m += ps.m;
    return m;
}

int32_t cfmm_n_tokens(const cfmm_ctx* c) { return c ? c->n : 0; }

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

// D2H of trade rows [row0, row0 + count) of the context's buffers into Delta / Lambda ([count][2] each, may be
// null), decoding the compact records on the way.
static int download_trades(cfmm_ctx* c, int64_t row0, int64_t count, double* Delta, double* Lambda)
{
    if (count == 0) return CFMM_OK;
    if (!c->trades_compact) {
        if (Delta) HIP_TRY(c, hipMemcpy(Delta, c->d_delta + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        if (Lambda) HIP_TRY(c, hipMemcpy(Lambda, c->d_lambda + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        return CFMM_OK;
    }
    c->h_rec.resize((size_t)count);
    HIP_TRY(c, hipMemcpy(c->h_rec.data(), c->d_delta + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
    bool any_overflow = false;
    for (int64_t i = 0; i < count; ++i) {
        const double2 r = c->h_rec[(size_t)i];
        double d1 = 0.0, d2 = 0.0, l1 = 0.0, l2 = 0.0;
        if (r.y == -1.0) any_overflow = true;                         // both directions: fetched below
        else if (std::signbit(r.x)) { d2 = -r.x; l1 = r.y; }           // {−Δ₂, Λ₁}
        else { d1 = r.x; l2 = r.y; }                                   // {+Δ₁, Λ₂}
        if (Delta) { Delta[2 * i] = d1; Delta[2 * i + 1] = d2; }
        if (Lambda) { Lambda[2 * i] = l1; Lambda[2 * i + 1] = l2; }
    }
    if (any_overflow) {
        c->h_ovA.resize((size_t)count);
        c->h_ovB.resize((size_t)count);
        HIP_TRY(c, hipMemcpy(c->h_ovA.data(), c->d_lambda + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_ovB.data(), c->d_over + row0, (size_t)count * sizeof(double2), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < count; ++i) {
            if (c->h_rec[(size_t)i].y != -1.0) continue;
            if (Delta) { Delta[2 * i] = c->h_ovA[(size_t)i].x; Delta[2 * i + 1] = c->h_ovA[(size_t)i].y; }
            if (Lambda) { Lambda[2 * i] = c->h_ovB[(size_t)i].x; Lambda[2 * i + 1] = c->h_ovB[(size_t)i].y; }
        }
    }
    return CFMM_OK;
}

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
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return download_trades(c, 0, c->m_total, Delta, Lambda);
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

int cfmm_update_reserves(cfmm_ctx* c)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {
        if (!c->have_trades) return fail(c, CFMM_ERR_STATE, "no materialised trades: call cfmm_find_arb / cfmm_route first");
        for (size_t d = 0; d < c->shards.size(); ++d) {
            cfmm_ctx* child = c->shards[d];
            if (child->segs.empty()) continue;
            const int rc = cfmm_update_reserves(child);
            if (rc != CFMM_OK) return fail(c, rc, "shard %d: %s", (int)d, child->err.c_str());
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
    for (Segment& s : c->segs) {
        if (s.kind != CFMM_KIND_UNIV3) {   // R <- R + γΔ − Λ on the device, no host traffic
            hipError_t e = launch_update_two_coin(s.R, s.gamma, c->d_delta + s.trade_off, c->d_lambda + s.trade_off,
                                                  c->d_over + s.trade_off, c->trades_compact,
                                                  s.kind == CFMM_KIND_GEOMEAN ? s.lR : nullptr, s.eta, s.m, c->stream);
            if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "update launch failed: %s", hipGetErrorString(e));
            continue;
        }
        // UniV3: the pool's state is its price.  find_arb! (src/cfmms.jl:339-395) moves a trading pool to
        // the internal price P = p/γ (price falling, :361) or γ·p (price rising, :381 in the flipped
        // frame), p = v₁/v₂ -- through every fully drained tick and part of the last one -- and leaves a
        // pool inside its no-arbitrage band (:347-349) alone.  P above the first tick means the pool ran
        // out of liquidity on that side and rests at the first tick's upper price.  Tick constants are
        // then re-derived exactly as at upload (compute_at_tick, :294-313).
        const double* v = c->trade_v.data();
        std::vector<double> cp(s.h_cp);
        for (int64_t i = 0; i < s.m; ++i) {
            const double g = s.h_gamma[(size_t)i], q = s.h_cp[(size_t)i];
            const double pr = v[s.h_ai[(size_t)(2 * i)]] / v[s.h_ai[(size_t)(2 * i + 1)]];   // :340
            if (g * q <= pr && pr <= q / g) continue;                                        // :347-349
            double P = pr < g * q ? pr / g : g * pr;
            const double top = s.h_lt[(size_t)s.h_tick_off[(size_t)i]];
            cp[(size_t)i] = P > top ? top : P;
        }
        Segment ns;
        const int rc = univ3_build(c, ns, s.m, cp.data(), s.h_gamma.data(), s.h_ai.data(), s.h_tick_off.data(),
                                   s.h_lt.data(), s.h_liq.data());
        if (rc != CFMM_OK) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // nothing in flight still reads the old constants
        (void)hipFree(s.pg); (void)hipFree(s.Ai); (void)hipFree(s.cur_a); (void)hipFree(s.cur_b); (void)hipFree(s.cur_c);
        (void)hipFree(s.curR); (void)hipFree(s.walk); (void)hipFree(s.ks); (void)hipFree(s.dt); (void)hipFree(s.rout);
        s.pg = ns.pg; s.Ai = ns.Ai; s.cur_a = ns.cur_a; s.cur_b = ns.cur_b; s.cur_c = ns.cur_c; s.curR = ns.curR;
        s.walk = ns.walk; s.ks = ns.ks; s.dt = ns.dt; s.rout = ns.rout; s.deep = ns.deep; s.has_walk = ns.has_walk;
        (void)hipFree(s.cp); (void)hipFree(s.pk);
        s.cp = ns.cp; s.pk = ns.pk; s.gvals.swap(ns.gvals);
        s.h_cp.swap(cp);
    }
    c->have_trades = false;   // consumed: the trades no longer describe an arbitrage of the stored pools
    c->have_out = false;
    c->trade_v.clear();
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

int cfmm_sweep_dev(cfmm_ctx* c, const double* d_v, double* d_out, int materialize)
{
    if (!c || !d_v || !d_out) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_sweep_dev");
    c->have_out = false; // results live on the device; the host copy is stale
    if (materialize) c->trade_v.clear();   // the library has not seen these prices
    return enqueue_sweep(c, d_v, d_out, materialize != 0);
}

int cfmm_trades_dev(cfmm_ctx* c, const double** d_delta, const double** d_lambda)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_trades_dev");
    int rc = ensure_geometry(c);
    if (rc != CFMM_OK) return rc;
    if (c->have_trades ? c->trades_compact != 0 : (c->opt_compact_trades != 0 && !global_bins(c))) {
        // device consumers get the reference's layout: the compact records of the latest materialising sweep are
        // expanded (asynchronously, on the context's stream) into {Δ₁, Δ₂} / {Λ₁, Λ₂} arrays -- call again after
        // every sweep whose trades are wanted
        if (c->m_total > c->x_cap) {
            (void)hipFree(c->d_xdelta); (void)hipFree(c->d_xlambda);
            c->d_xdelta = c->d_xlambda = nullptr;
            c->x_cap = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_xdelta), (size_t)c->m_total * sizeof(double2)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_xlambda), (size_t)c->m_total * sizeof(double2)));
            c->x_cap = c->m_total;
        }
        if (c->have_trades && c->trades_compact) {
            hipError_t e = launch_expand_trades(c->d_delta, c->d_lambda, c->d_over, c->d_xdelta, c->d_xlambda, c->m_total, c->stream);
            if (e != hipSuccess) return fail(c, CFMM_ERR_HIP, "expand launch failed: %s", hipGetErrorString(e));
        }
        if (d_delta) *d_delta = reinterpret_cast<const double*>(c->d_xdelta);
        if (d_lambda) *d_lambda = reinterpret_cast<const double*>(c->d_xlambda);
        return CFMM_OK;
    }
    if (d_delta) *d_delta = reinterpret_cast<const double*>(c->d_delta);
    if (d_lambda) *d_lambda = reinterpret_cast<const double*>(c->d_lambda);
    return CFMM_OK;
}

int cfmm_kernel_times(cfmm_ctx* c, int64_t* sweep_launches, double* sweep_ms, int64_t* reduce_launches,
                      double* reduce_ms)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {   // totals over the shards
        int64_t sn = 0, rn = 0;
        double sm = 0, rm = 0;
        for (cfmm_ctx* child : c->shards) {
            int64_t a = 0, b = 0;
            double x = 0, y = 0;
            int rc = cfmm_kernel_times(child, &a, &x, &b, &y);
            if (rc != CFMM_OK) return fail(c, rc, "%s", child->err.c_str());
            sn += a; rn += b; sm += x; rm += y;
        }
        if (sweep_launches) *sweep_launches = sn;
        if (sweep_ms) *sweep_ms = sm;
        if (reduce_launches) *reduce_launches = rn;
        if (reduce_ms) *reduce_ms = rm;
        return CFMM_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& p : c->pending) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
        if (p.what == 0) { c->t_sweep_n++; c->t_sweep_ms += ms; }
        else { c->t_reduce_n++; c->t_reduce_ms += ms; }
    }
    c->pending.clear();
    c->ev_used = 0;
    if (sweep_launches) *sweep_launches = c->t_sweep_n;
    if (sweep_ms) *sweep_ms = c->t_sweep_ms;
    if (reduce_launches) *reduce_launches = c->t_reduce_n;
    if (reduce_ms) *reduce_ms = c->t_reduce_ms;
    c->t_sweep_n = c->t_reduce_n = 0;
    c->t_sweep_ms = c->t_reduce_ms = 0;
    return CFMM_OK;
}

int cfmm_set_peers(cfmm_ctx* c, const uint64_t* peer_buffers, int32_t world, int32_t rank, uint64_t seq)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_set_peers");
    if (world == 0) { // back to single-GPU operation
        c->peers.clear();
        return CFMM_OK;
    }
    if (!peer_buffers || world < 1 || world > 16 || rank < 0 || rank >= world)
        return fail(c, CFMM_ERR_INVALID_ARG, "bad peer configuration");
    if (global_bins(c)) return fail(c, CFMM_ERR_UNSUPPORTED, "sharded operation is limited to n_tokens <= %d", kMaxLdsTokens);
    c->peers.assign(peer_buffers, peer_buffers + world);
    c->peer_rank = rank;
    c->peer_seq = seq;
    if (const char* e = getenv("CFMM_AMD_PEER_TIMEOUT_S")) {
        const double t = atof(e);
        if (t > 0.0) c->peer_timeout_ticks = (long long)(t * 1e8);
    }
    return CFMM_OK;
}

int cfmm_peer_buffer_alloc(cfmm_ctx* c, uint64_t* d_buf, unsigned char handle[CFMM_IPC_HANDLE_BYTES])
{
    if (!c || !d_buf || !handle) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_peer_buffer_alloc");
    static_assert(sizeof(hipIpcMemHandle_t) == CFMM_IPC_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)(6 * (c->n + 1) + 2) * sizeof(double);
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(c, CFMM_ERR_HIP, "peer buffer export failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
    }
    std::memcpy(handle, &h, sizeof h);
    *d_buf = reinterpret_cast<uint64_t>(p);
    return CFMM_OK;
}

int cfmm_peer_buffer_open(cfmm_ctx* c, const unsigned char handle[CFMM_IPC_HANDLE_BYTES], uint64_t* d_peer)
{
    if (!c || !d_peer || !handle) return CFMM_ERR_INVALID_ARG;
    CFMM_SINGLE_ONLY(c, "cfmm_peer_buffer_open");
    HIP_TRY(c, hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void* p = nullptr;
    HIP_TRY(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *d_peer = reinterpret_cast<uint64_t>(p);
    return CFMM_OK;
}

int cfmm_peer_buffer_close(cfmm_ctx* c, uint64_t d_peer)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipIpcCloseMemHandle(reinterpret_cast<void*>(d_peer)));
    return CFMM_OK;
}

int cfmm_peer_buffer_free(cfmm_ctx* c, uint64_t d_buf)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(reinterpret_cast<void*>(d_buf)));
    return CFMM_OK;
}

int cfmm_lbfgsb_minimize(int32_t n, double* x, const double* lower, const double* upper, const int32_t* nbd,
                         cfmm_fg_callback fg, void* user, int32_t m, double factr, double pgtol, int32_t maxfun,
                         int32_t maxiter, int32_t boxed_from_nbd, cfmm_route_info* info)
{
    if (n < 1 || !x || !nbd || !fg) return fail(nullptr, CFMM_ERR_INVALID_ARG, "bad argument to cfmm_lbfgsb_minimize");
    LbfgsbOptions opt;
    opt.boxed_from_nbd = boxed_from_nbd != 0;
    opt.m = m;
    opt.factr = factr;
    opt.pgtol = pgtol;
    opt.maxfun = maxfun;
    opt.maxiter = maxiter;
    std::vector<int> nb(nbd, nbd + n);
    LbfgsbResult r = lbfgsb_minimize(n, x, lower, upper, nb.data(),
                                     [&](const double* xx, double* gg) { return fg(user, xx, gg); }, opt);
    if (info) {
        info->f = r.f;
        info->proj_grad = r.proj_grad;
        info->iterations = r.iterations;
        info->evaluations = r.evaluations;
        info->sweeps = r.evaluations;
        info->status = r.status;
        info->sweep_seconds = 0.0;
        info->total_seconds = 0.0;
    }
    return CFMM_OK;
}

int cfmm_route(cfmm_ctx* c, int32_t objective_kind, const double* objective_vec, int32_t objective_index,
               const double* v0, int32_t m, double factr, double pgtol, int32_t maxfun, int32_t maxiter,
               double* v_out, double* psi_out, cfmm_route_info* info)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    const int n = c->n;
    if (!objective_vec) return fail(c, CFMM_ERR_INVALID_ARG, "objective vector is null");
    if (objective_kind != CFMM_OBJ_LINEAR_NONNEGATIVE && objective_kind != CFMM_OBJ_BASKET_LIQUIDATION)
        return fail(c, CFMM_ERR_INVALID_ARG, "unknown objective kind %d", objective_kind);
    const bool linear = objective_kind == CFMM_OBJ_LINEAR_NONNEGATIVE;
    if (linear) {
        for (int j = 0; j < n; ++j)
            if (!(objective_vec[j] > 0.0)) // src/objectives.jl:54
                return fail(c, CFMM_ERR_INVALID_ARG, "all elements must be strictly positive");
    } else if (objective_index < 0 || objective_index >= n) {
        return fail(c, CFMM_ERR_INVALID_ARG, "Invalid index i"); // src/objectives.jl:97
    }
    const double* ov = objective_vec;
    const int oi = objective_index;

    // bounds: src/router.jl:67-70 with lower_limit/upper_limit of src/objectives.jl:78-79, :123-129
    std::vector<double> lo(n), up(n, INFINITY), v(n), rv(n);
    std::vector<int> nbd(n, 2);
    const double sqrt_eps = std::sqrt(2.220446049250313e-16);
    for (int j = 0; j < n; ++j) lo[j] = linear ? ov[j] + 1e-8 : sqrt_eps;
    if (!linear) lo[oi] = 1.0 + sqrt_eps;
    for (int j = 0; j < n; ++j) rv[j] = v0 ? v0[j] : 1.0 / n; // src/router.jl:61-65

    int sweeps = 0, rc_inner = CFMM_OK;
    double sweep_s = 0.0;
    const auto t_begin = std::chrono::steady_clock::now();
    const bool armed = can_arm(c);
    struct ArmGuard {   // whatever path leaves this function: no launch stays behind waiting for a price vector
        cfmm_ctx* c;
        ~ArmGuard() { armed_cancel(c); }
    } arm_guard{c};
    auto timed_sweep = [&](const double* x, bool mat) {
        const auto t0 = std::chrono::steady_clock::now();
        if (mat) armed_cancel(c);
        const int rc = (armed && !mat) ? armed_eval(c, x) : host_sweep(c, x, mat);
        sweep_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++sweeps;
        return rc;
    };
    auto sweep = [&](const double* x) { // fused evaluation: Ψ and acc into c->last_out
        rc_inner = timed_sweep(x, false);
        return rc_inner == CFMM_OK;
    };
    // f(objective, v) and grad!(G, objective, v): src/objectives.jl:62-76, :106-121
    auto obj_f = [&](const double* x) -> double {
        if (linear) {
            for (int j = 0; j < n; ++j)
                if (!(ov[j] <= x[j])) return INFINITY;
            return 0.0;
        }
        if (!(x[oi] >= 1.0)) return INFINITY;
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += (j == oi) ? 0.0 : ov[j] * x[j];
        return s;
    };
    auto fg = [&](const double* x, double* G) -> double {
        bool same = true; // src/router.jl:74 / :92: one sweep per evaluation
        for (int j = 0; j < n && same; ++j) same = x[j] == rv[j];
        if (!same) {
            if (!sweep(x)) return NAN;
            std::copy(x, x + n, rv.begin());
        }
        const double fo = obj_f(x);
        const bool feasible = std::isfinite(fo);
        for (int j = 0; j < n; ++j) {
            const double gobj = !feasible ? INFINITY : (linear ? 0.0 : (j == oi ? 0.0 : ov[j]));
            G[j] = gobj + c->last_out[(size_t)j]; // src/router.jl:96-100
        }
        return fo + c->last_out[(size_t)n];        // src/router.jl:85
    };

    if (!sweep(rv.data())) return rc_inner; // src/router.jl:104
    std::copy(rv.begin(), rv.end(), v.begin());
    LbfgsbOptions opt;
    opt.m = m;
    opt.factr = factr;
    opt.pgtol = pgtol;
    opt.maxfun = maxfun;
    opt.maxiter = maxiter;
    opt.boxed_from_nbd = true; // bounds[1,:] .= 2 with an infinite upper limit: the Fortran's "boxed" path
    LbfgsbResult r = lbfgsb_minimize(n, v.data(), lo.data(), up.data(), nbd.data(), fg, opt); // :105
    if (rc_inner != CFMM_OK) return rc_inner;
    int rc = timed_sweep(v.data(), true); // src/router.jl:106-107: r.v = v*, find_arb!(r, v*)
    if (rc != CFMM_OK) return rc;
    if (v_out) std::copy(v.begin(), v.end(), v_out);
    if (psi_out) std::memcpy(psi_out, c->last_out.data(), (size_t)n * sizeof(double));
    if (info) {
        info->f = r.f;
        info->proj_grad = r.proj_grad;
        info->iterations = r.iterations;
        info->evaluations = r.evaluations;
        info->sweeps = sweeps;
        info->status = r.status;
        info->sweep_seconds = sweep_s;
        info->total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return CFMM_OK;
}

int32_t cfmm_segment_count(const cfmm_ctx* c)
{
    return c ? (int32_t)(c->shards.empty() ? c->segs.size() : c->psegs.size()) : 0;
}

int cfmm_segment_info(const cfmm_ctx* c, int32_t seg, int32_t* kind, int64_t* m, int32_t* block, int32_t* grid,
                      int32_t* unroll)
{
    if (!c) return CFMM_ERR_INVALID_ARG;
    if (!c->shards.empty()) {   // kind and size of the whole segment; launch geometry of shard 0's block
        if (seg < 0 || seg >= (int32_t)c->psegs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
        const int cs = child_segment(c, seg, 0);
        if (cs >= 0) {
            int rc = cfmm_segment_info(c->shards[0], cs, kind, nullptr, block, grid, unroll);
            if (rc != CFMM_OK) return rc;
        }
        if (kind) *kind = c->psegs[(size_t)seg].kind;
        if (m) *m = c->psegs[(size_t)seg].m;
        return CFMM_OK;
    }
    if (seg < 0 || seg >= (int32_t)c->segs.size()) return fail(c, CFMM_ERR_INVALID_ARG, "segment out of range");
    int rc = ensure_geometry(const_cast<cfmm_ctx*>(c));
    if (rc != CFMM_OK) return rc;
    const Segment& s = c->segs[(size_t)seg];
    if (kind) *kind = s.kind;
    if (m) *m = s.m;
    if (block) *block = s.block;
    if (grid) *grid = s.grid;
    if (unroll) *unroll = s.unroll;
    return CFMM_OK;
}

} // extern "C"
